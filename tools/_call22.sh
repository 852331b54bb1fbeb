set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_generate.py -q -x -k "training_pass_vs_oracle_autograd or with_dropout_matches or training_pass_gradients_match or two_training_steps" --durations=8 > $OUT/r03_train_long_T.log 2>&1
tail -15 $OUT/r03_train_long_T.log
timeout 300 python tools/gemm_bf16_bench.py --cold --vendor --tiles 0 > $OUT/r03_gemm_bf16_bench_v9_vendor_yardstick_cold.log 2>&1
cat $OUT/r03_gemm_bf16_bench_v9_vendor_yardstick_cold.log | grep -v amdgpu.ids
cd /tmp
RAW=/tmp/rgrg_prof; mkdir -p $RAW
P32="python $ROOT/bench.py --no-cpu-baseline --no-config2 --batch 32 --dtype bf16 --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-include-regex "gemm_bf16_glds|attn_decode" --output-format csv -d $RAW/$c -- $P32 > $OUT/r03_pmc_${c}_b32.log 2>&1
  echo "pmc $c rc=$? $(find $RAW/$c -name '*counter_collection.csv' | wc -l) csv"
done
python $ROOT/tools/pmc_traffic.py $RAW/FETCH_SIZE $RAW/WRITE_SIZE S923_bf16 $OUT/r03_pmc_traffic.json
python $ROOT/tools/pmc_summary.py $RAW/FETCH_SIZE $OUT/r03_pmc_fetch_size_b32_bf16.md > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/WRITE_SIZE $OUT/r03_pmc_write_size_b32_bf16.md > /dev/null
