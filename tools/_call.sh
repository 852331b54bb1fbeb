#!/bin/bash
# scratch GPU call: the round's final check - full GPU suite, smoke, default bench line, kernel traces of the final build
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_test_suite.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r04_gpu_test_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r04_bench_b1.json 2> gpurun_out/r04_bench_b1.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/r04_bench_b1.json
export TMPDIR=/tmp; cd /tmp
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b32 -- python $ROOT/bench.py --no-cpu-baseline --no-config2 --batch 32 --dtype bf16 --steps 2 --warmup 1 > $ROOT/gpurun_out/r04_kt_b32.log 2>&1
python $ROOT/tools/prof_summary.py /tmp/kt_b32 $ROOT/gpurun_out/r04_kernel_trace_summary_b32_bf16.md > /dev/null
cp $(find /tmp/kt_b32 -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/r04_kernel_stats_b32_bf16.csv 2>/dev/null
head -10 $ROOT/gpurun_out/r04_kernel_trace_summary_b32_bf16.md | cut -c1-120
