set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_script.py tests/test_gpu_generate.py -x -q -m gpu -k "persistent or script or presents_are_invalidated or incremental" > gpurun_out/r04_tests_new.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r04_tests_new.log
timeout 900 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
echo "bench rc=$?"; tail -c 6000 gpurun_out/r04_bench_default.json; tail -5 gpurun_out/r04_bench_default.err
