set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_generate.py tests/test_gpu_kernels.py -x -q -m gpu -k "incremental or presents or beam or roi_align or decoder_matches" > gpurun_out/r04_tests_api.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r04_tests_api.log
