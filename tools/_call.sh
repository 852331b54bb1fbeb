#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_samplers.py tests/test_gpu_detector_losses.py -x -q -m gpu > gpurun_out/r04_tests_samplers.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r04_tests_samplers.log
timeout 600 python tools/sampler_bench.py > gpurun_out/r04_sampler_bench.log 2>&1
echo "rc=$?"
tail -4 gpurun_out/r04_sampler_bench.log
