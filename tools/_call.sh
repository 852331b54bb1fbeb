#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_tests_gpu_full.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04_tests_gpu_full.log
