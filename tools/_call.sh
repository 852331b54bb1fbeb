#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "roi_align or detector or autocast or fp16 or bf16" > gpurun_out/r04_tests_roi.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r04_tests_roi.log
for a in "32" "32 bf16"; do
  timeout 600 python tools/detector_bench.py $a 2>/dev/null | grep -E "^batch|RoIAlign"
done
