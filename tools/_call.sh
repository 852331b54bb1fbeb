#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for a in "32" "32 bf16"; do
  timeout 300 python tools/detector_bench.py $a 2>/dev/null | grep -E "RoIAlign|fc6"
done
