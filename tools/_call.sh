#!/bin/bash
# Scratch script for `gpurun -- 'bash tools/_call.sh'` calls: edit per call.  This version is the round-end check -
# smoke, the full GPU suite, the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_test_suite.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/gpu_test_suite.log
timeout 900 python bench.py > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err
echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_b1.json
