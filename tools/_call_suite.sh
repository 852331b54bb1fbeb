#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -q -m gpu --durations=8 2>&1 | grep -v "^/opt" | tail -24 | tee gpurun_out/r05_gpu_test_suite.log
