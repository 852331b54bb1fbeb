#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm_folded or lds_dma" > gpurun_out/r04_tests_lnfold_unit.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r04_tests_lnfold_unit.log | cut -c1-250
