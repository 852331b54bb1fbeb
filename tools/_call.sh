#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -k "bf16 or fp16 or autocast or configs2 or layernorm_folded" > gpurun_out/r04_tests_lnfold.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04_tests_lnfold.log
