#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_generate.py tests/test_gpu_persistent.py tests/test_gpu_parity_gaps.py -x -q -m gpu > gpurun_out/r04_tests_attn.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04_tests_attn.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config2 2>/dev/null | grep '^{' | python -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('b1', j['value'], j['ms_per_step'])"
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config2 2>/dev/null | grep '^{' | python -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('b1', j['value'], j['ms_per_step'])"
