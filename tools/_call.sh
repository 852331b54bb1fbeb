#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_generate.py tests/test_gpu_parity_gaps.py tests/test_gpu_parity_r03.py tests/test_gpu_fp16.py tests/test_gpu_persistent.py -x -q -m gpu > gpurun_out/r04_tests_attn.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04_tests_attn.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('b1', j['value'], j['ms_per_step'], 'attn', j['roofline_secondary']['avg_launch_us'], j['roofline_secondary']['frac'], 'gemm', j['roofline']['frac']); c=j['config2']; print('config2', c['value'], c['ms_per_step'], c['roofline_secondary']['kernel'], c['roofline_secondary']['avg_launch_us'], c['roofline_secondary']['frac'])"
