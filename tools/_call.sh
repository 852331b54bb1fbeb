#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_train16.py -x -q -k "argmax" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity_gaps.py -x -q -k "split_k or argmax_epilogue" 2>&1 | tail -25
for c in 1 0 1 0; do RGRG_LMHEAD_CAND=$c timeout 600 python bench.py --no-cpu-baseline --no-config2 --batch 32 --dtype bf16 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lmhead_cand=$c', round(r['value'],2), round(r['ms_per_step'],1), 'gemm ms/step', round(r['roofline']['ms_per_decode_step'],3))"; done
