#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity_r03.py -x -q -k "configs2" 2>&1 | tail -2
for cfg in "1 1" "2 1" "4 1" "2 2"; do set -- $cfg; RGRG_SK_MLP=$1 RGRG_SK_ATTN=$2 timeout 600 python bench.py --no-cpu-baseline --no-config2 --batch 32 --dtype bf16 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('sk_mlp=$1 sk_attn=$2', round(r['value'],2), round(r['ms_per_step'],1), 'gemm ms/step', round(r['roofline']['ms_per_decode_step'],3) if r['roofline'].get('bound')=='mfma' else round(r['roofline_secondary']['ms_per_decode_step'],3))"; done
