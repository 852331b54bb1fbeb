#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
for f in 0 1; do
  RGRG_LN_FOLD=$f timeout 600 python bench.py --batch 32 --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline --no-config2 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('fold=$f', j['value'], j['ms_per_step'])"
done
done
