#!/bin/bash
# scratch GPU call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('b1', j['value'], j['ms_per_step'])
for k in ('roofline','roofline_secondary','roofline_detector','roofline_roialign'):
    r=j[k]; print(k, r.get('frac'), r.get('avg_launch_us'), r.get('ms'))
c=j['config2']; print('config2', c['value'], c['ms_per_step'], c['roofline']['frac'], c['roofline']['avg_launch_us'], c['roofline_secondary']['frac'], c['roofline_roialign']['frac'])
print('config4', j['config4']['value'])"
