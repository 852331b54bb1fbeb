#!/bin/bash
# Round-end check on the MI355X box (run through gpurun from the repo root): build + smoke, the whole `-m gpu` suite, the default
# bench line (configs[2]; configs[1], the beam mode and configs[4] ride as legs; the all-cores CPU run is on by default).  Outputs under gpurun_out/ (copy what is judged to profiles/).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r06}
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 | tee gpurun_out/${TAG}_gpu_test_suite.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python - <<PY
import json
r = json.loads(open("gpurun_out/${TAG}_bench_default.json").read().strip().splitlines()[-1])
print("configs[2] (default line)", round(r["value"], 2), r["unit"], "frac", round(r["roofline"]["frac"], 4))
for k in ("config1", "beam4", "config4"):
    v = r[k]; print(k, round(v["value"], 3), round(v["ms_per_step"], 2), "ms", "frac", round((v.get("roofline") or {}).get("frac", 0), 4))
cb = r["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], json.dumps(cb.get("all_physical_cores_full_run")))
PY
