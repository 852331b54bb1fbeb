#!/bin/bash
# Round-end check on the MI355X box (run through gpurun from the repo root): build + smoke, the whole `-m gpu` suite, the default
# bench line (with the opt-in full CPU run on all physical cores).  Outputs under gpurun_out/ (copy what is judged to profiles/).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r05}
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 | tee gpurun_out/${TAG}_gpu_test_suite.log
timeout 900 python bench.py --cpu-baseline-all-cores > gpurun_out/${TAG}_bench_b1.json 2> gpurun_out/${TAG}_bench_b1.err
python - <<PY
import json
r = json.loads(open("gpurun_out/${TAG}_bench_b1.json").read().strip().splitlines()[-1])
print("configs[1]", round(r["value"], 3), r["unit"], "frac", round(r["roofline"]["frac"], 4))
print("configs[2]", round(r["config2"]["value"], 2), "frac", round(r["config2"]["roofline"]["frac"], 4))
c4 = r["config4"]; print("configs[4]", round(c4["ms_per_step"], 2), "ms", round(c4["value"], 1), "frac", round(c4["roofline"]["frac"], 4))
cb = r["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], json.dumps(cb.get("all_physical_cores_full_run")))
json.dump(cb, open("gpurun_out/${TAG}_cpu_baseline_all_cores.json", "w"), indent=1)
PY
