#!/bin/bash
# Collects the per-round rocprofv3 evidence on the MI355X box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of `bench.py` at batch 1 fp32 and batch 32 bf16 -> gpurun_out/<tag>_kernel_trace_summary_*.md
#   2. separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same two commands (counters in their own runs, no
#      trace domains beside them) -> gpurun_out/<tag>_pmc_traffic.json + per-kernel tables
# The raw CSVs are summarised on the box and deleted (gpurun merges at most 64 MiB back).
# Usage: tools/collect_profiles.sh r03
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
RAW=/tmp/rgrg_prof
mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
cd /tmp

# round 6: the default bench line is batch 32 under bf16 autocast (BASELINE configs[2]); batch 1 in fp32 (configs[1]) is asked for
B1="python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --batch 1 --dtype f32 --steps 3 --warmup 1"
B32="python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1"
P1="python $ROOT/bench.py --no-cpu-baseline --no-extra-legs --batch 1 --dtype f32 --steps 1 --warmup 0"
# batch-32 counter passes: the many-sequence decode process alone with its steps launched EAGERLY, the row ranges one after the
# other on one stream - rocprofv3 --pmc segfaults on the hipGraph replays of that step; on the final round-5 tree it does not
# survive this form either (profiles/r05_rocprofv3_pmc_crash_reproducer.md), the two passes then leave no CSV and
# pmc_traffic.py keeps the previous entries
P32="python $ROOT/tools/decode_pmc_probe.py 923 128 0"

run() {  # name, rocprof args..., -- command
    local name=$1; shift
    echo "== $name"
    timeout 600 rocprofv3 "$@" > "$OUT/${TAG}_$name.log" 2>&1
    echo "rc=$? $(tail -1 "$OUT/${TAG}_$name.log" | cut -c1-300)"
}

run kt_b1  --kernel-trace --stats --output-format csv -d $RAW/kt_b1  -- $B1
python $ROOT/tools/prof_summary.py $RAW/kt_b1 "$OUT/${TAG}_kernel_trace_summary_b1.md" > /dev/null
cp $(find $RAW/kt_b1 -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_kernel_stats_b1.csv" 2>/dev/null
run kt_b32 --kernel-trace --stats --output-format csv -d $RAW/kt_b32 -- $B32
python $ROOT/tools/prof_summary.py $RAW/kt_b32 "$OUT/${TAG}_kernel_trace_summary_b32_bf16.md" > /dev/null
cp $(find $RAW/kt_b32 -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_kernel_stats_b32_bf16.csv" 2>/dev/null
python $ROOT/tools/step_timeline.py $RAW/kt_b32 "$OUT/${TAG}_step_timeline_b32_bf16.md" > /dev/null
# the same command with the step as ONE row range: every launch covers all ~923 rows - the launches the line's roofline times
# (rocprofv3's kernel trace serialises the queues of the 4-range step anyway: its per-launch durations are those of ~231-row
# launches running alone, which the real step never sees)
RGRG_DECODE_CHAINS=1 run kt_b32_one --kernel-trace --stats --output-format csv -d $RAW/kt_b32_one -- $B32
python $ROOT/tools/prof_summary.py $RAW/kt_b32_one "$OUT/${TAG}_kernel_trace_summary_b32_bf16_one_range.md" > /dev/null
rm -rf $RAW/kt_b1 $RAW/kt_b32 $RAW/kt_b32_one

run pmc_fetch_b1  --pmc FETCH_SIZE --output-format csv -d $RAW/f1 -- $P1
run pmc_write_b1  --pmc WRITE_SIZE --output-format csv -d $RAW/w1 -- $P1
python $ROOT/tools/pmc_traffic.py $RAW/f1 $RAW/w1 S29_f32 "$OUT/${TAG}_pmc_traffic.json" > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/f1 "$OUT/${TAG}_pmc_fetch_size_b1.md" > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/w1 "$OUT/${TAG}_pmc_write_size_b1.md" > /dev/null
rm -rf $RAW/f1 $RAW/w1
# batch 32: the counter passes run over the step's kernels as a micro-bench (tools/collect_pmc_r06.sh): rocprofv3 --pmc does not
# survive the decode process itself (profiles/r05_rocprofv3_pmc_crash_reproducer.md)
rm -rf $RAW
ls -la "$OUT" | grep "${TAG}_" | tail -30
