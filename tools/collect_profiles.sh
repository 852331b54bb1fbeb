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

B1="python $ROOT/bench.py --no-cpu-baseline --no-config2 --steps 3 --warmup 1"
B32="python $ROOT/bench.py --no-cpu-baseline --no-config2 --batch 32 --dtype bf16 --steps 2 --warmup 1"
P1="python $ROOT/bench.py --no-cpu-baseline --no-config2 --steps 1 --warmup 0"
# batch-32 counter passes: the many-sequence decode process alone with its steps launched EAGERLY, the row ranges one after the
# other on one stream - rocprofv3 --pmc segfaults on the hipGraph replays of that step; on the final round-5 tree it does not
# survive this form either (profiles/r05_rocprofv3_pmc_crash_reproducer.md), the two passes then leave no CSV and
# pmc_traffic.py keeps the previous entries
P32="python $ROOT/tools/decode_pmc_probe.py 923 128 0"

run() {  # name, rocprof args..., -- command
    local name=$1; shift
    echo "== $name"
    timeout 420 rocprofv3 "$@" > "$OUT/${TAG}_$name.log" 2>&1
    echo "rc=$? $(tail -1 "$OUT/${TAG}_$name.log" | cut -c1-300)"
}

run kt_b1  --kernel-trace --stats --output-format csv -d $RAW/kt_b1  -- $B1
python $ROOT/tools/prof_summary.py $RAW/kt_b1 "$OUT/${TAG}_kernel_trace_summary_b1.md" > /dev/null
cp $(find $RAW/kt_b1 -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_kernel_stats_b1.csv" 2>/dev/null
run kt_b32 --kernel-trace --stats --output-format csv -d $RAW/kt_b32 -- $B32
python $ROOT/tools/prof_summary.py $RAW/kt_b32 "$OUT/${TAG}_kernel_trace_summary_b32_bf16.md" > /dev/null
cp $(find $RAW/kt_b32 -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_kernel_stats_b32_bf16.csv" 2>/dev/null
rm -rf $RAW/kt_b1 $RAW/kt_b32

run pmc_fetch_b1  --pmc FETCH_SIZE --output-format csv -d $RAW/f1 -- $P1
run pmc_write_b1  --pmc WRITE_SIZE --output-format csv -d $RAW/w1 -- $P1
python $ROOT/tools/pmc_traffic.py $RAW/f1 $RAW/w1 S29_f32 "$OUT/${TAG}_pmc_traffic.json" > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/f1 "$OUT/${TAG}_pmc_fetch_size_b1.md" > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/w1 "$OUT/${TAG}_pmc_write_size_b1.md" > /dev/null
rm -rf $RAW/f1 $RAW/w1
run pmc_fetch_b32 --pmc FETCH_SIZE --output-format csv -d $RAW/f32 -- $P32
run pmc_write_b32 --pmc WRITE_SIZE --output-format csv -d $RAW/w32 -- $P32
python $ROOT/tools/pmc_traffic.py $RAW/f32 $RAW/w32 S923_bf16 "$OUT/${TAG}_pmc_traffic.json" > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/f32 "$OUT/${TAG}_pmc_fetch_size_b32_bf16.md" > /dev/null
python $ROOT/tools/pmc_summary.py $RAW/w32 "$OUT/${TAG}_pmc_write_size_b32_bf16.md" > /dev/null
rm -rf $RAW
ls -la "$OUT" | grep "${TAG}_" | tail -30
