#!/bin/bash
# Round-6 counter evidence for the many-sequence 16-bit decode step (run through gpurun from the repo root): separate rocprofv3
# --pmc FETCH_SIZE / WRITE_SIZE passes (counters in their own runs, no trace domains) over tools/pmc_decode_kernels.py - the
# step's kernels at their shapes as a micro-bench; the counter mode does not survive the decode process itself
# (profiles/r05_rocprofv3_pmc_crash_reproducer.md).  Output: gpurun_out/r06_pmc_decode_kernels.md / .json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
RAW=/tmp/rgrg_pmc6
rm -rf $RAW; mkdir -p "$OUT" "$RAW"
export TMPDIR=/tmp
cd /tmp
for what in ${1:-gemm attn}; do
  extra=""; [ $what = attn ] && extra="--attention"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== $what $c"
    timeout 420 rocprofv3 --pmc $c --output-format csv -d $RAW/${what}_$c -- python $ROOT/tools/pmc_decode_kernels.py $extra --manifest $RAW/${what}_manifest.json > $OUT/r06_pmc_${what}_$c.log 2>&1
    echo "rc=$? $(tail -1 $OUT/r06_pmc_${what}_$c.log | cut -c1-200)"
  done
  python $ROOT/tools/pmc_decode_summary.py $RAW/${what}_FETCH_SIZE $RAW/${what}_WRITE_SIZE $RAW/${what}_manifest.json $OUT/r06_pmc_decode_kernels.md $OUT/r06_pmc_decode_kernels.json
done
