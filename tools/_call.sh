#!/bin/bash
# scratch GPU call: PMC traffic of the batch-32 decode GEMMs through the micro-benchmark
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/F -- python $ROOT/tools/gemm_bf16_bench.py --cold --tiles 0 --iters 6 > /tmp/f.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/W -- python $ROOT/tools/gemm_bf16_bench.py --cold --tiles 0 --iters 6 > /tmp/w.log 2>&1; echo "write rc=$?"
python $ROOT/tools/pmc_gemm_step_traffic.py /tmp/F /tmp/W $ROOT/gpurun_out/r04_pmc_traffic.json | tail -30
