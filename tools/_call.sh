#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_train16.py -x -q -k "attention_paths or dropout or partition" 2>&1 | tail -3
timeout 300 python tools/train_bench.py 8 64 3 bf16 2>&1 | tail -1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_train -- python $ROOT/tools/train_bench.py 8 64 3 bf16 > $ROOT/gpurun_out/r05_train_b8_v5.log 2>&1
python $ROOT/tools/prof_summary.py /tmp/kt_train $ROOT/gpurun_out/r05_kernel_trace_summary_train_b8_v5.md > /dev/null
head -16 $ROOT/gpurun_out/r05_kernel_trace_summary_train_b8_v5.md
