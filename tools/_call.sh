#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" 2>&1 | tail -1
timeout 900 python tools/_chains_check.py 2>&1 | tail -18
