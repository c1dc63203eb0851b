#!/bin/bash
# round 3, GPU call 21: whole GPU suite (no -x)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3u
mkdir -p $O
timeout 2000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/pytest_all.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_all.log | tail -30
