#!/bin/bash
# round 3, GPU call 26: weight-gradient fork behind a kernel completion event -- dependency check, A/B, whole GPU suite
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a
mkdir -p $O
timeout 300 python -m pytest tests/test_round3_gpu.py -q -m gpu -k "completion_event or weighted" > $O/pytest_dep.log 2>&1
echo "dep rc=$?"; tail -3 $O/pytest_dep.log
for v in 1 0 1 0; do
  VIRCONV_DEBUG_SET="pass_fork_ext_event=$v" timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_ext${v}_$RANDOM.log 2>&1
done
for f in $O/bench_ext*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f)"; done
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
echo "gpu suite rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_all.log | tail -8
