#!/bin/bash
# round 3, GPU call 22: fp64 level 0 -- the tests that failed in call 21 + the finish tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
timeout 1500 python -m pytest tests/test_bn_finish_gpu.py tests/test_fullsize_fixture.py tests/test_ops_gpu.py tests/test_conv_v4_gpu.py tests/test_fullsize_gpu.py -q -m gpu > $O/pytest.log 2>&1
echo "rc=$?" | tee -a $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -12
grep -n "d2_conv1.1.weight\|worst" gpurun_out/parity_fixture_8x_train.txt | head
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_$i.log 2>&1; echo "bench $i $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log)"; done
