#!/bin/bash
# round 3, GPU call 28: last sanity run of the final tree (smoke, the newest tests, one bench line)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4c
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python -m pytest tests/test_bn_finish_gpu.py tests/test_round3_gpu.py tests/test_conv_pc_gpu.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
python bench.py --no-cpu-baseline --family-steps 0 > $O/bench.log 2>&1; echo "bench $(grep -o '"ms_per_step": [0-9.]*' $O/bench.log)"
