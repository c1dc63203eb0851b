#!/bin/bash
# round 3, GPU call 12: in-kernel BatchNorm finish -- bit-identity tests, then the A/B of the train step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3l
mkdir -p $O
timeout 900 python -m pytest tests/test_bn_finish_gpu.py -x -q -m gpu > $O/pytest_finish.log 2>&1
echo "pytest finish rc=$?" >> $O/pytest_finish.log
tail -5 $O/pytest_finish.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_golden_8x.py tests/test_round3_gpu.py -x -q -m gpu -k "native or pass or post_act or one_launch or bit" > $O/pytest_pass.log 2>&1
echo "pytest pass rc=$?" >> $O/pytest_pass.log
tail -3 $O/pytest_pass.log
for v in 1 0 1 0; do
  VIRCONV_DEBUG_SET="conv_bn_finish=$v" timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_fin$v.log 2>&1
  echo "fin=$v: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_fin$v.log)"
done
timeout 300 python tools/step_phases.py > $O/step_phases.txt 2>&1
tail -2 $O/step_phases.txt
