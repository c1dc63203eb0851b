#!/bin/bash
# round 3, GPU call 14: pair-compacted forward kernel -- parity tests, kernel table A/B, train step A/B
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3n
mkdir -p $O
timeout 900 python -m pytest tests/test_conv_pc_gpu.py -x -q -m gpu > $O/pytest_pc.log 2>&1
echo "pytest pc rc=$?" | tee -a $O/pytest_pc.log
tail -15 $O/pytest_pc.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_conv_v4_gpu.py -x -q -m gpu > $O/pytest_reg.log 2>&1
echo "pytest regression rc=$?" | tee -a $O/pytest_reg.log
tail -4 $O/pytest_reg.log
for v in 1 0; do
  VIRCONV_DEBUG_SET="conv_pc=$v" timeout 300 python tools/kbench.py --layers down,conv_out --only fwd --autopack > $O/kbench_pc$v.txt 2>&1
  grep -E "down|conv_out" $O/kbench_pc$v.txt | tail -5
done
for v in 1 0 1 0; do
  VIRCONV_DEBUG_SET="conv_pc=$v" timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_pc${v}_$RANDOM.log 2>&1
done
for f in $O/bench_pc*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f)"; done
VIRCONV_DEBUG_SET="conv_pc=1" timeout 200 python bench.py --mode infer --batch-size 4 > $O/infer4_pc1.log 2>&1
VIRCONV_DEBUG_SET="conv_pc=0" timeout 200 python bench.py --mode infer --batch-size 4 > $O/infer4_pc0.log 2>&1
for f in $O/infer4*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f)"; done
