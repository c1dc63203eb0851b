#!/bin/bash
# round 3, GPU call 15: pair-compacted forward kernel, deeper pipeline -- parity tests + kernel table A/B
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
timeout 900 python -m pytest tests/test_conv_pc_gpu.py -x -q -m gpu > $O/pytest_pc.log 2>&1
echo "pytest pc rc=$?" | tee -a $O/pytest_pc.log
tail -4 $O/pytest_pc.log
for v in 1 0; do
  VIRCONV_DEBUG_SET="conv_pc=$v" timeout 300 python tools/kbench.py --layers down,conv_out --only fwd --autopack > $O/kbench_pc$v.txt 2>&1
  grep -E "down|conv_out" $O/kbench_pc$v.txt | tail -4
done
for v in 1 0; do
  VIRCONV_DEBUG_SET="conv_pc=$v" timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_pc${v}.log 2>&1
done
for f in $O/bench_pc*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f)"; done
