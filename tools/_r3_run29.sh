#!/bin/bash
# round 3, GPU call 29: box-to-box check (two bench lines + the stand-alone kernel totals on the same box)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4d
mkdir -p $O
for i in 1 2; do python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_$i.log 2>&1; echo "bench $i $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log)"; done
timeout 100 python tools/kbench.py --iters 10 > $O/kbench.txt 2>&1; tail -1 $O/kbench.txt
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
