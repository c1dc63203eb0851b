#!/bin/bash
# round 3, GPU call 30: host-side cost of a step (the step time varies 5.07-5.47 ms across boxes at identical kernel times)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4e
mkdir -p $O
timeout 150 python tools/hostprof.py > $O/hostprof.txt 2>&1
head -45 $O/hostprof.txt | cut -c1-150
lscpu | grep -i "model name\|MHz" | head -3
