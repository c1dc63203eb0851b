#!/bin/bash
# round 3, GPU call 31: the single-rank RCCL form of the bench (flat gradient all-reduce + the extra roofline steps) on the final tree
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4f
mkdir -p $O
VIRCONV_FORCE_DDP=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_force_ddp.log 2>&1
tail -n 1 $O/bench_force_ddp.log | cut -c1-400
