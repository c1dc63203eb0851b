#!/bin/bash
# round 3, GPU call 32: bench output order with RCCL initialised, plain form still fine
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4g
mkdir -p $O
VIRCONV_FORCE_DDP=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --family-steps 0 > $O/bench_force_ddp.log 2>&1
echo "last line starts with: $(tail -n 1 $O/bench_force_ddp.log | cut -c1-60)"
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --family-steps 0 2>/dev/null | tail -n 1 | cut -c1-120
timeout 200 python bench.py --mode infer --batch-size 1 --steps 10 --warmup 3 2>/dev/null | tail -n 1 | cut -c1-120
