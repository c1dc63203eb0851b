#!/bin/bash
# round 3, GPU call 33: bench exit status and output order in its three forms (plain, single-rank RCCL, inference)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4h
mkdir -p $O
VIRCONV_FORCE_DDP=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --family-steps 0 > $O/ddp.log 2>&1; echo "ddp rc=$? last: $(tail -n 1 $O/ddp.log | cut -c1-70)"
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/plain.log 2>&1; echo "plain rc=$? last: $(tail -n 1 $O/plain.log | cut -c1-110)"
timeout 200 python bench.py --mode infer --batch-size 1 --steps 10 --warmup 3 > $O/infer.log 2>&1; echo "infer rc=$? last: $(tail -n 1 $O/infer.log | cut -c1-110)"
