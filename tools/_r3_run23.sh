#!/bin/bash
# round 3, GPU call 23: the driver-form bench line with the GEMM-only roofline fields
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r3x
mkdir -p $O
python bench.py > $O/bench_full.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.log 2>&1
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline > $O/bench_8x.log 2>&1
for f in $O/*.log; do echo $f; tail -n 1 $f | cut -c1-300; done
