#!/bin/bash
# round 3, GPU call 27: bench lines of the final build
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r4b
mkdir -p $O
python bench.py > $O/bench_full.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.log 2>&1
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $O/bench_8x.log 2>&1
for f in $O/*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"value": [0-9.]*' $f | head -1); done
