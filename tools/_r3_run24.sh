#!/bin/bash
# round 3, GPU call 24: table loads of the gather-GEMM prologue batched -- kernel table, step, then the whole GPU suite
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3y
mkdir -p $O
timeout 300 python tools/kbench.py > $O/kbench.txt 2>&1
tail -18 $O/kbench.txt | cut -c1-170
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_$i.log 2>&1; echo "bench $i $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log)"; done
timeout 200 python bench.py --mode infer --batch-size 1 > $O/infer_bs1.log 2>&1; echo "infer1 $(grep -o '"ms_per_step": [0-9.]*' $O/infer_bs1.log)"
timeout 200 python bench.py --mode infer --batch-size 4 > $O/infer_bs4.log 2>&1; echo "infer4 $(grep -o '"ms_per_step": [0-9.]*' $O/infer_bs4.log)"
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1
echo "gpu suite rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_all.log | tail -8
