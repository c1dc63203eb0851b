cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5k; mkdir -p $D
timeout 600 python -m pytest tests/test_bev_stem.py tests/test_golden_8x.py tests/test_plan_stress_gpu.py -m gpu -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $D/tests.log | cut -c1-300
timeout 300 python tools/bevbench.py > $D/bevbench.txt 2>&1; tail -n 4 $D/bevbench.txt | cut -c1-250
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
timeout 300 python bench.py --model 8x $B40 > $D/bench_8x.log 2>&1; echo "8x $(grep -o '"ms_per_step": [0-9.]*' $D/bench_8x.log | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $D/bench_8x.log | head -1)"
timeout 300 python bench.py --model 8x $B40 > $D/bench_8x_2.log 2>&1; echo "8x $(grep -o '"ms_per_step": [0-9.]*' $D/bench_8x_2.log | head -1)"
for no in "" "--no-xcd"; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$D/pmc_s3down${no}" -o x -- \
      python "$GRAFT_REPO_ROOT/tools/kbench.py" --layers s3.down --only bwd --iters 5 $no > "$GRAFT_REPO_ROOT/$D/p_s3down${no}.log" 2>&1 ); done
find $D -name '*kernel_trace.csv' -size +20M -delete
echo finished
