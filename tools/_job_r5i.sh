cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5i; mkdir -p $D
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
line() { for f in "$@"; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' "$f" | head -1)"; done; }
for i in 1 2; do
  timeout 200 python bench.py $B40 > $D/b40_base_$i.log 2>&1
  VIRCONV_DEBUG_SET=group_plan_onesweep=1 timeout 200 python bench.py $B40 > $D/b40_onesweep_$i.log 2>&1
  VIRCONV_DEBUG_SET=bw_rows_per_split=2048 timeout 200 python bench.py $B40 > $D/b40_rows2048_$i.log 2>&1
done
VIRCONV_DEBUG_SET=bw_rows_per_split=4096 timeout 200 python bench.py $B40 > $D/b40_rows4096_1.log 2>&1
VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so timeout 200 python bench.py $B40 > $D/b40_agprform_1.log 2>&1
line $D/b40_*.log
VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so timeout 400 python tools/det_check.py --reps 30 --bs 2 base > $D/det_check_agprform.txt 2>&1; echo "det(agpr-form) rc=$?"; grep "reps differ" $D/det_check_agprform.txt
timeout 300 python bench.py --model 8x --mode infer --steps 30 --warmup 10 > $D/infer_8x_rot3.log 2>&1; line $D/infer_8x_rot3.log
timeout 300 python bench.py --model 8x --operand f16 $B40 > $D/bench_8x_f16.log 2>&1; line $D/bench_8x_f16.log
timeout 300 python bench.py --operand f16 $B40 > $D/bench_f16.log 2>&1; line $D/bench_f16.log
timeout 300 python bench.py --frontend $B40 > $D/bench_frontend.log 2>&1; line $D/bench_frontend.log
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$D/pmc_$c" -o x -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --family-steps 0 --exact-steps 0 > "$GRAFT_REPO_ROOT/$D/p_$c.log" 2>&1 ); done
find $D -name '*kernel_trace.csv' -size +20M -delete
ls $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE | head
echo finished
