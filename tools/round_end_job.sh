# The round-end measurement job of round 6 (profiles/r06_README.md): full GPU suite, the -DVC_EXPERIMENTS suite, smoke, the bench lines,
# phases, host split, kernel table, rocprof stats + PMC passes, the A.17 lab's short form.
#   python -m virconv_amd.build; VIRCONV_LIB_OUT=$PWD/virconv_amd/libvirconv_hip_exp.so VIRCONV_HIPCC_EXTRA=-DVC_EXPERIMENTS python -m virconv_amd.build
#   gpurun --timeout 3400 -- bash tools/round_end_job.sh
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; D=gpurun_out/${TAG:-r6z}; mkdir -p $D
line() { for f in "$@"; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"value": [0-9.]*' "$f" | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' "$f" | head -1)"; done; }
# LINES_ONLY=1: the bench lines, phases and host splits only (2 minutes; r6u)
if [ -z "$LINES_ONLY" ]; then
timeout 1700 python -m pytest tests -m gpu -q > $D/tests.log 2>&1; echo "testsall rc=$?"; tail -n 2 $D/tests.log
fi
if [ -z "$LINES_ONLY" ] && [ -f virconv_amd/libvirconv_hip_exp.so ]; then
  VIRCONV_LIB=$R/virconv_amd/libvirconv_hip_exp.so timeout 900 python -m pytest tests/test_conv_v4_gpu.py tests/test_conv_pc_gpu.py tests/test_ops_gpu.py tests/test_round3_gpu.py tests/test_plan_gpu.py -m gpu -q > $D/experiments_suite.log 2>&1
  echo "experiments suite rc=$?"; tail -n 1 $D/experiments_suite.log
fi
[ -z "$LINES_ONLY" ] && { timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log; }
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_form.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_form_2.log 2>&1
line $D/bench_driver_form.log $D/bench_driver_form_2.log
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
timeout 300 python bench.py $B40 > $D/bench40.log 2>&1
VIRCONV_FLAT_PARAMS=0 timeout 300 python bench.py $B40 > $D/bench40_per_module_params.log 2>&1
VIRCONV_PLAN_GUARD_EARLY=0 timeout 300 python bench.py $B40 > $D/bench40_guard_at_forward_entry.log 2>&1
VIRCONV_FUSED_OPT=0 timeout 300 python bench.py $B40 > $D/bench40_stock_clip_adamw.log 2>&1
VIRCONV_DEBUG_SET=pass_dw_flush_mb=0 timeout 300 python bench.py $B40 > $D/bench40_one_reduce_launch.log 2>&1
timeout 300 python bench.py --model 8x $B40 > $D/bench_8x.log 2>&1
VIRCONV_DEBUG_SET=conv_nw8_below=0 timeout 300 python bench.py --model 8x $B40 > $D/bench_8x_four_wave_blocks.log 2>&1
VIRCONV_FLAT_PARAMS=0 timeout 300 python bench.py --model 8x $B40 > $D/bench_8x_per_module_params.log 2>&1
timeout 300 python bench.py --model 8x --mode infer --steps 30 --warmup 10 > $D/infer_8x_rot3.log 2>&1
timeout 300 python bench.py --mode infer --batch-size 1 > $D/infer_bs1.log 2>&1
timeout 300 python bench.py --mode infer --batch-size 4 > $D/infer_bs4.log 2>&1
timeout 300 python bench.py --frontend $B40 > $D/bench_frontend.log 2>&1
timeout 300 python bench.py --operand f16 $B40 > $D/bench_f16.log 2>&1
timeout 300 python bench.py --model 8x --operand f16 $B40 > $D/bench_8x_f16.log 2>&1
line $D/bench40.log $D/bench40_per_module_params.log $D/bench40_guard_at_forward_entry.log $D/bench40_stock_clip_adamw.log $D/bench40_one_reduce_launch.log $D/bench_8x.log $D/bench_8x_four_wave_blocks.log $D/bench_8x_per_module_params.log $D/infer_8x_rot3.log $D/infer_bs1.log $D/infer_bs4.log $D/bench_frontend.log $D/bench_f16.log $D/bench_8x_f16.log
timeout 200 python tools/step_phases.py > $D/phases.txt 2>&1; tail -n 3 $D/phases.txt
timeout 200 python tools/hostsplit.py 30 > $D/hostsplit.txt 2>&1; head -n 2 $D/hostsplit.txt
MODEL=8x timeout 200 python tools/hostsplit.py 30 > $D/hostsplit_8x.txt 2>&1; head -n 2 $D/hostsplit_8x.txt
[ -n "$LINES_ONLY" ] && { echo finished; exit 0; }
timeout 400 python tools/kbench.py > $D/kbench.txt 2>&1; tail -n 1 $D/kbench.txt
timeout 200 python tools/bevbench.py > $D/bevbench.txt 2>&1; tail -n 4 $D/bevbench.txt
VIRCONV_STRESS_STEPS=512 timeout 600 python -m pytest tests/test_plan_stress_gpu.py -q -k "checksums or inference" > $D/stress_512.log 2>&1; echo "plan stress 512 rc=$?"; tail -n 1 $D/stress_512.log
VIRCONV_SOAK_STEPS=256 timeout 600 python -m pytest tests/test_soak_gpu.py -q > $D/soak_256.log 2>&1; echo "soak 256 rc=$?"; tail -n 1 $D/soak_256.log
timeout 300 python tools/a17_lab.py micro --aggr step,step_exact,fwd:s3 --mode 0 2>/dev/null | grep '^{' > $D/a17_micro_final.jsonl; cat $D/a17_micro_final.jsonl | cut -c1-260
cd /tmp && export TMPDIR=/tmp
P="--steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 --exact-steps 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/prof -o x -- python $R/bench.py $P > $R/$D/prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$D/pmc_$c -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --family-steps 0 --exact-steps 0 > $R/$D/p_$c.log 2>&1
done
cd $R; f=$(find $D/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $D/kernel_stats.csv
f=$(find $D/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_gaps.py "$f" 4 --sequence > $D/sequence.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do f=$(find $D/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $D/pmc_$c.csv; done
find $D -name '*kernel_trace.csv' -delete; rm -rf $D/prof $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE
ls -la $D | head -50
echo finished
