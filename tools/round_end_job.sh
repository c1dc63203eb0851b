# The round-end measurement job of round 5 (gpurun calls r5z and, after LOG.md A.20, r5ae: profiles/r05_README.md): full GPU suite, smoke, the bench lines, phases, the plan soak.
#   gpurun --timeout 3400 -- bash tools/round_end_job.sh
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/${TAG:-r5ae}; mkdir -p $D
line() { for f in "$@"; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"value": [0-9.]*' "$f" | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' "$f" | head -1)"; done; }
timeout 1700 python -m pytest tests -m gpu -q > $D/tests.log 2>&1; echo "testsall rc=$?"; tail -n 2 $D/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_form.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_form_2.log 2>&1
line $D/bench_driver_form.log $D/bench_driver_form_2.log
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
timeout 300 python bench.py $B40 > $D/bench40.log 2>&1
timeout 300 python bench.py --model 8x $B40 > $D/bench_8x.log 2>&1
timeout 300 python bench.py --model 8x --mode infer --steps 30 --warmup 10 > $D/infer_8x_rot3.log 2>&1
timeout 300 python bench.py --mode infer --batch-size 1 > $D/infer_bs1.log 2>&1
timeout 300 python bench.py --mode infer --batch-size 4 > $D/infer_bs4.log 2>&1
timeout 300 python bench.py --frontend $B40 > $D/bench_frontend.log 2>&1
timeout 300 python bench.py --operand f16 $B40 > $D/bench_f16.log 2>&1
timeout 300 python bench.py --model 8x --operand f16 $B40 > $D/bench_8x_f16.log 2>&1
line $D/bench40.log $D/bench_8x.log $D/infer_8x_rot3.log $D/infer_bs1.log $D/infer_bs4.log $D/bench_frontend.log $D/bench_f16.log $D/bench_8x_f16.log
timeout 200 python tools/step_phases.py > $D/phases.txt 2>&1; tail -n 3 $D/phases.txt
VIRCONV_STRESS_STEPS=512 timeout 600 python -m pytest tests/test_plan_stress_gpu.py -q -k "checksums or inference" > $D/soak_512.log 2>&1; echo "soak 512 rc=$?"; tail -n 2 $D/soak_512.log
VIRCONV_STRESS_STEPS=64 VIRCONV_PLAN_GUARD=0 timeout 300 python -m pytest tests/test_plan_stress_gpu.py -q > $D/stress_guard0.log 2>&1; echo "stress guard0 rc=$? (expected to fail)"; grep -o "[0-9]* structures of [0-9]* \(steps\|frames\) differ" $D/stress_guard0.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 --exact-steps 0 > $GRAFT_REPO_ROOT/$D/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $D/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $D/kernel_stats.csv; rm -rf $D/prof
timeout 200 python tools/hostsplit.py 30 > $D/hostsplit.txt 2>&1; head -n 2 $D/hostsplit.txt
echo finished
