cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5j; mkdir -p $D
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
line() { for f in "$@"; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' "$f" | head -1)"; done; }
VIRCONV_PLAN_GUARD=0 timeout 300 python -m pytest tests/test_plan_stress_gpu.py -q > $D/stress_guard0_agpr.log 2>&1; echo "stress guard0 (AGPR-form build) rc=$?"; tail -n 3 $D/stress_guard0_agpr.log | cut -c1-300
VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so VIRCONV_PLAN_GUARD=0 timeout 300 python -m pytest tests/test_plan_stress_gpu.py -q > $D/stress_guard0_vgpr.log 2>&1; echo "stress guard0 (VGPR-form build) rc=$?"; tail -n 3 $D/stress_guard0_vgpr.log | cut -c1-300
timeout 400 python tools/det_check.py --reps 40 --bs 2 base argflag > $D/det_agpr.txt 2>&1; grep "reps differ" $D/det_agpr.txt
VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so timeout 400 python tools/det_check.py --reps 40 --bs 2 base argflag > $D/det_vgpr.txt 2>&1; grep "reps differ" $D/det_vgpr.txt
timeout 600 python -m pytest tests/test_split_gpu.py tests/test_plan_gpu.py tests/test_plan_stress_gpu.py tests/test_golden_8x.py -m gpu -q > $D/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $D/tests.log | cut -c1-300
for i in 1 2; do timeout 200 python bench.py $B40 > $D/b40_agpr_$i.log 2>&1; VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so timeout 200 python bench.py $B40 > $D/b40_vgpr_$i.log 2>&1; done
timeout 300 python bench.py --model 8x $B40 > $D/bench_8x.log 2>&1
line $D/b40_*.log $D/bench_8x.log
timeout 400 python tools/kbench.py > $D/kbench.txt 2>&1; tail -n 1 $D/kbench.txt
echo finished
