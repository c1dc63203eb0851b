cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5f; mkdir -p $D
timeout 600 python -m pytest tests/test_plan_gpu.py tests/test_plan_stress_gpu.py -m gpu -q > $D/plan_tests.log 2>&1; echo "plan+stress rc=$?"; tail -n 6 $D/plan_tests.log | cut -c1-400
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
for g in 1 0 1 0; do VIRCONV_PLAN_GUARD=$g timeout 200 python bench.py $B40 > $D/bench40_guard${g}_$RANDOM.log 2>&1; done
for f in $D/bench40_guard*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $f | head -1)"; done
timeout 300 python bench.py --model 8x $B40 > $D/bench8x.log 2>&1; echo "8x $(grep -o '"ms_per_step": [0-9.]*' $D/bench8x.log | head -1)"
timeout 300 python bench.py --mode infer --batch-size 1 > $D/infer1.log 2>&1; echo "infer1 $(grep -o '"ms_per_step": [0-9.]*' $D/infer1.log | head -1)"
VIRCONV_PLAN_GUARD=0 timeout 300 python bench.py --mode infer --batch-size 1 > $D/infer1_g0.log 2>&1; echo "infer1 guard0 $(grep -o '"ms_per_step": [0-9.]*' $D/infer1_g0.log | head -1)"
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_golden_8x.py tests/test_ops_gpu.py -m gpu -q -x > $D/more_tests.log 2>&1; echo "more tests rc=$?"; tail -n 3 $D/more_tests.log | cut -c1-300
echo finished
