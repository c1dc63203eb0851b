cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5m; mkdir -p $D
timeout 400 python -m pytest tests/test_plan_stress_gpu.py -q -k "front_end" > $D/frontend_stress.log 2>&1; echo "frontend stress rc=$?"; tail -n 3 $D/frontend_stress.log | cut -c1-600
VIRCONV_STRESS_STEPS=256 timeout 400 python -m pytest tests/test_plan_stress_gpu.py -q -k "front_end" > $D/frontend_stress256.log 2>&1; echo "frontend stress 256 rc=$?"; tail -n 3 $D/frontend_stress256.log | cut -c1-600
timeout 300 python -m pytest tests/test_plan_gpu.py tests/test_plan_stress_gpu.py -q -m gpu > $D/plan.log 2>&1; echo "plan rc=$?"; tail -n 2 $D/plan.log | cut -c1-300
echo finished
