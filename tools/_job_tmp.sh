cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5t; mkdir -p $D
for i in 1 2 3; do VIRCONV_PLAN_GUARD=0 timeout 300 python -m pytest tests/test_plan_stress_gpu.py -q -k "unsynchronised_train or inference" > $D/stress_$i.log 2>&1; echo "guard0 keep-live build run $i rc=$? $(tail -n 1 $D/stress_$i.log) $(grep -o '[0-9]* structures of [0-9]* \(steps\|frames\) differ' $D/stress_$i.log | tr '\n' ';')"; done
VIRCONV_STRESS_STEPS=512 VIRCONV_PLAN_GUARD=0 timeout 600 python -m pytest tests/test_plan_stress_gpu.py -q -k "checksums or inference" > $D/soak512_guard0.log 2>&1; echo "guard0 soak 512 rc=$? $(tail -n 1 $D/soak512_guard0.log) $(grep -o '[0-9]* structures of [0-9]* \(steps\|frames\) differ' $D/soak512_guard0.log | tr '\n' ';')"
timeout 300 python -m pytest tests/test_plan_gpu.py tests/test_plan_stress_gpu.py tests/test_ops_gpu.py -q -m gpu -k "plan or stress or project" > $D/plan.log 2>&1; echo "plan tests (guard on) rc=$? $(tail -n 1 $D/plan.log)"
echo finished
