cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5n; mkdir -p $D
for i in 1 2; do
VIRCONV_PLAN_GUARD=0 VIRCONV_DEBUG_SET=uv_first_lane=1 timeout 300 python -m pytest tests/test_plan_stress_gpu.py -q -k "unsynchronised_train or inference" > $D/stress_g0_firstlane_$i.log 2>&1; echo "guard0 first-lane run $i rc=$?"; tail -n 1 $D/stress_g0_firstlane_$i.log; grep -o "[0-9]* structures of [0-9]* \(steps\|frames\) differ" $D/stress_g0_firstlane_$i.log
done
VIRCONV_PLAN_GUARD=0 timeout 300 python -m pytest tests/test_plan_stress_gpu.py -q -k "unsynchronised_train or inference" > $D/stress_g0_control.log 2>&1; echo "guard0 control rc=$?"; tail -n 1 $D/stress_g0_control.log; grep -o "[0-9]* structures of [0-9]* \(steps\|frames\) differ" $D/stress_g0_control.log
timeout 300 python -m pytest tests/test_plan_gpu.py tests/test_plan_stress_gpu.py -q -m gpu > $D/plan.log 2>&1; echo "plan (guard on) rc=$?"; tail -n 1 $D/plan.log
echo finished
