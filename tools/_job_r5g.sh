cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5g; mkdir -p $D
timeout 600 python -m pytest tests/test_split_gpu.py tests/test_plan_gpu.py tests/test_plan_stress_gpu.py -m gpu -q > $D/tests.log 2>&1; echo "split+plan+stress rc=$?"; tail -n 3 $D/tests.log | cut -c1-300
timeout 400 python tools/kbench.py > $D/kbench_dot2.txt 2>&1; echo "kbench rc=$?"; tail -n 1 $D/kbench_dot2.txt
VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so timeout 400 python tools/kbench.py > $D/kbench_shiftmask.txt 2>&1; echo "kbench(ab) rc=$?"; tail -n 1 $D/kbench_shiftmask.txt
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
for i in 1 2; do timeout 200 python bench.py $B40 > $D/bench40_dot2_$i.log 2>&1; VIRCONV_LIB=$PWD/virconv_amd/libvirconv_ab.so timeout 200 python bench.py $B40 > $D/bench40_ab_$i.log 2>&1; done
for f in $D/bench40_*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' $f | head -1)"; done
timeout 300 python tools/hostprof.py > $D/hostprof.txt 2>&1; head -1 $D/hostprof.txt
echo finished
