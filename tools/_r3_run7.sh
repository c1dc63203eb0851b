D=gpurun_out/r3g
mkdir -p $D
timeout 900 python -m pytest tests/test_fullsize_fixture.py -x -q -m gpu > $D/t_fix.log 2>&1; echo "fixture gpu rc=$?"; tail -n 3 $D/t_fix.log
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_a.log 2>&1
GPU_MAX_HW_QUEUES=4 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_q4.log 2>&1
VIRCONV_PASS_DW_MAIN_TAIL=3 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_tail3.log 2>&1
VIRCONV_DEBUG_SET=conv_v4=2 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_v4.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_b.log 2>&1
for f in $D/bench*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
echo finished
