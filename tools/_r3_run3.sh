# round 3, GPU call 3: f3 tests, fixed tests, bench A/B (dW split size, side-stream priority), BEV stem bench
D=gpurun_out/r3c
mkdir -p $D
R=$PWD
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_bev_stem.py -x -q -m gpu > $D/t_new.log 2>&1; echo "new tests rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "reduced_operand or dense or post_act" > $D/t_ops.log 2>&1; echo "ops rc=$?"
tail -n 4 $D/t_new.log; tail -n 4 $D/t_ops.log
python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/bench.log 2>&1
VIRCONV_DEBUG_SET=bw_rows_per_split=2048 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_bw2048.log 2>&1
VIRCONV_DEBUG_SET=bw_rows_per_split=4096 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_bw4096.log 2>&1
VIRCONV_SIDE_PRIORITY=-1 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_sideprio_hi.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_again.log 2>&1
VIRCONV_DEBUG_SET=bw_rows_per_split=4096 timeout 200 python tools/kbench.py --only dw > $D/kbench_dw4096.txt 2>&1
timeout 200 python tools/bevbench.py > $D/bevbench.txt 2>&1
grep -h ms_per_step $D/*.log | cut -c1-160
for f in $D/bench*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
cat $D/bevbench.txt | tail -5
echo finished
