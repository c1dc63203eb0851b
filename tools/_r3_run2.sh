# round 3, GPU call 2: changed tests, bench, kernel table, rocprof stats
D=gpurun_out/r3b
mkdir -p $D
R=$PWD
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_fullsize_fixture.py -x -q -m gpu > $D/t_new.log 2>&1; echo "new tests rc=$?"
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s > $D/t_fullsize.log 2>&1; echo "fullsize rc=$?"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_conv_v4_gpu.py -q -m gpu > $D/t_ops.log 2>&1; echo "ops rc=$?"
tail -n 5 $D/t_new.log; tail -n 5 $D/t_fullsize.log; tail -n 8 $D/t_ops.log
python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/bench.log 2>&1
VIRCONV_ROW_ORDER=bwd python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_roworder_bwd.log 2>&1
timeout 200 python tools/kbench.py > $D/kbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 > $R/$D/p_stats.log 2>&1
cd $R
python tools/trace_gaps.py $(find $D/stats -name "*kernel_trace.csv" | head -1) > $D/gaps.txt 2>&1
find $D -name "*kernel_trace.csv" -delete
grep -h ms_per_step $D/*.log | cut -c1-300
echo finished
