# round 3, GPU call 1: new kernels' tests, the whole GPU suite, bench A/B (row order), kernel table, rocprof stats
D=gpurun_out/r3a
mkdir -p $D
R=$PWD
timeout 600 python -m pytest tests/test_round3_gpu.py -x -q > $D/t_round3.log 2>&1; echo "round3 tests rc=$?"
timeout 900 python -m pytest tests -m gpu -x -q > $D/t_all.log 2>&1; echo "all gpu tests rc=$?"
tail -3 $D/t_round3.log $D/t_all.log
python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/bench.log 2>&1
VIRCONV_ROW_ORDER=bwd python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_roworder_bwd.log 2>&1
python bench.py --mode infer --batch-size 1 > $D/infer_bs1.log 2>&1
timeout 200 python tools/kbench.py > $D/kbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 > $R/$D/p_stats.log 2>&1
cd $R
python tools/trace_gaps.py $(find $D/stats -name "*kernel_trace.csv" | head -1) > $D/gaps.txt 2>&1
find $D -name "*kernel_trace.csv" -delete
grep -h ms_per_step $D/*.log | cut -c1-400
echo finished
