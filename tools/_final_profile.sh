mkdir -p gpurun_out/f1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/f1/t.log 2>&1
python bench.py > gpurun_out/f1/bench_full.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline > gpurun_out/f1/b2.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f1/stats -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/f1/p_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/f1/fetch -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/f1/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/f1/write -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/f1/p_write.log 2>&1
cd $R
find gpurun_out/f1 -name "*kernel_trace.csv" -delete
timeout 100 python tools/kbench.py > gpurun_out/f1/kbench.txt 2>&1
tail -n 3 gpurun_out/f1/t.log
