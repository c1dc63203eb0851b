mkdir -p gpurun_out/ab
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "row_order" > gpurun_out/ab/t.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ab/stats -o x -- python /root/repo/bench.py --steps 10 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/ab/p.log 2>&1
cd /root/repo; find gpurun_out/ab -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/ab/t.log; grep -E "row_order|dense_kernel<true>" gpurun_out/ab/stats/x_kernel_stats.csv | cut -c1-120
