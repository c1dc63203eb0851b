mkdir -p gpurun_out/r2m
VIRCONV_CONV_NW=8 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv or subm or strided or post_act or epilogue" > gpurun_out/r2m/tests_nw8.log 2>&1
tail -n 3 gpurun_out/r2m/tests_nw8.log
timeout 200 python tools/kbench.py --only fwd --nw 4 > gpurun_out/r2m/kb_fwd_nw4.log 2>&1
timeout 200 python tools/kbench.py --only fwd --nw 8 > gpurun_out/r2m/kb_fwd_nw8.log 2>&1
timeout 200 python tools/kbench.py --only bwd --nw 4 > gpurun_out/r2m/kb_bwd_nw4.log 2>&1
timeout 200 python tools/kbench.py --only bwd --nw 8 > gpurun_out/r2m/kb_bwd_nw8.log 2>&1
grep -h TOTAL gpurun_out/r2m/kb_*.log
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | grep ms_per_step | cut -c1-170 > gpurun_out/r2m/bench_nw.txt
VIRCONV_CONV_NW=8 timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | grep ms_per_step | cut -c1-170 >> gpurun_out/r2m/bench_nw.txt
cat gpurun_out/r2m/bench_nw.txt
echo finished
