mkdir -p gpurun_out/r2t
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "channel_counts" > gpurun_out/r2t/tests2.log 2>&1
tail -n 25 gpurun_out/r2t/tests2.log
