mkdir -p gpurun_out/r2r
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "8x or native" > gpurun_out/r2r/tests.log 2>&1
tail -n 12 gpurun_out/r2r/tests.log
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 > gpurun_out/r2r/lines.txt
VIRCONV_NATIVE_PASS=0 python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2r/lines.txt
python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2r/lines.txt
cat gpurun_out/r2r/lines.txt
