mkdir -p gpurun_out/r2o
python bench.py --mode infer --batch-size 1 2>&1 | grep '"value"' | cut -c1-175 > gpurun_out/r2o/lines.txt
VIRCONV_NATIVE_PASS=0 python bench.py --mode infer --batch-size 1 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2o/lines.txt
python bench.py --mode infer --batch-size 4 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2o/lines.txt
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2o/lines.txt
VIRCONV_UNIT_OVERLAP_DW=0 python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2o/lines.txt
python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 >> gpurun_out/r2o/lines.txt
timeout 300 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "native or determin or eval or rulebook" 2>&1 | tail -n 2 >> gpurun_out/r2o/lines.txt
cat gpurun_out/r2o/lines.txt
