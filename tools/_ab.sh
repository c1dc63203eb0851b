mkdir -p gpurun_out/ab
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "row_order" > gpurun_out/ab/t.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('a', d['ms_per_step'])" >> gpurun_out/ab/res.txt
timeout 60 python tools/kbench.py --only rulebook 2>/dev/null | tail -13 >> gpurun_out/ab/res.txt
tail -2 gpurun_out/ab/t.log; cat gpurun_out/ab/res.txt
