mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > gpurun_out/ab/t.log 2>&1
run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])" >> gpurun_out/ab/res.txt; }
run a X=1
run b X=1
timeout 100 python tools/kbench.py --only fwd 2>/dev/null | grep -E "down|conv_out" | cut -c1-60,150-190 >> gpurun_out/ab/res.txt
tail -3 gpurun_out/ab/t.log; cat gpurun_out/ab/res.txt
