mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])" >> gpurun_out/ab/res.txt; }
run first_settle4 VIRCONV_SETTLE_SEC=4
run second_settle0 VIRCONV_SETTLE_SEC=0
run third_settle0 VIRCONV_SETTLE_SEC=0
cat gpurun_out/ab/res.txt
