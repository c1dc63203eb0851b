mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])" >> gpurun_out/ab/res.txt; }
run base1 X=1
run nooverlap VIRCONV_OVERLAP_DW=0
run q2 GPU_MAX_HW_QUEUES=2
run q4 GPU_MAX_HW_QUEUES=4
run base2 X=1
run roworder_none VIRCONV_ROW_ORDER=none
run roworder_all VIRCONV_ROW_ORDER=all
run win1024 VIRCONV_ROW_ORDER_WINDOW=1024
run win4096 VIRCONV_ROW_ORDER_WINDOW=4096
run base3 X=1
cat gpurun_out/ab/res.txt
