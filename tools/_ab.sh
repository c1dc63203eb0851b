mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])" >> gpurun_out/ab/res.txt; }
run warm X=1
run overlap1 VIRCONV_OVERLAP_DW=1
run nooverlap1 VIRCONV_OVERLAP_DW=0
run overlap2 VIRCONV_OVERLAP_DW=1
run nooverlap2 VIRCONV_OVERLAP_DW=0
run overlap3 VIRCONV_OVERLAP_DW=1
run nooverlap3 VIRCONV_OVERLAP_DW=0
cat gpurun_out/ab/res.txt
