mkdir -p gpurun_out/ab
run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['config'].get('cpu_affinity'))" >> gpurun_out/ab/res.txt; }
python -c "import os; a=sorted(os.sched_getaffinity(0)); print('affinity', len(a), a[:4], a[-4:])" >> gpurun_out/ab/res.txt
run first_bind VIRCONV_NUMA_BIND=1
run second_nobind VIRCONV_NUMA_BIND=0
run third_bind VIRCONV_NUMA_BIND=1
run fourth_nobind VIRCONV_NUMA_BIND=0
run fifth_bind VIRCONV_NUMA_BIND=1
cat gpurun_out/ab/res.txt
