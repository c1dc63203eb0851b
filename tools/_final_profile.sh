# Round-end measurement run on the GPU box: bench line(s), rocprofv3 --stats, PMC passes (each in its own run), kernel table.
# Raw output under gpurun_out/f2/; tools/make_profiles.py turns it into the summaries committed under profiles/.
D=gpurun_out/f4
mkdir -p $D
R=$PWD
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1
python bench.py > $D/bench_full.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/b2.log 2>&1
VIRCONV_FORCE_DDP=1 python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/bench_force_ddp.log 2>&1
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline > $D/bench_8x.log 2>&1
python bench.py --frontend --steps 40 --warmup 12 --no-cpu-baseline > $D/bench_frontend.log 2>&1
python bench.py --mode infer --batch-size 1 > $D/infer_bs1.log 2>&1
python bench.py --mode infer --batch-size 4 > $D/infer_bs4.log 2>&1
VIRCONV_NATIVE_PASS=0 python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/bench_node_path.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/$D/p_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$D/fetch -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/$D/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$D/write -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/$D/p_write.log 2>&1
cd $R
python tools/trace_gaps.py $(find $D/stats -name "*kernel_trace.csv" | head -1) > $D/gaps.txt 2>&1
find $D -name "*kernel_trace.csv" -delete
timeout 150 python tools/kbench.py > $D/kbench.txt 2>&1
timeout 100 python tools/step_phases.py > $D/phases.txt 2>&1
timeout 60 python tools/nmsbench.py > $D/nmsbench.txt 2>&1
(tools/ubench/gather_ubench; tools/ubench/gather_ubench 75991) > $D/gather_ubench.txt 2>&1
grep -h ms_per_step $D/*.log | cut -c1-200
echo finished
