# Round-3 measurement run on the GPU box: whole GPU suite, bench line(s), rocprofv3 --stats, PMC passes (each in its own run),
# kernel table, MFMA-busy counters of the traced kernel.  Raw output under gpurun_out/$1; tools/make_profiles.py --tag r03 turns
# it into the summaries committed under profiles/.
D=gpurun_out/${1:-r3p}
mkdir -p $D
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x > $D/t_all.log 2>&1; echo "gpu suite rc=$?"; tail -n 3 $D/t_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
python bench.py > $D/bench_full.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline > $D/b2.log 2>&1
VIRCONV_FORCE_DDP=1 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_force_ddp.log 2>&1
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_8x.log 2>&1
python bench.py --frontend --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_frontend.log 2>&1
python bench.py --operand f16 --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_f16.log 2>&1
python bench.py --model 8x --operand f16 --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_8x_f16.log 2>&1
python bench.py --mode infer --batch-size 1 > $D/infer_bs1.log 2>&1
python bench.py --mode infer --batch-size 4 > $D/infer_bs4.log 2>&1
VIRCONV_NATIVE_PASS=0 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_node_path.log 2>&1
VIRCONV_ROW_ORDER=strided python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_roworder_strided.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/stats -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 > $R/$D/p_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$D/fetch -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --family-steps 0 > $R/$D/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$D/write -o x -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --family-steps 0 > $R/$D/p_write.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $R/$D/mfma -o r --output-format csv -- python $R/tools/kbench.py --layers s3.d3_conv1,s3.down --only fwd --iters 5 --autopack > $R/$D/p_mfma.log 2>&1
cd $R
python tools/trace_gaps.py $(find $D/stats -name "*kernel_trace.csv" | head -1) > $D/gaps.txt 2>&1
find $D -name "*kernel_trace.csv" -delete
timeout 200 python tools/kbench.py > $D/kbench.txt 2>&1
timeout 100 python tools/step_phases.py > $D/phases.txt 2>&1
timeout 100 python tools/bevbench.py > $D/bevbench.txt 2>&1
for f in $D/bench*.log $D/b2.log $D/infer*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f) $(grep -o '"value": [0-9.]*' $f | head -1); done
echo finished
