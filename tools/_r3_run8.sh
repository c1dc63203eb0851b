D=gpurun_out/r3h
mkdir -p $D
timeout 600 python -m pytest tests/test_fullsize_fixture.py -x -q -m gpu > $D/t_fix.log 2>&1; echo "fixture gpu rc=$?"; tail -n 2 $D/t_fix.log
timeout 120 python tools/kbench.py --only bwd --layers d2 > $D/kb_rep0.txt 2>&1
VIRCONV_REP_FIRST_ORDER=1 timeout 120 python tools/kbench.py --only bwd --layers d2 > $D/kb_rep1.txt 2>&1
grep "2D" $D/kb_rep0.txt; grep "2D" $D/kb_rep1.txt
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_a.log 2>&1
VIRCONV_REP_FIRST_ORDER=1 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_rep1.log 2>&1
VIRCONV_PLAN_PRIORITY=0 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_planprio0.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_b.log 2>&1
VIRCONV_REP_FIRST_ORDER=1 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_rep1b.log 2>&1
for f in $D/bench*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
echo finished
