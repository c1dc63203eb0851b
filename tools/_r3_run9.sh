D=gpurun_out/r3i
mkdir -p $D
timeout 600 python -m pytest tests/test_round3_gpu.py -x -q -m gpu -k "weight_gradient_v2" > $D/t_bw2.log 2>&1; echo "bw v2 tests rc=$?"; tail -n 3 $D/t_bw2.log
timeout 200 python tools/kbench.py --only dw > $D/kb_dw_v1.txt 2>&1
VIRCONV_DEBUG_SET=bw_variant=2 timeout 200 python tools/kbench.py --only dw > $D/kb_dw_v2.txt 2>&1
paste <(tail -n 18 $D/kb_dw_v1.txt | cut -c1-60,118-150) <(tail -n 18 $D/kb_dw_v2.txt | cut -c118-150)
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_v1.log 2>&1
VIRCONV_DEBUG_SET=bw_variant=2 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_v2.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_v1b.log 2>&1
VIRCONV_DEBUG_SET=bw_variant=2 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_v2b.log 2>&1
for f in $D/bench*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
echo finished
