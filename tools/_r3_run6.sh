D=gpurun_out/r3f
mkdir -p $D
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "native_feature_pass or is_deterministic or bit" > $D/t_pass.log 2>&1; echo "pass tests rc=$?"; tail -n 3 $D/t_pass.log
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_a.log 2>&1
VIRCONV_DEBUG_SET=pass_defer_dw_reduce=0 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_nodefer.log 2>&1
python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_b.log 2>&1
VIRCONV_DEBUG_SET=pass_defer_dw_reduce=0 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_nodefer2.log 2>&1
python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_8x.log 2>&1
VIRCONV_DEBUG_SET=pass_defer_dw_reduce=0 python bench.py --model 8x --steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 > $D/bench_8x_nodefer.log 2>&1
for f in $D/bench*.log; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
echo finished
