mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2n/tests.log 2>&1
tail -n 4 gpurun_out/r2n/tests.log
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | grep ms_per_step > gpurun_out/r2n/bench.txt
timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | grep ms_per_step >> gpurun_out/r2n/bench.txt
cut -c1-170 gpurun_out/r2n/bench.txt
timeout 100 python tools/step_phases.py > gpurun_out/r2n/phases.txt 2>&1; tail -n 1 gpurun_out/r2n/phases.txt
echo finished
