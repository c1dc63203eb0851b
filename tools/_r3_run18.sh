#!/bin/bash
# round 3, GPU call 18: fused weighted-sum loss op -- tests, train step, kernel list of the torch side
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3r
mkdir -p $O
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_conv_pc_gpu.py tests/test_bn_finish_gpu.py -x -q -m gpu > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_$i.log 2>&1
  echo "bench $i: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.log)"
done
timeout 300 python tools/step_phases.py > $O/step_phases.txt 2>&1; tail -1 $O/step_phases.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 > $R/$O/p_stats.log 2>&1
cd $R
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
grep -E "weighted_sum|reduce_kernel|MulFunctor|rocblas_dot|direct_copy" $O/kernel_stats.csv | cut -c1-140
