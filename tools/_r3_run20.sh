#!/bin/bash
# round 3, GPU call 20: ticket part of the BatchNorm finish ahead of the output stores -- whole GPU suite, then on/off kernel stats
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3t
mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu > $O/pytest_all.log 2>&1
echo "gpu suite rc=$?" | tee -a $O/pytest_all.log
tail -5 $O/pytest_all.log
for v in 1 0 1 0; do
  VIRCONV_DEBUG_SET="conv_bn_finish=$v" timeout 300 python bench.py --no-cpu-baseline --family-steps 0 > $O/bench_fin${v}_$RANDOM.log 2>&1
done
for f in $O/bench_fin*.log; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $f)"; done
cd /tmp
for v in 1 0; do
  VIRCONV_DEBUG_SET="conv_bn_finish=$v" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats$v -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 > $R/$O/p_stats$v.log 2>&1
done
cd $R
for v in 1 0; do cp $(find $O/stats$v -name "*kernel_stats.csv" | head -1) $O/kernel_stats_fin$v.csv; done
find $O -name "*kernel_trace.csv" -delete
