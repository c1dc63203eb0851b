#!/bin/bash
# round 3, GPU call 13: per-kernel durations with and without the in-kernel BatchNorm finish (rocprofv3 --stats, same box)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/r3m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  VIRCONV_DEBUG_SET="conv_bn_finish=$v" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats$v -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 > $R/$O/p_stats$v.log 2>&1
done
cd $R
for v in 1 0; do
  f=$(find $O/stats$v -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/kernel_stats_fin$v.csv
  python tools/trace_gaps.py $(find $O/stats$v -name "*kernel_trace.csv" | head -1) > $O/gaps_fin$v.txt 2>&1
done
find $O -name "*kernel_trace.csv" -delete
for v in 1 0 1 0; do
  VIRCONV_DEBUG_SET="conv_bn_finish=$v" timeout 300 python tools/step_phases.py > $O/phases_fin${v}_$RANDOM.txt 2>&1
done
tail -n 1 $O/phases_fin*.txt
