#!/bin/bash
# round 3, GPU call 16: ablations of the pair-compacted kernel (which part of an item costs the microseconds)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
for a in 0 1 2 4 8 3 7 15; do
  VIRCONV_DEBUG_SET="conv_pc=1,conv_pc_ablate=$a" timeout 200 python tools/kbench.py --layers s3.down,s2.down --only fwd --autopack --iters 10 > $O/kbench_abl$a.txt 2>&1
  echo "ablate=$a: $(grep -E 's3.down|s2.down' $O/kbench_abl$a.txt | tail -2 | awk '{print $1, $7}' | tr '\n' ' ')"
done
cd /tmp
VIRCONV_DEBUG_SET="conv_pc=1" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o x -- python $R/tools/kbench.py --layers s3.down --only fwd --autopack --iters 10 > $R/$O/p.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
