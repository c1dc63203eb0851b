#!/bin/bash
# PMC comparison of the direct (v2) and the LDS-window (v3) gather-GEMM on one layer (developer tool; run on the GPU box).
# usage: tools/pmc_conv.sh <out_dir> "<kbench layer filter>"
out=${1:-gpurun_out/pmc_conv}; layers=${2:-s3.d3_conv1}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
for var in "nowin:--no-window" "win:" "win24:--winrows 24"; do
  name=${var%%:*}; flags=${var#*:}
  i=0
  for pmc in "$P1" "$P2"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc -d $R/$out/${name}_p$i -o r --output-format csv -- \
      python $R/tools/kbench.py --layers $layers --only fwd --iters 5 $flags > $R/$out/${name}_p$i.log 2>&1
  done
done
