#!/bin/bash
# Texture-addresser / L1 / issue counters of the gather-GEMM variants on one layer (developer tool; run on the GPU box):
#   gpurun --timeout 600 -- 'bash tools/pmc_ta.sh gpurun_out/pmc_ta s3.d3_conv1'
# At most FOUR TA / TCP counters per pass: asking for seven at once aborts rocprofv3 (signal 6) and the run then sits in its
# timeout.  Counters only (--kernel-trace + --pmc, no tracing domains).  tools/pmc_conv_summary.py prints the per-kernel means.
out=${1:-gpurun_out/pmc_ta}; layers=${2:-s3.d3_conv1}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$out; cd /tmp; export TMPDIR=/tmp
P1="TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P2="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
P3="SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"
for var in "v2_canonical:" "v2_image:--autopack" "v2_image_dxs:--autopack --dxs 1" "v4_image:--autopack --v4 1" "v5_image:--autopack --v5 1"; do
  name=${var%%:*}; flags=${var#*:}
  i=0
  for pmc in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 45 rocprofv3 --kernel-trace --pmc $pmc -d $R/$out/${name}_p$i -o r --output-format csv -- \
      python $R/tools/kbench.py --layers $layers --only fwd --iters 5 $flags > $R/$out/${name}_p$i.log 2>&1 \
      || echo "pass $name p$i failed (see $out/${name}_p$i.log)"
    find $R/$out/${name}_p$i -name "*kernel_trace.csv" -delete 2>/dev/null
  done
done
echo finished
