#!/bin/bash
# One parameterised GPU job script (replaces the 33 one-shot tools/_r3_run*.sh of round 3).
#
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <tag> <step> [<step> ...]'
#
# Raw output goes to gpurun_out/<tag>/ (scratch, merged back by gpurun); tools/make_profiles.py turns the rocprof output into the
# summaries committed under profiles/.  Steps (each is bounded by its own timeout so that a hang cannot eat the call):
#   tests[:<pytest -k expr>]  GPU test-suite, -x (or the selected tests); testsall: without -x    smoke      __graft_entry__.smoke()
#   bench                     python bench.py --gpus 1 --steps 20 --warmup 5  bench40    40 steps / 12 warm-up, no CPU baseline
#   benchenv:<K=V,...>        bench40 with environment variables set (A/B)    bench8x | benchf16 | benchfront | infer1 | infer4
#   hostprof | phases | kbench[:<args>] | bevbench | gaps
#   stats                     rocprofv3 --kernel-trace --stats of the bench command (+ trace_gaps)
#   pmc                       FETCH_SIZE / WRITE_SIZE passes of the bench command (separate runs, no other tracing)
#   mfma | mfmadw             MFMA-busy counters of the traced kernel and a strided forward | of two weight-gradient launches (kbench)
#   py:<file>[:<args>]        python <file> <args>
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R" || exit 1
TAG="${1:-job}"; shift
D="gpurun_out/$TAG"; mkdir -p "$D"
BENCH40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
line() { for f in "$@"; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"value": [0-9.]*' "$f" | head -1)"; done; }
for step in "$@"; do
  name="${step%%:*}"; arg=""; [ "$step" != "$name" ] && arg="${step#*:}"
  echo "== $step"
  case "$name" in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$arg" > "$D/tests_sel.log" 2>&1; echo "rc=$?"; tail -n 4 "$D/tests_sel.log"
      else timeout 1700 python -m pytest tests -m gpu -q -x > "$D/tests.log" 2>&1; echo "rc=$?"; tail -n 4 "$D/tests.log"; fi ;;
    testsall) timeout 1700 python -m pytest tests -m gpu -q > "$D/tests.log" 2>&1; echo "rc=$?"; tail -n 6 "$D/tests.log" ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$D/smoke.log" 2>&1; tail -n 2 "$D/smoke.log" ;;
    bench) timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$D/bench_driver_form.log" 2>&1; line "$D/bench_driver_form.log" ;;
    bench40) timeout 300 python bench.py $BENCH40 > "$D/bench40.log" 2>&1; line "$D/bench40.log" ;;
    benchenv)
      f="$D/bench40_$(echo "$arg" | tr -c 'A-Za-z0-9_=\n' '_').log"
      ( for kv in $(echo "$arg" | tr ',' ' '); do export "$kv"; done; timeout 300 python bench.py $BENCH40 > "$f" 2>&1 ); line "$f" ;;
    bench8x) timeout 300 python bench.py --model 8x $BENCH40 > "$D/bench_8x.log" 2>&1; line "$D/bench_8x.log" ;;
    benchf16) timeout 300 python bench.py --operand f16 $BENCH40 > "$D/bench_f16.log" 2>&1; line "$D/bench_f16.log" ;;
    benchfront) timeout 300 python bench.py --frontend $BENCH40 > "$D/bench_frontend.log" 2>&1; line "$D/bench_frontend.log" ;;
    infer1) timeout 300 python bench.py --mode infer --batch-size 1 > "$D/infer_bs1.log" 2>&1; line "$D/infer_bs1.log" ;;
    infer4) timeout 300 python bench.py --mode infer --batch-size 4 > "$D/infer_bs4.log" 2>&1; line "$D/infer_bs4.log" ;;
    hostprof) timeout 200 python tools/hostprof.py > "$D/hostprof.txt" 2>&1; grep -m1 enqueue "$D/hostprof.txt" ;;
    phases) timeout 200 python tools/step_phases.py > "$D/phases.txt" 2>&1; tail -n 2 "$D/phases.txt" ;;
    kbench) timeout 400 python tools/kbench.py $arg > "$D/kbench$(echo "$arg" | tr -c 'A-Za-z0-9\n' '_').txt" 2>&1; echo "rc=$?" ;;
    bevbench) timeout 200 python tools/bevbench.py > "$D/bevbench.txt" 2>&1; tail -n 6 "$D/bevbench.txt" ;;
    stats)
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$D/stats" -o x -- \
          python "$R/bench.py" --steps 10 --warmup 5 --no-cpu-baseline --family-steps 0 --exact-steps 0 > "$R/$D/p_stats.log" 2>&1 )
      python tools/trace_gaps.py "$(find "$D/stats" -name '*kernel_trace.csv' | head -1)" > "$D/gaps.txt" 2>&1
      find "$D" -name '*kernel_trace.csv' -size +20M -delete; line "$D/p_stats.log" ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$D/pmc_$c" -o x -- \
            python "$R/bench.py" --steps 5 --warmup 3 --no-cpu-baseline --family-steps 0 --exact-steps 0 > "$R/$D/p_$c.log" 2>&1 ); done
      echo done ;;
    mfma)
      ( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
          SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d "$R/$D/mfma" -o r --output-format csv -- \
          python "$R/tools/kbench.py" --layers s3.d3_conv1,s3.down --only fwd --iters 5 --autopack ${arg} > "$R/$D/p_mfma.log" 2>&1 ); echo done ;;
    mfmadw)
      ( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
          SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d "$R/$D/mfmadw" -o r --output-format csv -- \
          python "$R/tools/kbench.py" --layers s3.d3_conv1,s3.d3_conv2 --only dw --iters 5 ${arg} > "$R/$D/p_mfmadw.log" 2>&1 )
      ( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE \
          -d "$R/$D/tadw" -o r --output-format csv -- \
          python "$R/tools/kbench.py" --layers s3.d3_conv1,s3.d3_conv2 --only dw --iters 5 ${arg} > "$R/$D/p_tadw.log" 2>&1 )
      find "$D" -name '*kernel_trace.csv' -size +20M -delete; echo done ;;
    py) f="${arg%%:*}"; a=""; [ "$arg" != "$f" ] && a="${arg#*:}"; timeout 600 python "$f" $a > "$D/$(basename "$f" .py).txt" 2>&1; echo "rc=$?"; tail -n 5 "$D/$(basename "$f" .py).txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
echo finished
