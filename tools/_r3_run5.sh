D=gpurun_out/r3e
mkdir -p $D
timeout 600 python -m pytest tests/test_round3_gpu.py -x -q -m gpu -k "interleaved" > $D/t_il.log 2>&1; echo "il test rc=$?"; tail -n 3 $D/t_il.log
timeout 300 python tools/kbench.py --only fwd --il > $D/kbench_il.txt 2>&1
tail -n 19 $D/kbench_il.txt
echo finished
