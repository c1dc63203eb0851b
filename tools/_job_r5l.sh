cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r5l; mkdir -p $D
B40="--steps 40 --warmup 12 --no-cpu-baseline --family-steps 0 --exact-steps 0"
line() { for f in "$@"; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' "$f" | head -1) $(grep -o '"host_enqueue_ms_per_step": [0-9.]*' "$f" | head -1)"; done; }
for i in 1 2 3; do for g in 1 0; do VIRCONV_PLAN_GUARD=$g timeout 200 python bench.py --model 8x $B40 > $D/b8x_guard${g}_$i.log 2>&1; done; done
line $D/b8x_*.log
MODEL=8x timeout 200 python tools/hostprof.py > $D/hostprof_8x.txt 2>&1; sed -n 2,30p $D/hostprof_8x.txt | cut -c1-150
echo finished
