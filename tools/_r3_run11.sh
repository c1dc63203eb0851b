D=gpurun_out/r3k
mkdir -p $D
python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver_form.log 2>&1
python bench.py --mode infer --batch-size 1 > $D/infer_bs1.log 2>&1
python bench.py --mode infer --batch-size 4 > $D/infer_bs4.log 2>&1
for f in $D/*.log; do echo $f; tail -n 1 $f | cut -c1-700; done
echo finished
