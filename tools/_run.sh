mkdir -p gpurun_out/r2q
R=$PWD
for i in 1 2; do
VIRCONV_PASS_BWD_EPILOGUE=0 python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 | sed 's/^/epi0 /' >> gpurun_out/r2q/lines.txt
VIRCONV_PASS_BWD_EPILOGUE=1 python bench.py --steps 40 --warmup 12 --no-cpu-baseline 2>&1 | grep '"value"' | cut -c1-175 | sed 's/^/epi1 /' >> gpurun_out/r2q/lines.txt
done
cat gpurun_out/r2q/lines.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/stats1 -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2q/p1.log 2>&1
VIRCONV_PASS_BWD_EPILOGUE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2q/stats0 -o x -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r2q/p0.log 2>&1
cd $R
find gpurun_out/r2q -name "*kernel_trace.csv" -delete
echo finished
