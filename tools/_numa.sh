mkdir -p gpurun_out/numa
{
echo "== numa nodes"; ls /sys/devices/system/node/ | grep node | tr '\n' ' '; echo
for d in /sys/class/drm/card*/device; do echo "$d numa=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done
echo "== kfd"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n $(grep -E 'cpu_cores_count|simd_count|drm_render_minor' $n/properties | tr '\n' ' ')"; done
echo "== affinity now"; python -c "import os; print(len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], '...')"
nproc; lscpu | grep -E "NUMA|Model name|Socket" 
} > gpurun_out/numa/info.txt 2>&1
cat gpurun_out/numa/info.txt
