set -e
mkdir -p gpurun_out
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_ddp1" -o runc -- python3 "$root/bench.py" --gpus 1 --force-collectives --steps 5 --warmup 2 --no-roofline > "$root/gpurun_out/ddp1.log" 2>&1
cp "$root/gpurun_out/prof_ddp1"/*/runc_kernel_stats.csv "$root/gpurun_out/ddp1_kernel_stats.csv" 2>/dev/null || find "$root/gpurun_out/prof_ddp1" -name "*kernel_stats.csv" -exec cp {} "$root/gpurun_out/ddp1_kernel_stats.csv" \;
tail -1 "$root/gpurun_out/ddp1.log" | cut -c1-600
