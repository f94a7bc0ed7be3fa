#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/stats_only.sh <tag> [bench.py flags...]
# One rocprofv3 --kernel-trace --stats pass of bench.py (5 + 2 steps) -> gpurun_out/prof_<tag>/.../runc_kernel_stats.csv
set -e
tag=$1; shift
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag" -o runc -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" \
    > "$root/gpurun_out/bench_prof_$tag.log" 2>&1
find "$root/gpurun_out/prof_$tag" -name "*kernel_stats.csv" -exec cp {} "$root/gpurun_out/${tag}_kernel_stats.csv" \;
echo "[stats_only] $tag done"
