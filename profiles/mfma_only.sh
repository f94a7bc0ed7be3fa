#!/bin/bash
# the MFMA-busy PMC pass of collect.sh alone (run on the GPU box):  bash profiles/mfma_only.sh <tag>
tag=${1:-r01}; root=$(pwd); mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$root/gpurun_out/pmc_mfma_$tag" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-roofline > "$root/gpurun_out/pmc_mfma_$tag.log" 2>&1
echo "[collect] MFMA busy pass done"
