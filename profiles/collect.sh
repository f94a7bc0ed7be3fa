#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash profiles/collect.sh <tag>
# Three separate rocprofv3 passes (kernel stats; FETCH_SIZE; WRITE_SIZE) -- counters are never combined with other trace
# domains, and the program itself follows `--` (no env/bash wrappers).  Outputs land under gpurun_out/ and are turned
# into the committed summaries by profiles/summarize.py.
set -e
tag=${1:-r02}
root=$(pwd)
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag" -o runc -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline \
    > "$root/gpurun_out/bench_prof_$tag.log" 2>&1
echo "[collect] kernel stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$root/gpurun_out/pmc_fetch_$tag" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-roofline > "$root/gpurun_out/pmc_fetch_$tag.log" 2>&1
echo "[collect] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$root/gpurun_out/pmc_write_$tag" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-roofline > "$root/gpurun_out/pmc_write_$tag.log" 2>&1
echo "[collect] WRITE_SIZE pass done"
# 4th pass: MFMA pipe occupancy per dispatch (SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs; GRBM_GUI_ACTIVE = GPU-busy cycles)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$root/gpurun_out/pmc_mfma_$tag" -o runc -- python3 "$root/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-roofline > "$root/gpurun_out/pmc_mfma_$tag.log" 2>&1
echo "[collect] MFMA busy pass done"
