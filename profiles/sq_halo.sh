#!/bin/bash
# SQ stall counters of the conv kernels on the res-block shape (run on the GPU box):  bash profiles/sq_halo.sh
root=$(pwd); mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$root/gpurun_out/counters_list.txt" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
  --output-format csv -d "$root/gpurun_out/sq_halo" -o runc -- python3 "$root/tools/bench_conv.py" bf16 res3x3 > "$root/gpurun_out/sq_halo.log" 2>&1
echo "[sq] done rc=$?"
