set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_fused_norm_gpu.py -q -m gpu -k "label_path or paired" -x > gpurun_out/lp2_tests.log 2>&1 || (tail -40 gpurun_out/lp2_tests.log; exit 1)
tail -2 gpurun_out/lp2_tests.log
bash profiles/stats_only.sh r03_i_serial --no-wgrad-stream
grep "label_gb" gpurun_out/r03_i_serial_kernel_stats.csv | cut -c1-160
