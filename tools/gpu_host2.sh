set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_fused_norm_gpu.py tests/test_mae_gpu.py tests/test_ops_gpu.py -q -m gpu -x > gpurun_out/host2_tests.log 2>&1 || (tail -40 gpurun_out/host2_tests.log; exit 1)
tail -2 gpurun_out/host2_tests.log
bash tools/gpu_host.sh
