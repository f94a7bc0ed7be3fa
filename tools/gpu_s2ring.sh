set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_hot_shapes_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x > gpurun_out/s2ring_tests.log 2>&1 || (tail -40 gpurun_out/s2ring_tests.log; exit 1)
tail -2 gpurun_out/s2ring_tests.log
FLAG_A="" FLAG_B="--set-option dgrad_s2_ring=0" bash tools/gpu_ab3.sh
