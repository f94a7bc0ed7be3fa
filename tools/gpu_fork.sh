set -e
mkdir -p gpurun_out
python -m pytest tests/test_fused_norm_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_hot_shapes_gpu.py -q -m gpu -x > gpurun_out/fork_tests.log 2>&1 || (tail -40 gpurun_out/fork_tests.log; exit 1)
tail -2 gpurun_out/fork_tests.log
FLAG_A="--no-forked-chains" FLAG_B="" bash tools/gpu_ab3.sh
