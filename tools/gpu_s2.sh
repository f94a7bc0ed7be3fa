set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hot_shapes_gpu.py -q -m gpu -x -k "enc0 or enc1 or D1 or D2 or D3" > gpurun_out/s2_tests.log 2>&1 || (tail -30 gpurun_out/s2_tests.log; exit 1)
tail -2 gpurun_out/s2_tests.log
timeout -k 10 300 python tools/bench_s2.py 2>&1 | tee gpurun_out/bench_s2.log
