set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hot_shapes_gpu.py -q -m gpu -x -k "stem or heads or D0 or 2gib" > gpurun_out/thin_tests.log 2>&1 || (tail -30 gpurun_out/thin_tests.log; exit 1)
tail -2 gpurun_out/thin_tests.log
echo "== new" | tee gpurun_out/bench_thin.log
timeout -k 10 300 python tools/bench_thin.py 2>&1 | tee -a gpurun_out/bench_thin.log
if [ -f de-i2i-gan_amd/lib/prev.so ]; then echo "== prev.so" | tee -a gpurun_out/bench_thin.log; DEI2I_LIB=$PWD/de-i2i-gan_amd/lib/prev.so timeout -k 10 300 python tools/bench_thin.py 2>&1 | tee -a gpurun_out/bench_thin.log; fi
