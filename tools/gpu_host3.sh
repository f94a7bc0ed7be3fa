set -e
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x > gpurun_out/full_tests_h3.log 2>&1 || (tail -40 gpurun_out/full_tests_h3.log; exit 1)
tail -2 gpurun_out/full_tests_h3.log
bash tools/gpu_host.sh
