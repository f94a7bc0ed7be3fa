set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_fused_norm_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py -q -m gpu -x > gpurun_out/bng_tests.log 2>&1 || (tail -40 gpurun_out/bng_tests.log; exit 1)
tail -2 gpurun_out/bng_tests.log
for i in 1 2 3; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])"
done
