set -e
timeout -k 10 600 python3 -m pytest tests/test_mae_gpu.py tests/test_sean_gpu.py tests/test_adain_gpu.py -q -m gpu -x 2>&1 | tail -3
bash tools/gpu_mae3.sh
