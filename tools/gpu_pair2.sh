# padded-channel rerun + serial kernel stats of the paired-pass step
set -e
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_model_gpu.py -q -m gpu -k "odd_batch" > gpurun_out/pair_tests2.log 2>&1 || (tail -30 gpurun_out/pair_tests2.log; exit 1)
tail -2 gpurun_out/pair_tests2.log
bash profiles/stats_only.sh r03_d_serial --no-wgrad-stream
bash profiles/stats_only.sh r03_d
