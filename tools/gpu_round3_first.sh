set -e
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --deselect tests/test_hot_shapes_gpu.py --deselect tests/test_fused_norm_gpu.py > gpurun_out/r03_t2.log 2>&1 || (tail -40 gpurun_out/r03_t2.log; exit 1)
tail -3 gpurun_out/r03_t2.log
python bench.py 2>gpurun_out/r03_a_bench.err | tail -1 > gpurun_out/r03_a_bench.json
cat gpurun_out/r03_a_bench.json
bash profiles/stats_only.sh r03_a_serial --no-wgrad-stream
