# paired generator passes: parity tests
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_fused_norm_gpu.py tests/test_model_gpu.py -q -m gpu -s > gpurun_out/pair_tests.log 2>&1
grep -n "paired vs four\|whole gradient\|passed\|failed\|^FAILED\|Error" gpurun_out/pair_tests.log | head -40
