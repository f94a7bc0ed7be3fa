mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_starganv2_gpu.py -q -m gpu > gpurun_out/sg_tests.log 2>&1; rc=$?
tail -40 gpurun_out/sg_tests.log
exit $rc
