set -e
mkdir -p gpurun_out
python -m pytest tests/test_fused_norm_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py tests/test_ops_gpu.py -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || (tail -40 gpurun_out/ab_tests.log; exit 1)
tail -2 gpurun_out/ab_tests.log
for i in 1 2; do
  a=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $FLAG_A 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  b=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $FLAG_B 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "A [$FLAG_A]: $a | B [$FLAG_B]: $b"
done
python bench.py --gpus 1 --force-collectives --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_force_collectives.json
python bench.py --gpus 1 --force-collectives --comm-dtype bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_force_collectives_bf16.json
python - <<'PY'
import json
for f in ('gpurun_out/r03_force_collectives.json','gpurun_out/r03_force_collectives_bf16.json'):
    j=json.load(open(f)); print(f, j['ms_per_step'], json.dumps(j['ddp']))
PY
