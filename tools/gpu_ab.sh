# same-box A/B of de-i2i-gan_amd/lib/prev.so against the current library after the tests named in $TESTS pass
set -e
mkdir -p gpurun_out
python -m pytest ${TESTS:-tests/test_fused_norm_gpu.py tests/test_hot_shapes_gpu.py tests/test_ops_gpu.py} -q -m gpu -x > gpurun_out/ab_tests.log 2>&1 || (tail -40 gpurun_out/ab_tests.log; exit 1)
tail -2 gpurun_out/ab_tests.log
bash profiles/ab_lib.sh ${REPS:-2} $BENCH_FLAGS | tee gpurun_out/ab_lib.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_bench.json
python - <<'PY'
import json
j=json.load(open('gpurun_out/ab_bench.json')); r=j['roofline']
print('step %.2f ms | dominant %.1f us frac %.3f (%s) | other %.1f us frac %.3f | family %.3f' % (j['ms_per_step'], 1e3*r['avg_launch_ms'], r['frac'], r['kernel'][:46], 1e3*r['other_instance']['avg_launch_ms'], r['other_instance']['frac'], r['halo16_family_flop_weighted']['frac']))
PY
