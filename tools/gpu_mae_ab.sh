mkdir -p gpurun_out
for i in 1 2 3; do
  a=$(python3 _ab_prev/bench.py --stage mae --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  b=$(python3 bench.py --stage mae --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "mae prev(r02): $a | new: $b" | tee -a gpurun_out/mae_ab.log
done
for i in 1 2; do
  a=$(python3 _ab_prev/bench.py --use-spectral --add-noise --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  b=$(python3 bench.py --use-spectral --add-noise --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "spectral+noise prev(r02): $a | new: $b" | tee -a gpurun_out/mae_ab.log
done
