mkdir -p gpurun_out
for f in "" "--set-option dgrad_s2_ring=0" "--no-fold-eval-bn" "--set-option halo16_s2=0" "--no-wgrad-stream" "--no-fuse-bwd"; do
  r=$(python3 bench.py --use-spectral --add-noise --steps 10 --warmup 3 --no-cpu-baseline --no-roofline $f 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "spectral+noise [$f]: $r" | tee -a gpurun_out/spec.log
done
r=$(python3 bench.py --stage mae --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
echo "mae: $r" | tee -a gpurun_out/spec.log
bash profiles/stats_only.sh r03_spec --use-spectral --add-noise
