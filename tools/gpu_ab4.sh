mkdir -p gpurun_out
for i in 1 2; do
  for f in "" "--no-roofline" "--no-forked-chains" "--no-forked-chains --no-roofline"; do
    a=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $f 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j.get('roofline'); print('%.2f ms' % j['ms_per_step'], ('| dom %.1f us other %.1f us' % (1e3*r['avg_launch_ms'], 1e3*r['other_instance']['avg_launch_ms'])) if r else '')")
    echo "[$f]: $a" | tee -a gpurun_out/ab4.log
  done
done
