mkdir -p gpurun_out
for cfg in "" "--stage mae" "--use-spectral --add-noise"; do
 for i in 1 2; do
  python3 bench.py $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] wall %.2f ms host %.2f ms' % (j['ms_per_step'], j['host_enqueue_ms_per_step']))" | tee -a gpurun_out/host_cfg.log
 done
done
