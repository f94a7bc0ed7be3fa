for i in 1 2 3 4; do
  python3 bench.py --stage mae --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mae wall %.2f ms host %.2f ms' % (j['ms_per_step'], j['host_enqueue_ms_per_step']))"
done
python3 bench.py --stage mae --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mae (default flags, roofline on) wall %.2f ms host %.2f ms' % (j['ms_per_step'], j['host_enqueue_ms_per_step']))"
