# A/B of a bench flag: bash ab_flag.sh "--flag" reps
flag=$1; reps=${2:-3}
for i in $(seq 1 $reps); do
  a=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  b=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $flag 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "default: $a | $flag: $b"
done
