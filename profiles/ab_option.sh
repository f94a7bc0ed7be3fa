#!/bin/bash
# A/B of one library option on the headline bench, interleaved runs on one box:  bash profiles/ab_option.sh NAME=INT [reps] [extra bench args]
opt=$1; reps=${2:-3}; shift; shift
for i in $(seq 1 $reps); do
  a=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  halo %.0f TF/s' % (j['ms_per_step'], j['roofline']['achieved']))")
  b=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --set-option $opt "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms  halo %.0f TF/s' % (j['ms_per_step'], j['roofline']['achieved']))")
  echo "default: $a   |   $opt: $b"
done
