#!/bin/bash
# interleaved same-box A/B of the previous build of the library (de-i2i-gan_amd/lib/prev.so) against the current one
reps=${1:-3}; shift
for i in $(seq 1 $reps); do
  a=$(DEI2I_LIB=$PWD/de-i2i-gan_amd/lib/prev.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  b=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "prev.so: $a | new: $b"
done
