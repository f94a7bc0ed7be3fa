# same-box A/B of two bench.py flag sets: FLAG_A vs FLAG_B (e.g. --set-option halo16_s2=0)
mkdir -p gpurun_out
for i in 1 2 3; do
  a=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $FLAG_A 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  b=$(python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $FLAG_B 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms' % j['ms_per_step'])")
  echo "A [$FLAG_A]: $a | B [$FLAG_B]: $b" | tee -a gpurun_out/ab3.log
done
