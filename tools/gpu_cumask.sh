# experiment: two bench.py processes side by side, each on half of the CUs (HSA_CU_MASK), against one process on all of them
mkdir -p gpurun_out
one() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step %.1f pairs/s' % (j['ms_per_step'], j['value']))"; }
echo "full GPU: $(one)" | tee gpurun_out/cumask.log
echo "half (0-127) alone: $(HSA_CU_MASK=0:0-127 one)" | tee -a gpurun_out/cumask.log
(HSA_CU_MASK=0:0-127 one > gpurun_out/cm_a.txt) &
(HSA_CU_MASK=0:128-255 one > gpurun_out/cm_b.txt) &
wait
echo "two halves side by side: A $(cat gpurun_out/cm_a.txt) | B $(cat gpurun_out/cm_b.txt)" | tee -a gpurun_out/cumask.log
(one > gpurun_out/cm_a.txt) &
(one > gpurun_out/cm_b.txt) &
wait
echo "two unmasked side by side: A $(cat gpurun_out/cm_a.txt) | B $(cat gpurun_out/cm_b.txt)" | tee -a gpurun_out/cumask.log
