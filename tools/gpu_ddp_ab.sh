# what the gradient reducer costs on one GPU (1-rank RCCL group), by variant
mkdir -p gpurun_out
for f in "" "--ddp-no-measure" "--ddp-no-overlap" "--ddp-no-overlap --ddp-no-measure"; do
  python3 bench.py --gpus 1 --force-collectives --steps 20 --warmup 3 --no-cpu-baseline --no-roofline $f 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); d=j.get('ddp') or {}
print('[$f] with %.2f ms, without %.2f ms, cost %.2f ms' % (d.get('step_ms_with_reducer', j['ms_per_step']), d.get('step_ms_without_reducer', 0), d.get('reducer_cost_ms_per_step', 0)))" | tee -a gpurun_out/ddp_ab.log
done
