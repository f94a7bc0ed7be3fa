# full GPU suite, the round's bench lines, kernel stats (default + serial) and the PMC passes (tag $1)
set -e
tag=${1:-r03}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/${tag}_tests.log 2>&1 || (tail -40 gpurun_out/${tag}_tests.log; exit 1)
tail -2 gpurun_out/${tag}_tests.log
python3 bench.py 2>gpurun_out/bench_${tag}.err | tail -1 > gpurun_out/${tag}_bench.json
python3 -c "
import json; j=json.load(open('gpurun_out/${tag}_bench.json')); r=j['roofline']
print('step %.2f ms %.1f pairs/s | dominant %.1f us %.3f | other %.1f us %.3f | util %.3f' % (j['ms_per_step'], j['value'], 1e3*r['avg_launch_ms'], r['frac'], 1e3*r['other_instance']['avg_launch_ms'], r['other_instance']['frac'], j['mfma']['step_mfma_util']))"
bash profiles/collect.sh $tag
bash profiles/stats_only.sh ${tag}_serial --no-wgrad-stream --no-forked-chains
