# the round's last check: smoke, the whole GPU suite, the MAE bench line of the final tree
set -e
mkdir -p gpurun_out
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python -m pytest tests -q -m gpu > gpurun_out/final_tests.log 2>&1 || (tail -40 gpurun_out/final_tests.log; exit 1)
tail -1 gpurun_out/final_tests.log
python3 bench.py --stage mae --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_mae.json
python3 -c "import json; j=json.load(open('gpurun_out/r03_bench_mae.json')); print('mae %.2f ms %.1f images/s' % (j['ms_per_step'], j['value']))"
python3 bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_last.json
python3 -c "import json; j=json.load(open('gpurun_out/r03_bench_last.json')); print('default %.2f ms %.1f pairs/s' % (j['ms_per_step'], j['value']))"
