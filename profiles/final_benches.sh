#!/bin/bash
# Run ON THE GPU BOX from the repo root: the bench lines kept under profiles/ (tag = $1), one JSON line per file.
set -e
tag=${1:-r01}
mkdir -p gpurun_out
python3 bench.py 2>gpurun_out/bench_${tag}.err | tail -1 > gpurun_out/${tag}_bench.json
python3 bench.py --dtype fp8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_fp8.json
python3 bench.py --image-size 512 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_512.json
python3 bench.py --stage mae --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_mae.json
python3 bench.py --use-spectral --add-noise --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_spectral_noise.json
for f in gpurun_out/${tag}_bench*.json; do python3 -c "
import json,sys; j=json.load(open('$f')); print('$f', '%.2f ms/step' % j['ms_per_step'], '%.1f' % j['value'], j['unit'], 'halo %.0f TF/s' % j.get('roofline',{}).get('achieved',0))"; done
