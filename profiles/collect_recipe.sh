set -e
bash profiles/collect.sh r01_g
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_recipe" -o runc -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --use-spectral --add-noise > "$root/gpurun_out/bench_prof_recipe.log" 2>&1
echo recipe done
