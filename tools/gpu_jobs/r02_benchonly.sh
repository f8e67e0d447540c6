mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "bench_spawns" 2>&1 | tail -1
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r.get('cache_hit_fraction_lower_bound'), r.get('bound_in_practice'))
PY
