mkdir -p gpurun_out
( time timeout 1200 python bench.py > gpurun_out/r02_bench_default_args.json 2> gpurun_out/r02_bench_default_args.err ) 2>&1 | grep real; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench_default_args.json')); r=d['roofline']
print(d['n_gpus'], d['steps'], d['warmup'], d['value'], d['ms_per_step'], r['frac'], d['cpu_baseline']['value'], d.get('povs',{}).get('ms_per_step_avg'))
PY
