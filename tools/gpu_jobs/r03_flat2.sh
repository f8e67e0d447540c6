# flattened static geometry: whole GPU suite, then the driver's bench command (flattened, the default) and the reference layout beside it
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -40 > gpurun_out/r03_flat2_suite.log; tail -16 gpurun_out/r03_flat2_suite.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_flat2_bench.json 2> gpurun_out/r03_flat2_bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-static 0 --no-povs --no-config3 --no-cpu-baseline > gpurun_out/r03_flat2_bench_reference_layout.json 2> gpurun_out/r03_flat2_bench_reference_layout.err
python - <<'PY'
import json
for name in ('r03_flat2_bench', 'r03_flat2_bench_reference_layout'):
    d = json.load(open('gpurun_out/%s.json' % name)); r = d['roofline']
    print('%s: %.3f ms/step %.1f Mrays/s | binding %s | lanes %s | traffic %s' % (name, d['ms_per_step'], d['value'], r.get('binding', {}).get('frac'), r.get('binding', {}).get('lane_utilisation'), r.get('traffic')))
    for s in r.get('stages', []): print('   ', s['stage'], s['ms_per_step'], s.get('lane_utilisation'))
    if 'config3' in d: print('    config3', d['config3'].get('ms_per_frame'))
    if 'cpu_baseline' in d: print('    cpu', d['cpu_baseline']['value'])
PY
