# the driver's bench command at HEAD (flattened static geometry with spatial splits), with counters, stages, config 3, POVs
cd /root/repo
mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_flat4_bench.json 2> gpurun_out/r03_flat4_bench.err
timeout 900 python bench.py > gpurun_out/r03_flat4_bench_default.json 2> gpurun_out/r03_flat4_bench_default.err
python - <<'PY'
import json
for name in ('r03_flat4_bench', 'r03_flat4_bench_default'):
    d = json.load(open('gpurun_out/%s.json' % name)); r = d['roofline']
    print('%s: %.3f ms/step %.1f Mrays/s | frac %s | binding %s | traffic %s' % (name, d['ms_per_step'], d['value'], r.get('frac'), json.dumps(r.get('binding'))[:400], r.get('traffic')))
    for s in r.get('stages', []): print('   ', json.dumps(s))
    if 'config3' in d: print('    config3', d['config3'].get('ms_per_filtered_frame'), d['config3'].get('filter_ms_per_frame'))
    if 'cpu_baseline' in d: print('    cpu', d['cpu_baseline']['value'])
    print('    povs', d['config'].get('povs_ms_per_step_avg'), [k for k in d['config'].keys() if 'pov' in k])
PY
