# rays start inside the flattened tree: whole GPU suite, then the bench (20 steps, no counters)
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r03_flat5_bench.json 2> gpurun_out/r03_flat5_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_flat5_bench.json'))
print('%.3f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']))
for s in d['roofline'].get('stages', []): print('   ', s['stage'], s['ms_per_step'])
print('config3', d['config3'].get('ms_per_filtered_frame'), d['config3'].get('filter_ms_per_frame'))
PY
