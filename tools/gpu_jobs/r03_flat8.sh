# engine variant without TLAS / instance code for a scene that is one flattened tree: the trace / static-geometry / full-size tests, then the bench
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -6
for m in 1 1; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-static $m --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r03_flat8_m$m.json 2> gpurun_out/r03_flat8_m$m.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03_flat8_m$m.json'))
print('merge_static $m: %.3f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']), ' '.join('%s %.4f' % (s['stage'], s['ms_per_step']) for s in d['roofline'].get('stages', [])), 'config3', d['config3'].get('ms_per_filtered_frame'))
PY
done
