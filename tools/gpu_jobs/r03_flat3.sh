# flattened tree built with spatial splits (merge_static 1) vs the SAH sweep (2): tests, then the bench
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py tests/test_gpu_blas.py -m gpu -q -x 2>&1 | tail -8
for m in 1 2; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-static $m --no-cpu-baseline --no-povs --no-pmc --no-config3 > gpurun_out/r03_flat3_bench_m$m.json 2> gpurun_out/r03_flat3_bench_m$m.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03_flat3_bench_m$m.json'))
print('merge_static=$m: %.3f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']))
for s in d['roofline'].get('stages', []): print('   ', s['stage'], s['ms_per_step'])
PY
done
