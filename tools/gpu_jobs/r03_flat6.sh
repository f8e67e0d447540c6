# every static instance flattened (world-space copies of the transformed ones), rays never see the TLAS when nothing moves:
# whole GPU suite, then the bench for merge_static 1 (all) / 3 (identity instances only; two TLAS leaves left) / 0 on one box
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
for m in 1 3 0; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-static $m --no-cpu-baseline --no-povs --no-pmc --no-config3 > gpurun_out/r03_flat6_bench_m$m.json 2> gpurun_out/r03_flat6_bench_m$m.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03_flat6_bench_m$m.json'))
print('merge_static=$m: %.3f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']), ' '.join('%s %.4f' % (s['stage'], s['ms_per_step']) for s in d['roofline'].get('stages', [])))
PY
done
