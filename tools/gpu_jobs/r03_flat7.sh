# the collapse's triangle cost for the flattened tree: 1.0 (reference's converter) / 1.4 / 1.7 / 5.0, bench only
cd /root/repo
mkdir -p gpurun_out
for c in 1.0 1.4 1.7 5.0 1.0; do
  GRT_STATIC_PRIMITIVE_COST=$c timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 > gpurun_out/r03_flat7_c$c.json 2> gpurun_out/r03_flat7_c$c.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03_flat7_c$c.json'))
print('primitive cost $c: %.3f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']), ' '.join('%s %.4f' % (s['stage'], s['ms_per_step']) for s in d['roofline'].get('stages', [])))
PY
done
