cd /root/repo
mkdir -p gpurun_out
for w in 6 7 6 7; do
  RT_FLAT_WAVES=$w GRT_DEBUG=1 timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 > gpurun_out/r03_waves$w.json 2> gpurun_out/r03_waves$w.err
  python - <<PY
import json
d = json.load(open('gpurun_out/r03_waves$w.json'))
print('flat engine, $w waves per SIMD: %.3f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']), ' '.join('%s %.4f' % (s['stage'], s['ms_per_step']) for s in d['roofline'].get('stages', [])))
PY
  grep "trace kernel grid" gpurun_out/r03_waves$w.err | sort | uniq -c | tail -3
done
