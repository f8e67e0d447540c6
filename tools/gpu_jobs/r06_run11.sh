# round 6, GPU session 11: the skipping launch's parameters once more (rays are shorter now): refill thresholds, ray blocks per cursor atomic, LDS stack depth,
# the mixed / separate threshold, triangles per batch. The driver's command without the side sections; shipped first and last.
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in shipped nw8 nw24 nd6 fb256 lds8 lds12 mixed20 mixed0 tri3 tri1 shipped2; do
  lib=""; case $name in shipped|shipped2) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run11_$name.json 2> gpurun_out/r06_run11_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run11_$name.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f" % ("$name", d["ms_per_step"], st.get("traversal", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run11_$name.err").read()[-600:])
PY
done
