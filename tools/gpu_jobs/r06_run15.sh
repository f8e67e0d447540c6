# round 6, GPU session 15: block divisor / refill thresholds around the 256-ray block
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in default div1 div4 fb320 fb256nd6 fb256nd8 fb256nw8 default2; do
  lib=""; case $name in default|default2) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  for w in 0 8; do
    GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world $w > gpurun_out/r06_run15_${name}_$w.json 2> gpurun_out/r06_run15_${name}_$w.err
  done
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run15_${name}_0.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    e = json.load(open("gpurun_out/r06_run15_${name}_8.json"))
    print("%-10s %.4f ms/step  traversal %.4f | rank 0 of 8: %.4f ms/step" % ("$name", d["ms_per_step"], st.get("traversal", 0), e["ms_per_step"]))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run15_${name}_0.err").read()[-600:])
PY
done
