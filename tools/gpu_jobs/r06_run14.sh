# round 6, GPU session 14: guided self-scheduling of the ray cursor (a claim takes remaining / (D x waves), 64 .. max) against the fixed 256 (now the default build): the driver's
# command and rank 0's share of 8 (short launches), the trace tests on the default build first
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py -x -q -k "seating or bit_exact or statistics or burst" 2>&1 | tail -2
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in default g2m512 g4m512 g4m1024 g8m1024 g8m2048 g16m1024 default2; do
  lib=""; case $name in default|default2) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  for w in 0 8; do
    GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world $w > gpurun_out/r06_run14_${name}_$w.json 2> gpurun_out/r06_run14_${name}_$w.err
  done
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run14_${name}_0.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    e = json.load(open("gpurun_out/r06_run14_${name}_8.json"))
    print("%-10s %.4f ms/step  traversal %.4f  launch max %.3f | rank 0 of 8: %.4f ms/step" % ("$name", d["ms_per_step"], st.get("traversal", 0), r["launch_ms"]["max"], e["ms_per_step"]))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run14_${name}_0.err").read()[-600:])
PY
done
