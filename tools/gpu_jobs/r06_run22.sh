# round 6, GPU session 22: the WHOLE queue dealt in per-wave regions (RT_ENDGAME_WHOLE; the owner takes 64 / 128 / 256 rays at a time, helpers 64) against the shipped form (shared cursor + regions at the end)
mkdir -p gpurun_out
V=$PWD/gpu-raytracer_amd/csrc/_variants
GRT_DEVICE_LIB=$V/w128/libgrt_device.so timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k benchmarked 2>&1 | tail -1
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for name in default w64 w128 w256 default2 w128b; do
  lib=""; case $name in default*) ;; w128b) lib="$V/w128/libgrt_device.so";; *) lib="$V/$name/libgrt_device.so";; esac
  for W in 0 8; do
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world $W $B > gpurun_out/r06_run22_${name}_$W.json 2> gpurun_out/r06_run22_${name}_$W.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run22_${name}_$W.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s world %d  %.4f ms/step  traversal %.4f sort %.4f diffuse %.4f plastic %.4f" % ("$name", $W, d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run22_${name}_$W.err").read()[-600:])
PY
  done
done
