# round 6, GPU session 20: the end of a queue dealt in per-wave regions (RT_ENDGAME) against the shared cursor to the last ray (variant eg0): tests, the drain wave by wave, the step
mkdir -p gpurun_out
V=$PWD/gpu-raytracer_amd/csrc/_variants
echo skip tests
GRT_DEVICE_LIB=$V/waveclock/libgrt_device.so timeout 300 python tools/wave_clock_probe.py --world 1 2>&1 | grep -v WARNING | tail -11 > gpurun_out/r06_wave_clock_endgame_world1.txt; cat gpurun_out/r06_wave_clock_endgame_world1.txt
GRT_DEVICE_LIB=$V/waveclock/libgrt_device.so timeout 300 python tools/wave_clock_probe.py --world 8 2>&1 | grep -v WARNING | tail -9 > gpurun_out/r06_wave_clock_endgame_world8.txt; cat gpurun_out/r06_wave_clock_endgame_world8.txt
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for name in default eg0 default2 eg0b; do
  lib=""; case $name in default*) ;; eg0*) lib="$V/eg0/libgrt_device.so";; esac
  for W in 0 8; do
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world $W $B > gpurun_out/r06_run20_${name}_$W.json 2> gpurun_out/r06_run20_${name}_$W.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run20_${name}_$W.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s world %d  %.4f ms/step  traversal %.4f sort %.4f diffuse %.4f plastic %.4f" % ("$name", $W, d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run20_${name}_$W.err").read()[-600:])
PY
  done
done
