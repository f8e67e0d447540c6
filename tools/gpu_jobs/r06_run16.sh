# round 6, GPU session 16: the 256-ray block (default build) against 128 once more, alternating, on this box (session 15's box ran everything 7 % slower than session 14's)
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for name in default fb128 default2 fb128b default3; do
  lib=""; case $name in default*) ;; fb128*) lib="$PWD/$V/fb128/libgrt_device.so";; esac
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run16_$name.json 2> gpurun_out/r06_run16_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run16_$name.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f sort %.4f diffuse %.4f plastic %.4f | stream read %s GB/s" % ("$name", d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), r.get("measured_stream_read_gbps")))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run16_$name.err").read()[-600:])
PY
done
