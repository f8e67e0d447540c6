# round 6, GPU session 25: block size of the shared cursor now that the end of a queue is dealt in regions (RT_FETCH_BLOCK_MAX m, RT_ENDGAME_REGION_MAX r)
mkdir -p gpurun_out
V=$PWD/gpu-raytracer_amd/csrc/_variants
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for name in default m384 m512 m512r512 m1024r512 r128 r512 default2; do
  lib=""; case $name in default*) ;; *) lib="$V/$name/libgrt_device.so";; esac
  for W in 0 8; do
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world $W $B > gpurun_out/r06_run25_${name}_$W.json 2> gpurun_out/r06_run25_${name}_$W.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run25_${name}_$W.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s world %d  %.4f ms/step  traversal %.4f" % ("$name", $W, d["ms_per_step"], st.get("traversal", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run25_${name}_$W.err").read()[-600:])
PY
  done
done
