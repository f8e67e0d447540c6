# round 6, GPU session 12: rays claimed per cursor atomic (RT_FETCH_BLOCK_MAX) -- 128 shipped; 256 measured -4.9 % traversal in session 11
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in shipped fb192 fb256 fb384 fb512 fb1024 fb2048 fb512nd6 fb256 shipped2; do
  lib=""; case $name in shipped|shipped2) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run12_$name.json 2> gpurun_out/r06_run12_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run12_$name.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f  launches ms %s" % ("$name", d["ms_per_step"], st.get("traversal", 0), r.get("launch_ms")))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run12_$name.err").read()[-600:])
PY
done
