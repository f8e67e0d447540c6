# round 5, GPU sessions 9-..: issue priority by phase of the traversal round (s_setprio; RT_PRIO, kernels_trace.hip; VARIANTS="shipped prioNNNNN ..."); the shipped build for box drift.
# Same arithmetic, same instruction stream otherwise: no parity run needed for a priority hint; the driver's command without the side sections, stages on.
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in ${VARIANTS:-shipped prio1 prio2 prio3 prio4 shipped2}; do
  lib=""; case $name in shipped|shipped2) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r05_run9_$name.json 2> gpurun_out/r05_run9_$name.err
  python - <<PY | tee -a gpurun_out/r05_run9_summary.txt
import json
try:
    d = json.load(open("gpurun_out/r05_run9_$name.json")); st = {s["stage"]: s["ms_per_step"] for s in d["roofline"].get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f  sort %.4f  diffuse %.4f  plastic %.4f" % ("$name", d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r05_run9_$name.err").read()[-800:])
PY
done
