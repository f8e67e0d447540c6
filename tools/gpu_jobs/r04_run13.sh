# round 4, thirteenth GPU session: are the sort / material / traversal launches bound by the rate of their queue atomics? (half as many per launch)
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base shade_b512 sort_b1024 fetch256 fetch512 base2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run13_$v.json 2>gpurun_out/r04_run13_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run13_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f gen %.4f acc %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0), st.get("accumulate", 0)))
except Exception as e: print("$v failed", e)
PY
done
