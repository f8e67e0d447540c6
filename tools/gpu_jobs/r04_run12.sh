# round 4, twelfth GPU session: which of the LDS tables pay (A/B on one box), sort launch shapes
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base noscene notables sort_b256 sort_g4096 base2 noscene2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; noscene2) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/noscene/libgrt_device.so;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run12_$v.json 2>gpurun_out/r04_run12_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run12_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f gen %.4f acc %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0), st.get("accumulate", 0)))
except Exception as e: print("$v failed", e)
PY
done
