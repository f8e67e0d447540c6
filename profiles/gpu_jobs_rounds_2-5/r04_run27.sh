# round 4, twenty-seventh GPU session: do the traversal / shade launch parameters chosen under the pipelined schedule still hold when every launch carries one bounce of a burst?
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base flat_w6 flat_w8 tri3 fast2_w6 mixed64 shade_w5 base2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run27_$v.json 2>gpurun_out/r04_run27_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run27_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
