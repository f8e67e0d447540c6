# round 4, seventeenth GPU session: the sort kernel at twice the occupancy (64 registers + scratch in its rare branches; 512-thread groups at 80),
# the shade / post translation units without the SLP vectoriser (v_pk_mul / v_pk_add cost what two scalar ones cost here, plus the moves that pair the registers)
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-reference-layout"
for v in base sort_w8 sort_b512w6 noslp noslp_w8 base2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  C="--no-config3"; case $v in base|noslp) C="";; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B $C > gpurun_out/r04_run17_$v.json 2>gpurun_out/r04_run17_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run17_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    c3 = d.get("config3") or {}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f | config3 %s filter %s %s" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame"), [(k.get("kernel")[7:], k.get("ms_per_frame")) for k in c3.get("kernels", [])]))
except Exception as e: print("$v failed", e)
PY
done
