# round 4, first GPU session: decoded nodes (parity + A/B at the driver's 20 steps), the RCCL branch, threaded FrameSplit
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_run1_pytest.log; tail -6 gpurun_out/r04_run1_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc --no-config3"
for v in reference decoded d_w6 reference decoded; do
  unset GRT_DEVICE_LIB; fmt=$v
  if [ $v = d_w6 ]; then export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fmt=decoded; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --node-format $fmt > gpurun_out/r04_run1_$v.json 2>gpurun_out/r04_run1_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run1_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
