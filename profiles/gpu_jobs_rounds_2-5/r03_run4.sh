# round 3, fourth GPU session: XCD-aware image kernels, variants of sort / shade / traversal at the driver's 20 steps
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_reference_kernels.py tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py tests/test_gpu_widening.py -q 2>&1 | tail -15 > gpurun_out/r03_run4_pytest.log; tail -8 gpurun_out/r03_run4_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc"
for v in default sortg8k sortw6 sortw6g8k shadew5 shadew3 hold8 hold16 nd2nw8 mixall batch3 default; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  extra="--no-config3"; [ $v = default ] && extra=""
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B $extra > gpurun_out/r03_run4_$v.json 2>gpurun_out/r03_run4_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03_run4_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
    c3 = d.get("config3")
    if c3:
        print("   config3: %.3f ms per frame, filter %.4f ms: %s" % (c3["ms_per_filtered_frame"], c3["filter_ms_per_frame"], " ".join("%s %.4f" % (k["kernel"][7:], k["ms_per_frame"]) for k in c3["kernels"])))
except Exception as e: print("$v failed", e)
PY
done
