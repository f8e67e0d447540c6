# round 4, fourth GPU session: the node test rewritten for the chip's two vector pipes (RT_FAST_NODE), parity first, then timing
mkdir -p gpurun_out
R=$PWD
export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/f2/libgrt_device.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_static_geometry.py tests/test_gpu_full_size.py::test_benchmarked_sponza_frame_matches_the_oracle -x -q 2>&1 | tail -15 > gpurun_out/r04_run4_pytest_f2.log; tail -4 gpurun_out/r04_run4_pytest_f2.log
unset GRT_DEVICE_LIB
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base f1 f2 f2_w6 f2_w5 base f2; do
  unset GRT_DEVICE_LIB
  if [ $v != base ]; then export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run4_$v.json 2>gpurun_out/r04_run4_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run4_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
