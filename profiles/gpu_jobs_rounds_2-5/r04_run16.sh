# round 4, sixteenth GPU session: the flattened scene's engine with the next node's loads in flight behind the triangle tests
# (bvh8_trace_engine_flat_pipelined): parity of two variants, then timing against the shipped launch
mkdir -p gpurun_out
R=$PWD
T="tests/test_gpu_static_geometry.py::test_flattened_sponza_on_the_device_finds_what_the_reference_layout_finds tests/test_gpu_full_size.py::test_benchmarked_sponza_frame_matches_the_oracle tests/test_gpu_parity.py"
for v in pipe_w5t2 pipe_w6t1; do
  GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so timeout 600 python -m pytest $T -x -q 2>&1 | tail -6 > gpurun_out/r04_run16_pytest_$v.log; echo "$v: $(tail -1 gpurun_out/r04_run16_pytest_$v.log)"
done
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base pipe_w5t2 pipe_w6t1 pipe_w6t2 pipe_w5t1 base2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run16_$v.json 2>gpurun_out/r04_run16_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run16_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
