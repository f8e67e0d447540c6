# round 4, sixth GPU session: the top of the flattened tree in LDS (rt_set_node_cache): parity, then timing against the launch without it
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_node_format.py tests/test_gpu_parity.py tests/test_gpu_static_geometry.py tests/test_gpu_full_size.py::test_benchmarked_sponza_frame_matches_the_oracle -x -q 2>&1 | tail -15 > gpurun_out/r04_run6_pytest.log; tail -4 gpurun_out/r04_run6_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in nocache cached c_w7s8 c_w5 c_f2 c_w7s8_f2 nocache cached; do
  unset GRT_DEVICE_LIB; nc=1
  case $v in nocache) nc=0;; cached) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --node-cache $nc > gpurun_out/r04_run6_$v.json 2>gpurun_out/r04_run6_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run6_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f | %s" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), r.get("kernel")))
except Exception as e: print("$v failed", e)
PY
done
