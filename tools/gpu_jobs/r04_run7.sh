# round 4, seventh GPU session: background re-flatten (frame times), two contexts side by side (is there room beside the traversal launch?),
# 8 waves per SIMD for the flat engine
mkdir -p gpurun_out
R=$PWD
rm -f gpurun_out/parity_numbers.txt
timeout 900 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_tlas.py tests/test_gpu_node_format.py -x -q 2>&1 | tail -15 > gpurun_out/r04_run7_pytest.log; tail -5 gpurun_out/r04_run7_pytest.log; cat gpurun_out/parity_numbers.txt
for cfg in "1 -" "2 -" "2 5" "2 4" "2 3" "1 4"; do
  set -- $cfg; unset GRT_TRACE_BLOCKS_PER_CU; [ "$2" != "-" ] && export GRT_TRACE_BLOCKS_PER_CU=$2
  CONTEXTS=$1 timeout 300 python tools/two_context_overlap.py 32 2>/dev/null | tail -1
done
unset GRT_TRACE_BLOCKS_PER_CU
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base w8 w8b1; do
  unset GRT_DEVICE_LIB
  if [ $v != base ]; then export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run7_$v.json 2>gpurun_out/r04_run7_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run7_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
