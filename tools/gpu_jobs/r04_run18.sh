# round 4, eighteenth GPU session: the hit's chain in the material kernels -- material record + texture descriptor fetched beside the triangle (emat),
# the light sample picked before the surface set-up (elight), both (eboth), the round's two appends in one (oneappend), all three (eboth_oa);
# sort workgroups of 512 at 8 waves. Base = the new default build (sort at 8 waves per SIMD, shade / post units without the SLP vectoriser).
mkdir -p gpurun_out
R=$PWD
T="tests/test_gpu_full_size.py::test_benchmarked_sponza_frame_matches_the_oracle tests/test_gpu_parity.py tests/test_gpu_materials_svgf.py tests/test_gpu_widening.py tests/test_gpu_reference_kernels.py"
GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/eboth_oa/libgrt_device.so timeout 900 python -m pytest $T -x -q 2>&1 | tail -8 > gpurun_out/r04_run18_pytest_eboth_oa.log; echo "eboth_oa: $(tail -1 gpurun_out/r04_run18_pytest_eboth_oa.log)"
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base emat elight eboth oneappend eboth_oa sort_b512 base2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run18_$v.json 2>gpurun_out/r04_run18_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run18_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
