# round 5, second GPU session: cache-policy and addressing variants of the flattened scene's traversal launch (same arithmetic, same hits):
#   nt_tri / nt_rays / nt_both  non-temporal hint on the triangle loads / the ray loads + hit stores / both (the CU's L1 is the unit closest to its roof)
#   off32 / off32_nt            32-bit byte offsets from a uniform base for node and triangle fetches (no v_mad_u64_u32 per fetch), + nt_tri
# each: the bit-exact trace tests, then the driver's command without the side sections; the shipped build first and last (box drift).
# Then the SVGF variance pass that leaves at once when no pixel is young: filter tests + config 3.
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in shipped nt_tri nt_rays nt_both off32 off32_nt shipped2; do
  lib=""; case $name in shipped|shipped2) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  if [ -n "$lib" ]; then
    GRT_DEVICE_LIB=$lib timeout 600 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py -x -q -k "bit_exact or flattened or statistics" 2>&1 | tail -1
  fi
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r05_run2_$name.json 2> gpurun_out/r05_run2_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r05_run2_$name.json")); st = {s["stage"]: s["ms_per_step"] for s in d["roofline"].get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f  sort %.4f  diffuse %.4f  plastic %.4f" % ("$name", d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r05_run2_$name.err").read()[-800:])
PY
done
timeout 900 python -m pytest tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py::test_sponza_svgf_taa_with_a_moving_camera_at_full_size tests/test_gpu_parity.py -x -q -k "svgf" 2>&1 | tail -2
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-reference-layout --no-stages > gpurun_out/r05_run2_config3.json 2> gpurun_out/r05_run2_config3.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_run2_config3.json")); c3 = d.get("config3") or {}
    print("config3 %s ms per filtered frame, filter %s | %s" % (c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame"), [(k.get("kernel")[7:], k.get("ms_per_frame")) for k in c3.get("kernels", [])]))
except Exception as e: print("config3 failed", e); print(open("gpurun_out/r05_run2_config3.err").read()[-800:])
PY
