# round 4, second GPU session: why are fewer vector instructions not faster? instruction rates and node-fetch cost on the chip itself,
# then the node step's variants one by one (packed / scalar multiply-adds x 80 / 96 / 128-byte nodes); the one-hop parity tests
mkdir -p gpurun_out
R=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/microbench/valu_rates.hip 2>/dev/null && timeout 120 /tmp/valu_rates > gpurun_out/r04_valu_rates.txt 2>&1; cat gpurun_out/r04_valu_rates.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/node_fetch tools/microbench/node_fetch.hip 2>/dev/null && timeout 120 /tmp/node_fetch > gpurun_out/r04_node_fetch.txt 2>&1; cat gpurun_out/r04_node_fetch.txt
rm -f gpurun_out/parity_numbers.txt
timeout 900 python -m pytest tests/test_gpu_tlas.py::test_device_tlas_switched_on_after_the_scene_was_flattened tests/test_gpu_reference_kernels.py tests/test_gpu_full_size.py::test_benchmarked_sponza_frame_matches_the_oracle -x -q 2>&1 | tail -15 > gpurun_out/r04_run2_pytest.log; tail -6 gpurun_out/r04_run2_pytest.log; cat gpurun_out/parity_numbers.txt
B="--no-cpu-baseline --no-povs --no-pmc --no-config3"
for v in reference decoded d_nopk r_pk r_meta r_pk_meta d_128 d_nopk_128 reference; do
  unset GRT_DEVICE_LIB; fmt=decoded
  case $v in reference) fmt=reference;; decoded) ;; r_*) fmt=reference; export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --node-format $fmt > gpurun_out/r04_run2_$v.json 2>gpurun_out/r04_run2_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run2_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$v failed", e)
PY
done
