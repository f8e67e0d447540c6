# round 4, third GPU session: instruction costs (one asm block per kernel), the one-hop parity tests with the reference's light order,
# the driver's bench command with the reference's own textures + the reference-layout pass in the same line
mkdir -p gpurun_out
timeout 300 python tools/microbench/valu_rates2.py gpurun_out/r04_valu_rates2.txt | tail -70
rm -f gpurun_out/parity_numbers.txt
timeout 900 python -m pytest tests/test_gpu_reference_kernels.py tests/test_gpu_static_geometry.py tests/test_gpu_node_format.py -x -q 2>&1 | tail -15 > gpurun_out/r04_run3_pytest.log; tail -6 gpurun_out/r04_run3_pytest.log; cat gpurun_out/parity_numbers.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_run3_bench.json 2> gpurun_out/r04_run3_bench.err ) 2>&1 | tail -4
python - <<PY
import json
d=json.load(open("gpurun_out/r04_run3_bench.json")); r=d["roofline"]
st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
print("bench %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % (d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
print("data:", d["data"]); print("reference_layout:", d.get("reference_layout")); print("flatten_build_s:", d.get("flatten_build_s")); print("binding:", r.get("binding")); print("config3:", {k: v for k, v in d.get("config3", {}).items() if k != "kernels"})
PY
