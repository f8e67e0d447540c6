# round 6, GPU session 17: queue order of the primary rays (GRT_PRIMARY_ORDER = block width x band rows; 0 = scan lines, the reference's order)
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "benchmarked" > gpurun_out/r06_run17_pytest.log 2>&1; tail -3 gpurun_out/r06_run17_pytest.log
for name in 0 8x8 16x8 8x16 16x16 32x8 4x8 0 8x8; do
  GRT_PRIMARY_ORDER=$name timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run17_$name.json 2> gpurun_out/r06_run17_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run17_$name.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f sort %.4f diffuse %.4f plastic %.4f gen %.4f acc %.4f" % ("$name", d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0), st.get("accumulate", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run17_$name.err").read()[-600:])
PY
done
