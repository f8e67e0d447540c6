# round 6, GPU session 23: a submission's samples interleaved patch by patch in the queue of primary rays (GRT_PRIMARY_ORDER=WxHi) against sample after sample
mkdir -p gpurun_out
GRT_PRIMARY_ORDER=8x8i timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for name in 8x8 8x8i 16x8i 8x16i 32x8i 4x8i 8x8 8x8i; do
  for W in 0 8; do
  GRT_PRIMARY_ORDER=$name timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world $W $B > gpurun_out/r06_run23_${name}_$W.json 2> gpurun_out/r06_run23_${name}_$W.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run23_${name}_$W.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s world %d  %.4f ms/step  traversal %.4f sort %.4f diffuse %.4f plastic %.4f gen %.4f" % ("$name", $W, d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run23_${name}_$W.err").read()[-600:])
PY
  done
done
