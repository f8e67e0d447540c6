# round 6, GPU session 24: material queues of indices (the shade kernels read the trace queue's entries; the sort kernel writes 4 bytes per ray) -- tests, then the step
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_run24_pytest.log 2>&1; tail -3 gpurun_out/r06_run24_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for name in a b; do
  for W in 0 8; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world $W $B > gpurun_out/r06_run24_${name}_$W.json 2> gpurun_out/r06_run24_${name}_$W.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run24_${name}_$W.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s world %d  %.4f ms/step  traversal %.4f sort %.4f diffuse %.4f plastic %.4f gen %.4f" % ("$name", $W, d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0)))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run24_${name}_$W.err").read()[-600:])
PY
  done
done
