# round 4, twenty-second GPU session: a declared burst of whole frames (rt_set_stream_batch): parity, then the driver's command with and without
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "burst or pipelined or merged or advances" 2>&1 | tail -6 > gpurun_out/r04_run22_pytest.log; tail -3 gpurun_out/r04_run22_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for spec in "1 20" "0 20" "1 20" "1 32" "0 32" "0 160"; do
  set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $2 --warmup 5 --burst $1 $B > gpurun_out/r04_run22_b$1_s$2.json 2> gpurun_out/r04_run22_b$1_s$2.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run22_b$1_s$2.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("burst $1 steps %3d: %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f | launches %s frac %s" % ($2, d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), r.get("launches"), r.get("frac")))
except Exception as e: print("burst $1 steps $2 failed", e); print(open("gpurun_out/r04_run22_b$1_s$2.err").read()[-800:])
PY
done
