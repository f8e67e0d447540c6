# round 6, GPU session 9: is the traversal launch's 6th workgroup per CU resident? The persistent grid at 4 / 5 / 6 / 7 / 12 workgroups per CU (GRT_TRACE_BLOCKS_PER_CU), the driver's command
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for n in 6 5 4 7 12 6; do
  GRT_TRACE_BLOCKS_PER_CU=$n timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run9_$n.json 2> gpurun_out/r06_run9_$n.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run9_$n.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("blocks per CU %-3s %.4f ms/step  traversal %.4f" % ("$n", d["ms_per_step"], st.get("traversal", 0)))
except Exception as e: print("$n failed", e); print(open("gpurun_out/r06_run9_$n.err").read()[-800:])
PY
done
