# round 5, GPU session 18: what the slot learner's sample rays should be when the camera is NOT where the tree was trained: the driver's command + the reference's nine points of view
# (GRT_SLOT_LEARNING_MIX = camera share : free-space share of the budget; the rest surface rays)
mkdir -p gpurun_out
B="--no-cpu-baseline --no-pmc --no-config3 --no-reference-layout --no-stages"
for mix in ${MIXES:-0.75:0 0.5:0.25 0.25:0.5 0:0.75}; do
  GRT_SLOT_LEARNING_MIX=$mix timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r05_run18.json 2> gpurun_out/r05_run18.err
  python - <<PY | tee -a gpurun_out/r05_run18_summary.txt
import json
try:
    d = json.load(open("gpurun_out/r05_run18.json")); r = d["roofline"]; p = d["povs"]
    print("mix $mix: %.4f ms/step | nodes/tris per ray %.2f / %.2f, per shadow ray %.2f / %.2f | povs avg %.3f ms/step %s" % (d["ms_per_step"], r["nodes_per_ray"], r["triangles_per_ray"], r["nodes_per_shadow_ray"], r["triangles_per_shadow_ray"], p["ms_per_step_avg"], [x["ms_per_step"] for x in p["per_pov"]]))
except Exception as e: print("$mix failed", e); print(open("gpurun_out/r05_run18.err").read()[-800:])
PY
done
