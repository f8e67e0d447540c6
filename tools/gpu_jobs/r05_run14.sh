# round 5, GPU session 14: what the sample paths of the slot learner should contain (GRT_SLOT_LEARNING_SHADOW 0 / 1 / 2, GRT_SLOT_LEARNING_SURFACE share), 1 M rays, start 5
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for spec in ${SPECS:-1:0.25 2:0.25 0:0.25 2:0.5 0:0.5}; do
  shadow=$(echo $spec | cut -d: -f1); share=$(echo $spec | cut -d: -f2)
  GRT_SLOT_LEARNING_SHADOW=$shadow GRT_SLOT_LEARNING_SURFACE=$share timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r05_run14.json 2> gpurun_out/r05_run14.err
  python - <<PY | tee -a gpurun_out/r05_run14_summary.txt
import json
try:
    d = json.load(open("gpurun_out/r05_run14.json")); st = {s["stage"]: s["ms_per_step"] for s in d["roofline"].get("stages", [])}; r = d["roofline"]
    print("shadow mode $shadow surface share $share: %.4f ms/step  traversal %.4f  | nodes/tris per ray %.2f / %.2f, per shadow ray %.2f / %.2f" % (d["ms_per_step"], st.get("traversal", 0), r["nodes_per_ray"], r["triangles_per_ray"], r["nodes_per_shadow_ray"], r["triangles_per_shadow_ray"]))
except Exception as e: print("$spec failed", e); print(open("gpurun_out/r05_run14.err").read()[-800:])
PY
done
