# round 5, GPU session 12: the flattened tree's children re-seated by sample rays (bvh8_learn_slot_order, config static_slot_learning_rays) on top of slot assignment 5 / 0
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for spec in ${SPECS:-5:0 5:400000 5:4000000 0:4000000 5:0}; do
  mode=$(echo $spec | cut -d: -f1); rays=$(echo $spec | cut -d: -f2); view=$(echo $spec | cut -d: -f3)
  BENCH_SLOT_ASSIGNMENT=$mode BENCH_SLOT_LEARNING_RAYS=$rays BENCH_SLOT_LEARNING_VIEWPOINT=${view:-0} timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r05_run12.json 2> gpurun_out/r05_run12.err
  python - <<PY | tee -a gpurun_out/r05_run12_summary.txt
import json
try:
    d = json.load(open("gpurun_out/r05_run12.json")); st = {s["stage"]: s["ms_per_step"] for s in d["roofline"].get("stages", [])}; r = d["roofline"]
    print("slots %-2s learn %-8s view ${view:-0} %.4f ms/step  traversal %.4f  | nodes/tris per ray %.2f / %.2f, per shadow ray %.2f / %.2f | flatten %.2f s" % ("$mode", "$rays", d["ms_per_step"], st.get("traversal", 0), r["nodes_per_ray"], r["triangles_per_ray"], r["nodes_per_shadow_ray"], r["triangles_per_shadow_ray"], d.get("flatten_build_s") or -1))
except Exception as e: print("$spec failed", e); print(open("gpurun_out/r05_run12.err").read()[-800:])
PY
done
