# round 5, GPU session 11: how the flattened tree's 8-wide collapse deals a node's children to the octant slots (config static_slot_assignment; tools/wave_sim first:
# profiles/r05_slot_assignment.txt): the reference's greedy rule (0) against assignments of least total cost (1, 2, 5) and the exchange search (7); bit-exact trace tests per mode
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for mode in ${MODES:-0 1 2 5 7 0}; do
  BENCH_SLOT_ASSIGNMENT=$mode timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r05_run11_$mode.json 2> gpurun_out/r05_run11_$mode.err
  python - <<PY | tee -a gpurun_out/r05_run11_summary.txt
import json
try:
    d = json.load(open("gpurun_out/r05_run11_$mode.json")); st = {s["stage"]: s["ms_per_step"] for s in d["roofline"].get("stages", [])}; r = d["roofline"]
    print("slots %-2s %.4f ms/step  traversal %.4f  | nodes/tris per ray %.2f / %.2f, per shadow ray %.2f / %.2f | sort %.4f diffuse %.4f plastic %.4f" % ("$mode", d["ms_per_step"], st.get("traversal", 0), r["nodes_per_ray"], r["triangles_per_ray"], r["nodes_per_shadow_ray"], r["triangles_per_shadow_ray"], st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
except Exception as e: print("$mode failed", e); print(open("gpurun_out/r05_run11_$mode.err").read()[-800:])
PY
done
