# round 6, GPU session 1: "skip behind the hit" (rt_set_skip_behind_hit) -- the bit-exact trace / statistics tests in the new default walk, then the driver's
# command A / B / A / B (reference walk, skipping walk; the side sections off), then the whole GPU suite.
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
timeout 900 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py -x -q -k "bit_exact or flattened or statistics" 2>&1 | tail -3
for name in walk0_a skip1_a walk0_b skip1_b; do
  case $name in walk0*) v=0;; *) v=1;; esac
  BENCH_SKIP_BEHIND_HIT=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run1_$name.json 2> gpurun_out/r06_run1_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run1_$name.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f  sort %.4f  diffuse %.4f  plastic %.4f | nodes/tris %s %s shadow %s %s" % ("$name", d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), r.get("nodes_per_ray"), r.get("triangles_per_ray"), r.get("nodes_per_shadow_ray"), r.get("triangles_per_shadow_ray")))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run1_$name.err").read()[-1500:])
PY
done
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v WARNING | tail -25 > gpurun_out/r06_run1_pytest.log ) 2>&1 | grep real; tail -3 gpurun_out/r06_run1_pytest.log; grep -n "^FAILED" gpurun_out/r06_run1_pytest.log | head
