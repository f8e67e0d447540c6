# round 6, GPU session 4: the driver's command with every section on the shipped build (skipping walk with the pop as a loop; the roofline record names its
# binding unit, the L1 priced against its measured roof), then the whole GPU suite and smoke().
mkdir -p gpurun_out
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_run4.json 2> gpurun_out/r06_bench_run4.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_bench_run4.json")); r = d["roofline"]; b = r.get("binding", {})
    print("%.3f ms/step %.1f Mrays/s | bound %s frac %s (%s %s of %s) | alg/hbm %s hbm_frac %s | binding %s" % (d["ms_per_step"], d["value"], r.get("bound"), r.get("frac"), r.get("achieved"), r.get("unit"), r.get("peak"), r.get("algorithmic_bytes_over_hbm_peak"), r.get("hbm_frac"), b.get("utilisation_by_unit")))
    print("l1", {k: v for k, v in (b.get("l1") or {}).items() if k not in ("peak_derivation", "clocked_definition")})
    print("stages", {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, "| config3", (d.get("config3") or {}).get("ms_per_filtered_frame"), (d.get("config3") or {}).get("filter_ms_per_frame"), "| ref layout", (d.get("reference_layout") or {}).get("ms_per_step"), "| no-viewpoint seating", d.get("ms_per_step_seating_without_viewpoint"), "| povs", (d.get("povs") or {}).get("ms_per_step_avg"), "| errors", r.get("pmc_errors"), "| nodes/tris", r.get("nodes_per_ray"), r.get("triangles_per_ray"), r.get("nodes_per_shadow_ray"), r.get("triangles_per_shadow_ray"))
    print("counters", r.get("counters"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r06_bench_run4.err").read()[-3000:])
PY
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v WARNING | tail -25 > gpurun_out/r06_run4_pytest.log ) 2>&1 | grep real; tail -3 gpurun_out/r06_run4_pytest.log; grep -n "^FAILED" gpurun_out/r06_run4_pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
