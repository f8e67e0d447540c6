# round 5, GPU session 19: the whole GPU suite, smoke() and the driver's command on the last build (the learner: a quarter camera paths, half free-space rays, a quarter surface rays)
mkdir -p gpurun_out; rm -f gpurun_out/parity_numbers.txt gpurun_out/parity_pixel_breakdown.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v WARNING | tail -25 > gpurun_out/r05_run19_pytest.log ) 2>&1 | grep real; tail -3 gpurun_out/r05_run19_pytest.log; grep -n "^FAILED" gpurun_out/r05_run19_pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_run19.json 2> gpurun_out/r05_bench_run19.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_run19.json")); r = d["roofline"]; b = r.get("binding", {})
    print("%.3f ms/step %.1f Mrays/s | frac %s hbm_frac %s | binding %s" % (d["ms_per_step"], d["value"], r.get("frac"), r.get("hbm_frac"), b.get("utilisation_by_unit")))
    print("stages", {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, "| config3", (d.get("config3") or {}).get("ms_per_filtered_frame"), (d.get("config3") or {}).get("filter_ms_per_frame"), "| ref layout", (d.get("reference_layout") or {}).get("ms_per_step"), "| errors", r.get("pmc_errors"), "| nodes/tris", r.get("nodes_per_ray"), r.get("triangles_per_ray"), r.get("nodes_per_shadow_ray"), r.get("triangles_per_shadow_ray"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r05_bench_run19.err").read()[-2000:])
PY
