# round 5, first GPU session: the pruned traversal unit (compile-time variants, decoded nodes, node cache and the pipelined engine are gone; ABI 7),
# the advisor's fixes, the new parity tests (the benchmark's own burst at 1080p, configs 4 / 5 at 4 / 16 spp, pixel break-downs) -- the whole GPU
# suite, smoke(), and the driver's command with the reworked roofline record
mkdir -p gpurun_out; rm -f gpurun_out/parity_numbers.txt gpurun_out/parity_pixel_breakdown.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 > gpurun_out/r05_run1_pytest.log ) 2>&1 | grep real; tail -3 gpurun_out/r05_run1_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_run1.json 2> gpurun_out/r05_bench_run1.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_run1.json")); r = d["roofline"]; b = r.get("binding", {})
    print("%.3f ms/step %.1f Mrays/s | frac %s hbm_frac %s l1_frac_alg %s | binding frac %s issue %s mix %s l1 %s l2 %s hbm %s closest %s" % (d["ms_per_step"], d["value"], r.get("frac"), r.get("hbm_frac"), r.get("l1_frac_of_algorithmic_bytes"),
      b.get("frac"), b.get("issue_frac"), (b.get("mix_aware") or {}).get("frac_classes_serial"), (b.get("l1") or {}).get("busy"), (b.get("l2") or {}).get("frac"), (b.get("hbm") or {}).get("frac"), b.get("closest_to_its_roof")))
    print("stages", {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, "| config3", (d.get("config3") or {}).get("ms_per_filtered_frame"), (d.get("config3") or {}).get("filter_ms_per_frame"), "| ref layout", (d.get("reference_layout") or {}).get("ms_per_step"), "| errors", r.get("pmc_errors"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r05_bench_run1.err").read()[-2000:])
PY
