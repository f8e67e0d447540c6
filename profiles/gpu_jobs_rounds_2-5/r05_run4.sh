# round 5, fourth GPU session: everything that changed since the first one, through the WHOLE GPU suite (32-bit offsets in the flattened scene's engine, the listed /
# wave-per-pixel SVGF variance pass, early split clipping in front of the device BLAS build at its default 0.08), then the driver's command for the record
mkdir -p gpurun_out; rm -f gpurun_out/parity_numbers.txt gpurun_out/parity_pixel_breakdown.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v WARNING | tail -40 > gpurun_out/r05_run4_pytest.log ) 2>&1 | grep real; tail -4 gpurun_out/r05_run4_pytest.log; grep -n "^FAILED\|^E  " gpurun_out/r05_run4_pytest.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_run4.json 2> gpurun_out/r05_bench_run4.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_run4.json")); r = d["roofline"]; b = r.get("binding", {})
    print("%.3f ms/step %.1f Mrays/s | frac %s hbm_frac %s | binding %s" % (d["ms_per_step"], d["value"], r.get("frac"), r.get("hbm_frac"), b.get("utilisation_by_unit")))
    print("stages", {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, "| config3", (d.get("config3") or {}).get("ms_per_filtered_frame"), (d.get("config3") or {}).get("filter_ms_per_frame"), "| ref layout", (d.get("reference_layout") or {}).get("ms_per_step"), "| errors", r.get("pmc_errors"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r05_bench_run4.err").read()[-2000:])
PY
