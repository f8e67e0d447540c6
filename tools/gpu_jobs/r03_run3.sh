# round 3, third GPU session: the whole GPU suite (new tests, SVGF rewrite, texture unit, FrameSplit), quick bench, shade stage times
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03_run3_pytest.log; tail -25 gpurun_out/r03_run3_pytest.log
for steps in 20 64; do
timeout 600 python bench.py --steps $steps --warmup 5 --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r03_run3_bench_$steps.json 2> gpurun_out/r03_run3_bench_$steps.err
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r03_run3_bench_$steps.json")); r = d["roofline"]
    print("bench $steps steps: %.3f ms/step %.1f Mrays/s frac %.3f" % (d["ms_per_step"], d["value"], r["frac"]))
    for s in r.get("stages", []): print("  stage %-18s %.4f ms/step  frac %.3f" % (s["stage"], s["ms_per_step"], s["frac"]))
    c3 = d.get("config3", {})
    print("config3: %s ms per frame, filter %s ms" % (c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame")))
    for k in c3.get("kernels", []): print("   %-24s %.4f ms  frac_unique %.3f" % (k["kernel"], k["ms_per_frame"], k["frac_unique"]))
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/r03_run3_bench_$steps.err").read()[-1500:])
PY
done
