# round 4, twenty-ninth GPU session: bench.py with no arguments at all (64 steps, every section), timed
mkdir -p gpurun_out
( time timeout 1200 python bench.py > gpurun_out/r04_bench_64_steps.json 2> gpurun_out/r04_bench_64_steps.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_bench_64_steps.json")); r = d["roofline"]
    print("%d steps: %.3f ms/step %.1f Mrays/s frac %.3f | binding %s | traversal %s | config3 %s | povs %s | burst: %s" % (d["steps"], d["ms_per_step"], d["value"], r["frac"], r.get("binding", {}).get("frac"),
      [s["ms_per_step"] for s in r.get("stages", []) if s["stage"] == "traversal"], d.get("config3", {}).get("ms_per_filtered_frame"), d.get("povs", {}).get("ms_per_step_avg"), d["config"].get("burst")))
except Exception as e: print("bench failed", e); print(open("gpurun_out/r04_bench_64_steps.err").read()[-1500:])
PY
