# round 5, GPU session 21: `python bench.py` with no flags (N = 1, 64 steps, every section) on the round's last state: how long it takes, what it prints
mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_default.json")); r = d["roofline"]
    print("%d steps: %.3f ms/step %.1f Mrays/s | frac %s hbm_frac %s | config3 %s | cpu_baseline %s %s" % (d["steps"], d["ms_per_step"], d["value"], r.get("frac"), r.get("hbm_frac"), (d.get("config3") or {}).get("ms_per_filtered_frame"), d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"]))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r05_bench_default.err").read()[-2000:])
PY
