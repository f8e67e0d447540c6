# the two bench commands again (the counter passes now find the flat engine's kernel)
cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err ) 2>&1 | grep real
( time timeout 900 python bench.py > gpurun_out/r03_bench_64_steps.json 2> gpurun_out/r03_bench_64_steps.err ) 2>&1 | grep real
python - <<'PY'
import json
for name in ("r03_bench", "r03_bench_64_steps"):
    d = json.load(open("gpurun_out/%s.json" % name)); r = d["roofline"]
    print("%s: %.3f ms/step %.1f Mrays/s frac %.3f stream-frac %.3f | binding %s | cpu %s" % (name, d["ms_per_step"], d["value"], r["frac"], r.get("frac_of_measured_stream", 0), json.dumps(r.get("binding"))[:330], d.get("cpu_baseline", {}).get("value")))
    print("   traffic %s achieved %s kernel %s" % (r.get("traffic"), r.get("achieved"), r.get("kernel")))
    for s in r.get("stages", []): print("   stage", json.dumps(s))
    c3 = d.get("config3", {}); print("   config3 %s ms/frame filter %s" % (c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame")))
    print("   pov keys", [k for k in d if "pov" in k.lower()], [k for k in d["config"] if "pov" in k.lower()])
PY
