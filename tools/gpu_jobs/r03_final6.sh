# last call of the round: parity of the final kernels (trace / frame / flattening tests), then the driver's command for the record
cd /root/repo
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
( time timeout 140 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench.json")); r = d["roofline"]
print("%.3f ms/step %.1f Mrays/s frac %.3f | binding %s lanes %s | traversal %s | config3 %s | povs %s | cpu %s" % (d["ms_per_step"], d["value"], r["frac"], r.get("binding", {}).get("frac"), r.get("binding", {}).get("lane_utilisation"),
      [s["ms_per_step"] for s in r.get("stages", []) if s["stage"] == "traversal"], d.get("config3", {}).get("ms_per_filtered_frame"), d.get("povs", {}).get("ms_per_step_avg"), d.get("cpu_baseline", {}).get("value")))
PY
