# round 6, GPU session 10: the slot table / statistics rows travel with the iteration (one copy + the advance launch per iteration instead of a copy and a fill per submission):
# the scheduler tests, then the driver's command and rank 0's share of an 8-GPU split (--emulate-world 8)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -k "merged or pipelined or burst or advances or batch or samples or statistics or deterministic or benchmarked" 2>&1 | grep -v WARNING | tail -5
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for w in 0 8; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world $w > gpurun_out/r06_run10_w$w.json 2> gpurun_out/r06_run10_w$w.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run10_w$w.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("emulate-world %-2s %.4f ms/step  traversal %.4f  %s" % ("$w", d["ms_per_step"], st.get("traversal", 0), d["value"]))
except Exception as e: print("$w failed", e); print(open("gpurun_out/r06_run10_w$w.err").read()[-800:])
PY
done
