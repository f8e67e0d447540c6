# round 4, twenty-eighth GPU session: the two-pipe node test (RT_FAST_NODE 2) at 6 waves as the flattened scene's default: the whole GPU suite, the driver's command
# for the record, one rank's share of a 2 / 4 / 8-way split with every rank declaring its burst
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_run28_pytest.log; echo "suite: $(tail -1 gpurun_out/r04_run28_pytest.log)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_bench.json")); r = d["roofline"]
    print("%.3f ms/step %.1f Mrays/s frac %.3f | binding %s | stages %s | config3 %s filter %s | povs %s | cpu %s | reference layout %s" % (d["ms_per_step"], d["value"], r["frac"], r.get("binding", {}).get("frac"),
      {s["stage"]: (s["ms_per_step"], s.get("lane_utilisation"), s.get("valu_busy"), s.get("waves_per_simd")) for s in r.get("stages", [])}, d.get("config3", {}).get("ms_per_filtered_frame"), d.get("config3", {}).get("filter_ms_per_frame"), d.get("povs", {}).get("ms_per_step_avg"), d.get("cpu_baseline", {}).get("value"), (d.get("reference_layout") or {}).get("ms_per_step")))
except Exception as e: print("bench failed", e); print(open("gpurun_out/r04_bench.err").read()[-1500:])
PY
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages"
for spec in "20 2" "20 4" "20 8" "20 0"; do
  set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $1 --warmup 5 --emulate-world $2 $B > gpurun_out/r04_run28_s$1_w$2.json 2> gpurun_out/r04_run28_s$1_w$2.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r04_run28_s$1_w$2.json')); print('steps $1 emulate-world $2: %.4f ms/step %.1f Mrays/s | per iteration %s' % (d['ms_per_step'], d['value'], d['config'].get('submissions_per_iteration')))
except Exception as e: print('steps $1 world $2 failed', e); print(open('gpurun_out/r04_run28_s$1_w$2.err').read()[-600:])"
done
