# round 4, eighth GPU session: the whole GPU suite on the tree as it stands, then the driver's bench command
mkdir -p gpurun_out
rm -f gpurun_out/parity_numbers.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_run8_pytest.log; tail -8 gpurun_out/r04_run8_pytest.log; cat gpurun_out/parity_numbers.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_run8_bench.json 2> gpurun_out/r04_run8_bench.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_run8_bench.json")); r=d["roofline"]
st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
print("bench %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f" % (d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0)))
print({k: d["roofline"].get(k) for k in ("achieved", "frac", "traffic", "kernel")}); print("reference_layout:", d.get("reference_layout")); print("cpu_baseline:", d.get("cpu_baseline"))
PY
