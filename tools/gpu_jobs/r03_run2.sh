# round 3, second GPU session: new parity tests, SVGF with hardware exp2/log2, the full bench line, shade experiments
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_kernels.py tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -25 > gpurun_out/r03_run2_pytest.log; cat gpurun_out/r03_run2_pytest.log
( time timeout 900 python bench.py > gpurun_out/r03_run2_bench.json 2> gpurun_out/r03_run2_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_run2_bench.json"))
    r = d["roofline"]
    print("bench: %.3f ms/step %.1f Mrays/s frac %.3f of peak, %.3f of stream; binding %s" % (d["ms_per_step"], d["value"], r["frac"], r.get("frac_of_measured_stream", 0), json.dumps(r.get("binding"))))
    for s in r.get("stages", []): print("  stage", json.dumps(s))
    c3 = d.get("config3", {})
    print("config3: %s ms per frame, filter %s ms" % (c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame")))
    for k in c3.get("kernels", []): print("  ", json.dumps(k))
    print("pmc errors:", r.get("pmc_errors"))
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/r03_run2_bench.err").read()[-2000:])
PY
timeout 600 python tools/shade_experiments.py 32 2>&1 | grep -v WARNING | tee gpurun_out/r03_run2_shade.log
