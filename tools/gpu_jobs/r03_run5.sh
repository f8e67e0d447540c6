# round 3, fifth GPU session: device BLAS build (tests + timing), SVGF variance-pair images, whole suite
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_blas.py -x -q 2>&1 | tail -30 > gpurun_out/r03_run5_blas.log; tail -30 gpurun_out/r03_run5_blas.log
timeout 300 python tools/blas_bench.py 2>&1 | grep -v WARNING | tee gpurun_out/r03_run5_blas_bench.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_blas.py 2>&1 | tail -15 > gpurun_out/r03_run5_pytest.log; tail -8 gpurun_out/r03_run5_pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r03_run5_bench.json 2> gpurun_out/r03_run5_bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r03_run5_bench.json")); c3 = d["config3"]
    print("bench 20 steps: %.3f ms/step; config3: %.3f ms per frame, filter %.4f ms: %s" % (d["ms_per_step"], c3["ms_per_filtered_frame"], c3["filter_ms_per_frame"], " ".join("%s %.4f" % (k["kernel"][7:], k["ms_per_frame"]) for k in c3["kernels"])))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r03_run5_bench.err").read()[-1500:])
PY
