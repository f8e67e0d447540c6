# round 5, third GPU session (second attempt: the first stopped at a check of the test itself, its builder numbers are in profiles/r05_device_blas_presplit.txt): (1) the flattened scene's engine with 32-bit offsets as the default, the SVGF variance pass over the listed young pixels,
# early split clipping in front of the device BLAS build -- the tests that cover them; (2) device-built flattened Sponza with and without the
# pre-split against the host tree (tools/blas_bench.py); (3) config 3
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_blas.py tests/test_gpu_materials_svgf.py tests/test_gpu_static_geometry.py tests/test_gpu_full_size.py::test_sponza_svgf_taa_with_a_moving_camera_at_full_size tests/test_gpu_parity.py -x -q --durations=6 2>&1 | tail -14 ) 2>&1 | tail -16
DEVICE_PRESPLIT=0.3,0.2,0.15,0.1,0.075 timeout 900 python tools/blas_bench.py 2>&1 | grep device_blas | tee gpurun_out/r05_run3_blas_bench.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-reference-layout --no-stages > gpurun_out/r05_run3_config3.json 2> gpurun_out/r05_run3_config3.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_run3_config3.json")); c3 = d.get("config3") or {}
    print("%.4f ms/step | config3 %s ms per filtered frame, filter %s | %s" % (d["ms_per_step"], c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame"), [(k.get("kernel")[7:], k.get("ms_per_frame")) for k in c3.get("kernels", [])]))
except Exception as e: print("config3 failed", e); print(open("gpurun_out/r05_run3_config3.err").read()[-800:])
PY
