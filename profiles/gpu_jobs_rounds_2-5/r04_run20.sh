# round 4, twentieth GPU session: the a-trous passes with a workgroup's taps in LDS, rows `step` apart (rt_set_svgf_tiles): parity (bit-identical to the
# untiled passes; the oracle tests; the 1080p moving-camera frames), then config 3 with and without
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py::test_sponza_svgf_taa_with_a_moving_camera_at_full_size -x -q 2>&1 | tail -12 > gpurun_out/r04_run20_pytest.log; tail -5 gpurun_out/r04_run20_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc --no-reference-layout --no-stages"
for t in 1 0 1; do
  BENCH_SVGF_TILES=$t timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run20_tiles$t.json 2>gpurun_out/r04_run20_tiles$t.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run20_tiles$t.json")); c3 = d.get("config3") or {}
    print("tiles $t  %.4f ms/step | config3 %s ms per filtered frame, filter %s | %s" % (d["ms_per_step"], c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame"), [(k.get("kernel")[7:], k.get("ms_per_frame")) for k in c3.get("kernels", [])]))
except Exception as e: print("tiles $t failed", e)
PY
done
