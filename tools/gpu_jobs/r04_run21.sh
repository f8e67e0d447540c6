# round 4, twenty-first GPU session: the tiled a-trous passes with the neighbourhood loads in front of the barrier; 8 / 4 / 2 rows per workgroup;
# kernel_taa with kernel_taa_finalize folded in (third history image, pointers swapped per frame): parity, then config 3
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py::test_sponza_svgf_taa_with_a_moving_camera_at_full_size tests/test_gpu_parity.py -x -q 2>&1 | tail -12 > gpurun_out/r04_run21_pytest.log; tail -3 gpurun_out/r04_run21_pytest.log
B="--no-cpu-baseline --no-povs --no-pmc --no-reference-layout --no-stages"
for v in base atrous_ty4 atrous_ty2 untiled base2; do
  unset GRT_DEVICE_LIB BENCH_SVGF_TILES
  case $v in base|base2) ;; untiled) export BENCH_SVGF_TILES=0;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run21_$v.json 2>gpurun_out/r04_run21_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run21_$v.json")); c3 = d.get("config3") or {}
    print("%-10s %.4f ms/step | config3 %s ms per filtered frame, filter %s | %s" % ("$v", d["ms_per_step"], c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame"), [(k.get("kernel")[7:], k.get("ms_per_frame")) for k in c3.get("kernels", [])]))
except Exception as e: print("$v failed", e)
PY
done
