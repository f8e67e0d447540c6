# round 3, sixth GPU session: SVGF reproject/variance fusion, fast division in the shade kernels (parity?), 8-rank emulation
mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py tests/test_gpu_blas.py tests/test_gpu_widening.py -q 2>&1 | tail -12 > gpurun_out/r03_run6_pytest.log; tail -6 gpurun_out/r03_run6_pytest.log
export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/shadefast/libgrt_device.so
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03_run6_pytest_shadefast.log; tail -12 gpurun_out/r03_run6_pytest_shadefast.log
unset GRT_DEVICE_LIB
B="--no-cpu-baseline --no-povs --no-pmc"
for v in default shadefast default shadefast; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r03_run6_$v.json 2>gpurun_out/r03_run6_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03_run6_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    c3 = d.get("config3", {})
    print("%-10s %.4f ms/step | trav %.4f sort %.4f diff %.4f plas %.4f | config3 %.3f ms/frame filter %.4f (%s)" % ("$v", d["ms_per_step"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), c3.get("ms_per_filtered_frame", 0), c3.get("filter_ms_per_frame", 0), " ".join("%.3f" % k["ms_per_frame"] for k in c3.get("kernels", []))))
except Exception as e: print("$v failed", e)
PY
done
unset GRT_DEVICE_LIB
for steps in 20 160; do
  timeout 300 python bench.py --gpus 1 --steps $steps --warmup 5 $B --no-config3 --no-stages --emulate-world 8 > gpurun_out/r03_run6_emu8_$steps.json 2>gpurun_out/r03_run6_emu8_$steps.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r03_run6_emu8_$steps.json')); print('emulated rank 0 of 8, $steps steps: %.4f ms/step' % d['ms_per_step'])
except Exception as e: print('emu failed', e)"
done
timeout 300 python bench.py --gpus 1 --steps 160 --warmup 5 $B --no-config3 --no-stages > gpurun_out/r03_run6_k160.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_run6_k160.json')); print('whole frame, 160 steps: %.4f ms/step' % d['ms_per_step'])"
