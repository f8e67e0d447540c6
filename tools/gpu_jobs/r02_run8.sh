mkdir -p gpurun_out
R=$PWD
for v in default shade3; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r02_variant_$v.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 40 --warmup 4 --no-cpu-baseline --no-povs --no-pmc --emulate-world 8 > gpurun_out/r02_variant_emu8_$v.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open('gpurun_out/r02_variant_$v.json')); e=json.load(open('gpurun_out/r02_variant_emu8_$v.json')); r=d['roofline']
print('%-10s N=1 %.3f ms/step frac %.4f steady %.4f stages %s | emu8 %.3f ms/step steady %.4f' % ('$v', d['ms_per_step'], r['frac'], r['steady_state']['frac'], d['config']['stage_ms_per_step_one_frame_alone'], e['ms_per_step'], e['roofline']['steady_state']['frac']))"
done
unset GRT_DEVICE_LIB
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_materials_svgf.py -x -q -k "merged or glass or sponza or cornell_render or animated" ) > gpurun_out/r02_run8_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r02_run8_tests.log
timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_441.log 2>&1; grep TLAS gpurun_out/r02_animation_441.log
ANIM_INSTANCES=4000 timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_4000.log 2>&1; grep TLAS gpurun_out/r02_animation_4000.log
