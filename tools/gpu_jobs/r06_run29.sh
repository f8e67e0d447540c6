# round 6, GPU session 29: waves per SIMD of the diffuse material kernel alone (RT_SHADE_WAVES_DIFFUSE; shipped 4 = 125 registers)
mkdir -p gpurun_out
V=$PWD/gpu-raytracer_amd/csrc/_variants
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for name in default dw3 dw5 dw6 default2 dw5b; do
  lib=""; case $name in default*) ;; dw5b) lib="$V/dw5/libgrt_device.so";; *) lib="$V/$name/libgrt_device.so";; esac
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); st={s['stage']: s['ms_per_step'] for s in d['roofline']['stages']}; print('%-9s %.4f ms/step  diffuse %.4f plastic %.4f sort %.4f' % ('$name', d['ms_per_step'], st['material_diffuse'], st['material_plastic'], st['sort']))"
done
