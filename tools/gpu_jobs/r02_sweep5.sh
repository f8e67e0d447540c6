mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc"
for v in default g8k default g8k; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world 8 > gpurun_out/r02_sweep_e8_$v.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world 8 > gpurun_out/r02_sweep_e8l_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_sweep_e8_$v.json")); e=json.load(open("gpurun_out/r02_sweep_e8l_$v.json"))
print("%-10s emu8 k20 %.4f ms/step   k160 %.4f ms/step" % ("$v", d["ms_per_step"], e["ms_per_step"]))
PY
done
