mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc"
for v in waves7 waves6s8 waves6 waves7; do
  export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r02_sweep_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_sweep_$v.json")); r=d["roofline"]
print("%-10s %.4f ms/step  value %.1f  frac %.4f steady %.4f  trace share %.3f" % ("$v", d["ms_per_step"], d["value"], r["frac"], r["steady_state"]["frac"], r["time_share_of_step"]))
PY
done
for v in waves6 waves7; do
  GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world 8 > gpurun_out/r02_sweep_emu8_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r02_sweep_emu8_$v.json')); print('$v emu8 k160 %.4f ms/step' % d['ms_per_step'])"
done
