mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc"
for v in default mixed32m tri1 fetch256 sort4096 sort1024 default; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r02_sweep_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_sweep_$v.json")); r=d["roofline"]
print("%-10s %.4f ms/step  value %.1f  frac %.4f trace share %.3f  stages %s" % ("$v", d["ms_per_step"], d["value"], r["frac"], r["time_share_of_step"], d["config"]["stage_ms_per_step_one_frame_alone"]))
PY
done
