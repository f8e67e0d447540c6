# round 3, first GPU session: instance entry before the node step, octant-ordered appends
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc"
for v in r02 early_noct default w5 r02 default; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r03_run1_$v.json 2>gpurun_out/r03_run1_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03_run1_$v.json")); r=d["roofline"]
    print("%-12s %.4f ms/step  value %.1f  frac %.4f steady %.4f  trace share %.3f  launch ms %s" % ("$v", d["ms_per_step"], d["value"], r["frac"], r["steady_state"]["frac"], r["time_share_of_step"], r["launch_ms"]))
except Exception as e: print("$v failed", e)
PY
done
unset GRT_DEVICE_LIB
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_run1_pytest.log; cat gpurun_out/r03_run1_pytest.log
