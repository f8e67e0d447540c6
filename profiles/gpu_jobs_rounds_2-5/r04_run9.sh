# round 4, ninth GPU session: BC1 textures expanded at upload + material kernels without the per-fetch decode, sincosf, light tables in LDS,
# one random_path per hit, shared plastic terms, queue fields fetched beside the slot table; shadow rays far end first.
mkdir -p gpurun_out
R=$PWD
rm -f gpurun_out/parity_numbers.txt
tools/microbench/sincos_check > gpurun_out/r04_run9_sincos.txt 2>&1; cat gpurun_out/r04_run9_sincos.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_run9_pytest.log; tail -5 gpurun_out/r04_run9_pytest.log; cp gpurun_out/parity_numbers.txt gpurun_out/r04_run9_parity_numbers.txt 2>/dev/null
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base nearfirst compressed base2; do
  unset GRT_DEVICE_LIB; X=""
  [ $v = nearfirst ] && export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/nearfirst/libgrt_device.so
  [ $v = compressed ] && X="--expand-textures 0"
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B $X > gpurun_out/r04_run9_$v.json 2>gpurun_out/r04_run9_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run9_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f gen %.4f acc %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0), st.get("accumulate", 0)))
except Exception as e: print("$v failed", e)
PY
done
