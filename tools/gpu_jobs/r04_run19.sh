# round 4, nineteenth GPU session: the whole GPU suite on the new default build (sort at 8 waves, no SLP in shade / post, one append per shade round), then:
# the sort kernel with its material / roulette loads up front, the traversal unit without SLP, two refill thresholds
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_run19_pytest.log; echo "suite: $(tail -1 gpurun_out/r04_run19_pytest.log)"
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
for v in base sortearly trace_noslp nd8nw32 nd2nw8 base2; do
  unset GRT_DEVICE_LIB
  case $v in base|base2) ;; *) export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so;; esac
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r04_run19_$v.json 2>gpurun_out/r04_run19_$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r04_run19_$v.json")); r=d["roofline"]
    st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-12s %.4f ms/step  %.1f Mrays/s | trav %.4f sort %.4f diff %.4f plas %.4f gen %.4f acc %.4f" % ("$v", d["ms_per_step"], d["value"], st.get("traversal", 0), st.get("sort", 0), st.get("material_diffuse", 0), st.get("material_plastic", 0), st.get("generate", 0), st.get("accumulate", 0)))
except Exception as e: print("$v failed", e)
PY
done
