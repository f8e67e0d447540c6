mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_exact or edge_cases or statistics or merged_wavefront_equals") > gpurun_out/r02_valu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_valu_tests.log
B="--no-cpu-baseline --no-povs --no-pmc"
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r02_valu_n1_$i.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r02_valu_n1_$i.json")); r=d["roofline"]
print("N=1 %.4f ms/step  value %.1f  frac %.4f steady %.4f  trace share %.3f" % (d["ms_per_step"], d["value"], r["frac"], r["steady_state"]["frac"], r["time_share_of_step"]))
PY
done
