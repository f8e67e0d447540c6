mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r02_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -14 gpurun_out/r02_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="--no-cpu-baseline --no-povs --no-pmc"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r02_reval_n1.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world 8 > gpurun_out/r02_reval_emu8_k20.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world 8 > gpurun_out/r02_reval_emu8_k160.json 2>/dev/null
python - <<'PY'
import json
for n in ("n1","emu8_k20","emu8_k160"):
    d=json.load(open("gpurun_out/r02_reval_%s.json"%n)); r=d["roofline"]
    print("%-10s %.4f ms/step  value %.1f  frac %.3f launches %s" % (n, d["ms_per_step"], d["value"], r["frac"], r.get("launches")))
PY
