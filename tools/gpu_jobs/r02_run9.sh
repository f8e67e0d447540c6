mkdir -p gpurun_out
(time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "merged or small_pipelined or bench_spawns or tile_split or split_invariant") > gpurun_out/r02_run9_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r02_run9_tests.log
B="--no-cpu-baseline --no-povs --no-pmc"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r02_run9_n1.json 2>gpurun_out/r02_run9_n1.err
for W in 8 4 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world $W > gpurun_out/r02_run9_emu${W}_k20.json 2>gpurun_out/r02_run9_emu${W}.err
done
timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B > gpurun_out/r02_run9_n1_k160.json 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world 8 > gpurun_out/r02_run9_emu8_k160.json 2>/dev/null
python - <<'PY'
import json
for n in ("n1","emu8_k20","emu4_k20","emu2_k20","n1_k160","emu8_k160"):
    try:
        d=json.load(open("gpurun_out/r02_run9_%s.json"%n)); r=d["roofline"]
        print("%-10s %.4f ms/step  value %.1f  frac %.3f launches %s" % (n, d["ms_per_step"], d["value"], r["frac"], r.get("launches")))
    except Exception as e: print(n, "failed", e)
PY
