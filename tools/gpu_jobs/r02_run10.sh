mkdir -p gpurun_out
(time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "merged or small_pipelined or bench_spawns or tile_split or split_invariant or sample_batches or accumulation") > gpurun_out/r02_run10_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r02_run10_tests.log
B="--no-cpu-baseline --no-povs --no-pmc"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r02_run10_n1.json 2>gpurun_out/r02_run10_n1.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world 8 > gpurun_out/r02_run10_emu8_k20.json 2>gpurun_out/r02_run10_emu8.err
timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world 8 > gpurun_out/r02_run10_emu8_k160.json 2>/dev/null
python - <<'PY'
import json
for n in ("n1","emu8_k20","emu8_k160"):
    try:
        d=json.load(open("gpurun_out/r02_run10_%s.json"%n)); r=d["roofline"]
        print("%-10s %.4f ms/step  value %.1f  frac %.3f launches %s" % (n, d["ms_per_step"], d["value"], r["frac"], r.get("launches")))
    except Exception as e: print(n, "failed", e)
PY
cd /tmp && export TMPDIR=/tmp
for K in 160 20; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_emu8_k$K -o emu8 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps $K --warmup 8 --no-cpu-baseline --no-povs --no-pmc --emulate-world 8 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_emu8_k160 | head
python tools/rocpd_gaps.py $(find gpurun_out/prof_emu8_k160 -name "*.db" | head -1) 60 > gpurun_out/r02_run10_emu8_k160_kernels.txt 2>&1 || true
python tools/rocpd_gaps.py $(find gpurun_out/prof_emu8_k20 -name "*.db" | head -1) 10 > gpurun_out/r02_run10_emu8_k20_kernels.txt 2>&1 || true
cat gpurun_out/r02_run10_emu8_k20_kernels.txt
rm -rf gpurun_out/prof_emu8_k160 gpurun_out/prof_emu8_k20
head -30 gpurun_out/r02_run10_emu8_k160_kernels.txt
