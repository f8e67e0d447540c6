# round 3, after the flattening: the whole GPU suite, the driver's command as the driver runs it (and the reference layout beside it),
# the default command, the rocprofv3 kernel trace of the driver's command, the config suite
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -60 > gpurun_out/r03_gpu_suite.log; tail -4 gpurun_out/r03_gpu_suite.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err ) 2>&1 | grep real
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-static 0 --no-cpu-baseline --no-povs --no-config3 > gpurun_out/r03_bench_reference_layout.json 2> gpurun_out/r03_bench_reference_layout.err
( time timeout 900 python bench.py > gpurun_out/r03_bench_64_steps.json 2> gpurun_out/r03_bench_64_steps.err ) 2>&1 | grep real
timeout 600 python tools/config_suite.py 2>&1 | grep "^config" | tee gpurun_out/r03_config_suite.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r03_prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r03_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-stages > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_prof.log
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/r03_prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r03_bench_kernel_trace.txt 2>&1
python tools/rocpd_gaps.py $DB 40 > gpurun_out/r03_bench_gaps.txt 2>&1
head -8 gpurun_out/r03_bench_gaps.txt
python - <<'PY'
import json
for name in ("r03_bench", "r03_bench_reference_layout", "r03_bench_64_steps"):
    try:
        d = json.load(open("gpurun_out/%s.json" % name)); r = d["roofline"]
        print("%s: %.3f ms/step %.1f Mrays/s frac %.3f stream-frac %.3f | binding %s | l2 %s l1 %s | cpu %s" % (name, d["ms_per_step"], d["value"], r["frac"], r.get("frac_of_measured_stream", 0),
              r.get("binding", {}).get("frac"), r.get("binding", {}).get("l2_hit_rate"), r.get("binding", {}).get("l1_hit_rate"), d.get("cpu_baseline", {}).get("value")))
        print("   lanes %s issue %s traffic %s achieved %s" % (r.get("binding", {}).get("lane_utilisation"), r.get("binding", {}).get("issue_slots_used"), r.get("traffic"), r.get("achieved")))
        for s in r.get("stages", []): print("   stage", json.dumps(s))
        c3 = d.get("config3", {}); print("   config3 %s ms/frame filter %s" % (c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame")))
        print("   povs", [k for k in d["config"] if "pov" in k], d["config"].get("ms_per_step_at_the_references_povs"))
    except Exception as e:
        print(name, "failed:", e)
PY
