# round 3: the driver's command as the driver runs it, its rocprofv3 kernel trace, the whole GPU suite, the config suite
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r03_gpu_suite.log; tail -6 gpurun_out/r03_gpu_suite.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err ) 2>&1 | grep real
( time timeout 900 python bench.py > gpurun_out/r03_bench_default_64_steps.json 2> gpurun_out/r03_bench_default.err ) 2>&1 | grep real
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r03_prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r03_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-stages > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_prof.log
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/r03_prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r03_bench_kernel_trace.txt 2>&1
python tools/rocpd_gaps.py $DB > gpurun_out/r03_bench_gaps.txt 2>&1
head -14 gpurun_out/r03_bench_gaps.txt
python - <<'PY'
import json
for name in ("r03_bench", "r03_bench_default_64_steps"):
    try:
        d = json.load(open("gpurun_out/%s.json" % name)); r = d["roofline"]
        print("%s: %.3f ms/step %.1f Mrays/s frac %.3f stream-frac %.3f | binding %s | l2 %s l1 %s | cpu %s" % (name, d["ms_per_step"], d["value"], r["frac"], r.get("frac_of_measured_stream", 0),
              r.get("binding", {}).get("frac"), r.get("binding", {}).get("l2_hit_rate"), r.get("binding", {}).get("l1_hit_rate"), d.get("cpu_baseline", {}).get("value")))
        print("   lanes %s valu busy %s traffic %s" % (r.get("counters", {}).get("valu_lane_utilisation"), r.get("counters", {}).get("valu_busy"), r.get("traffic")))
        for s in r.get("stages", []): print("   stage", json.dumps(s))
        c3 = d.get("config3", {}); print("   config3 %s ms/frame filter %s" % (c3.get("ms_per_filtered_frame"), c3.get("filter_ms_per_frame")))
    except Exception as e:
        print(name, "failed:", e)
PY
