# round 6, GPU session 37 (the round's last state): the driver's command with every section on the build with the endgame regions + 8 x 8 primary-ray patches; rocprofv3 --kernel-trace --stats of the same command; rank 0 of 8
mkdir -p gpurun_out
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_run37.json 2> gpurun_out/r06_bench_run37.err ) 2>&1 | grep real
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_run37_pytest.log 2>&1; tail -3 gpurun_out/r06_run37_pytest.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r06_bench_run37.json")); r = d["roofline"]; b = r.get("binding", {})
    print("%.3f ms/step %.1f Mrays/s | bound %s frac %s (%s %s of %s) | alg/hbm %s hbm_frac %s | binding %s" % (d["ms_per_step"], d["value"], r.get("bound"), r.get("frac"), r.get("achieved"), r.get("unit"), r.get("peak"), r.get("algorithmic_bytes_over_hbm_peak"), r.get("hbm_frac"), b.get("utilisation_by_unit")))
    print("l1", {k: v for k, v in (b.get("l1") or {}).items() if k not in ("peak_derivation", "clocked_definition")})
    print("stages", {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, "| config3", (d.get("config3") or {}).get("ms_per_filtered_frame"), (d.get("config3") or {}).get("filter_ms_per_frame"), "| ref layout", (d.get("reference_layout") or {}).get("ms_per_step"), "| no-viewpoint seating", d.get("ms_per_step_seating_without_viewpoint"), "| povs", (d.get("povs") or {}).get("ms_per_step_avg"), "| errors", r.get("pmc_errors"), "| nodes/tris", r.get("nodes_per_ray"), r.get("triangles_per_ray"), r.get("nodes_per_shadow_ray"), r.get("triangles_per_shadow_ray"))
    print("counters", r.get("counters")); print("cpu", d.get("cpu_baseline"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r06_bench_run37.err").read()[-3000:])
PY
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find gpurun_out/r06_prof -name "*.db" | head -1) 2>/dev/null | head -24 | tee gpurun_out/r06_bench_kernel_trace.txt
rm -rf gpurun_out/r06_prof
for W in 2 4 8; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world $W --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('world $W: %.4f ms/step' % d['ms_per_step'])"; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
