# round 5, GPU session 16 (the record of the round's last state): the whole GPU suite, smoke(), the driver's command, the kernel trace of the same command, rank 0's share of a
# 2 / 4 / 8-way split (bench.py --emulate-world; the timed region of a split runs without per-launch events now), BASELINE configs 3 / 4 / 5, the hint on the TLAS engine
mkdir -p gpurun_out; rm -f gpurun_out/parity_numbers.txt gpurun_out/parity_pixel_breakdown.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v WARNING | tail -25 > gpurun_out/r05_run16_pytest.log ) 2>&1 | grep real; tail -3 gpurun_out/r05_run16_pytest.log; grep -n "^FAILED" gpurun_out/r05_run16_pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_run16.json 2> gpurun_out/r05_bench_run16.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r05_bench_run16.json")); r = d["roofline"]; b = r.get("binding", {})
    print("%.3f ms/step %.1f Mrays/s | frac %s hbm_frac %s | binding %s" % (d["ms_per_step"], d["value"], r.get("frac"), r.get("hbm_frac"), b.get("utilisation_by_unit")))
    print("stages", {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, "| config3", (d.get("config3") or {}).get("ms_per_filtered_frame"), (d.get("config3") or {}).get("filter_ms_per_frame"), "| ref layout", (d.get("reference_layout") or {}).get("ms_per_step"), "| errors", r.get("pmc_errors"), "| nodes/tris", r.get("nodes_per_ray"), r.get("triangles_per_ray"), r.get("nodes_per_shadow_ray"), r.get("triangles_per_shadow_ray"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r05_bench_run16.err").read()[-2000:])
PY
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages"
for spec in "20 0" "20 2" "20 4" "20 8" "160 0" "160 8"; do
  set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $1 --warmup 5 --emulate-world $2 $B > gpurun_out/r05_run16_s$1_w$2.json 2> gpurun_out/r05_run16_s$1_w$2.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r05_run16_s$1_w$2.json')); print('steps $1 emulate-world $2: %.4f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']))
except Exception as e: print('steps $1 world $2 failed', e); print(open('gpurun_out/r05_run16_s$1_w$2.err').read()[-600:])" | tee -a gpurun_out/r05_run16_emulation.txt
done
timeout 600 python tools/config_suite.py 2>&1 | grep -v WARNING | tail -12 | tee gpurun_out/r05_config_suite.txt
for lib in "" "$PWD/gpu-raytracer_amd/csrc/_variants/prio_all/libgrt_device.so"; do
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-static 0 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout > gpurun_out/r05_run16_ref.json 2> gpurun_out/r05_run16_ref.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r05_run16_ref.json')); st={s['stage']: s['ms_per_step'] for s in d['roofline'].get('stages', [])}; print('reference layout, lib [%s]: %.4f ms/step, traversal %.4f' % ('$lib'[-40:], d['ms_per_step'], st.get('traversal', 0)))
except Exception as e: print('ref layout failed', e); print(open('gpurun_out/r05_run16_ref.err').read()[-600:])" | tee -a gpurun_out/r05_run16_prio_all.txt
done
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find gpurun_out/r05_prof -name "*.db" | head -1) 2>/dev/null | head -24 | tee gpurun_out/r05_bench_kernel_trace.txt
rm -rf gpurun_out/r05_prof
