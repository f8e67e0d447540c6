# round 4, twenty-third GPU session: the record of the round's build -- the whole GPU suite, the driver's command (complete line: counters, config 3, the nine
# points of view, the reference's layout, the CPU baseline), the rocprofv3 kernel trace of the same command, per-kernel counters; longer runs as bursts of 8;
# one rank of an 8-way split
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_run23_pytest.log; echo "suite: $(tail -1 gpurun_out/r04_run23_pytest.log)"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_bench.json")); r = d["roofline"]
    print("%.3f ms/step %.1f Mrays/s frac %.3f | binding %s | stages %s | config3 %s filter %s | povs %s | cpu %s | reference layout %s" % (d["ms_per_step"], d["value"], r["frac"], r.get("binding", {}).get("frac"),
      {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}, d.get("config3", {}).get("ms_per_filtered_frame"), d.get("config3", {}).get("filter_ms_per_frame"), d.get("povs", {}).get("ms_per_step_avg"), d.get("cpu_baseline", {}).get("value"), (d.get("reference_layout") or {}).get("ms_per_step")))
except Exception as e: print("bench failed", e); print(open("gpurun_out/r04_bench.err").read()[-1500:])
PY
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace && timeout 300 rocprofv3 --kernel-trace -d /tmp/ktrace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $B --no-stages > /dev/null 2>&1
cd $R
DB=$(find /tmp/ktrace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r04_bench_kernel_trace.txt 2>&1; head -12 gpurun_out/r04_bench_kernel_trace.txt
python tools/rocpd_gaps.py $DB 8 > gpurun_out/r04_bench_gaps.txt 2>&1; head -6 gpurun_out/r04_bench_gaps.txt
timeout 600 python tools/kernel_counters.py --steps 20 --warmup 5 > gpurun_out/r04_kernel_counters.txt 2> gpurun_out/r04_kernel_counters.err; head -14 gpurun_out/r04_kernel_counters.txt
for spec in "64 0" "160 0" "20 8" "160 8"; do
  set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $1 --warmup 5 --emulate-world $2 $B --no-stages > gpurun_out/r04_run23_s$1_w$2.json 2> gpurun_out/r04_run23_s$1_w$2.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r04_run23_s$1_w$2.json')); print('steps $1 emulate-world $2: %.4f ms/step %.1f Mrays/s' % (d['ms_per_step'], d['value']))
except Exception as e: print('steps $1 world $2 failed', e); print(open('gpurun_out/r04_run23_s$1_w$2.err').read()[-600:])"
done
