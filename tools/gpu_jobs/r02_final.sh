# Round-2 final validation at HEAD: whole GPU suite, smoke, the driver's bench command (+ counters, POVs, CPU baseline),
# its rocprofv3 kernel trace, the emulated tile splits, the other BASELINE configs.
mkdir -p gpurun_out
R=$PWD
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12) > gpurun_out/r02_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -18 gpurun_out/r02_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'])
print({k:r.get(k) for k in ('achieved','frac','launches','launch_ms','steady_state','time_share_of_step','traffic','counters','pmc_errors')})
print(d.get('povs',{}).get('ms_per_step_avg'), d.get('povs',{}).get('ms_per_step_stddev')); print(d['cpu_baseline'])
PY
tail -3 gpurun_out/r02_bench.err
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_final -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > $R/gpurun_out/r02_prof_final.log 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_final -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_bench_kernel_trace.txt 2>&1; python tools/rocpd_gaps.py $f 55 > gpurun_out/r02_bench_gaps.txt 2>&1; done; head -14 gpurun_out/r02_bench_kernel_trace.txt; cat gpurun_out/r02_bench_gaps.txt | head -8
B="--no-cpu-baseline --no-povs --no-pmc"
timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B > gpurun_out/r02_bench_k160.json 2>/dev/null
for W in 8 4 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B --emulate-world $W > gpurun_out/r02_emulated_rank${W}_k20.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world $W > gpurun_out/r02_emulated_rank${W}_k160.json 2>/dev/null
done
python - <<'PY'
import json
base={}
for k,f in (("k20","gpurun_out/r02_bench.json"),("k160","gpurun_out/r02_bench_k160.json")):
    base[k]=json.load(open(f))["ms_per_step"]; print("N=1", k, base[k])
for W in (2,4,8):
    for k in ("k20","k160"):
        try:
            d=json.load(open("gpurun_out/r02_emulated_rank%d_%s.json"%(W,k)))
            print("emulated rank 0 of %d, %s: %.4f ms/step -> bound %.2fx  frac %.3f launches %s" % (W,k,d["ms_per_step"],base[k]/d["ms_per_step"],d["roofline"]["frac"],d["roofline"].get("launches")))
        except Exception as e: print(W,k,"failed",e)
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_emu8_final -o bench -- python $R/bench.py --gpus 1 --steps 160 --warmup 8 $B --emulate-world 8 > /dev/null 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_emu8_final -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_emulated_rank8_kernel_trace.txt 2>&1; python tools/rocpd_gaps.py $f 60 > gpurun_out/r02_emulated_rank8_gaps.txt 2>&1; done
rm -rf gpurun_out/r02_prof_final gpurun_out/r02_prof_emu8_final
timeout 900 python tools/config_suite.py > gpurun_out/r02_config_suite.log 2>&1; grep "^config" gpurun_out/r02_config_suite.log
