mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_run2_smoke.log 2>&1; echo "smoke rc=$?"
(time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "merged" ) > gpurun_out/r02_run2_merged.log 2>&1; echo "merged rc=$?"; tail -15 gpurun_out/r02_run2_merged.log
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=12) > gpurun_out/r02_gputest_2.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/r02_gputest_2.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_merged.json 2> gpurun_out/r02_bench_merged.err; echo "bench rc=$?"; head -c 3000 gpurun_out/r02_bench_merged.json; tail -3 gpurun_out/r02_bench_merged.err
BENCH_SCHEDULER=slots timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_slots.json 2> gpurun_out/r02_bench_slots.err; echo "bench slots rc=$?"; head -c 1200 gpurun_out/r02_bench_slots.json
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_merged_40.json 2>/dev/null; head -c 600 gpurun_out/r02_bench_merged_40.json
timeout 600 python bench.py --gpus 1 --steps 32 --warmup 4 --no-cpu-baseline --emulate-world 8 > gpurun_out/r02_bench_emu8.json 2>/dev/null; head -c 600 gpurun_out/r02_bench_emu8.json
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_merged -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r02_prof_merged.log 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_merged -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_prof_merged_summary.txt 2>&1; done; head -14 gpurun_out/r02_prof_merged_summary.txt
