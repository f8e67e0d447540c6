mkdir -p gpurun_out
(time timeout 600 python -m pytest tests/test_gpu_tlas.py tests/test_gpu_parity.py -x -q -k "tlas or merged" ) > gpurun_out/r02_run3_new.log 2>&1; echo "new tests rc=$?"; tail -30 gpurun_out/r02_run3_new.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_2pipe.json 2> gpurun_out/r02_bench_2pipe.err; echo "bench rc=$?"; head -c 2500 gpurun_out/r02_bench_2pipe.json; tail -3 gpurun_out/r02_bench_2pipe.err
GRT_STREAM_PIPELINES=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs > gpurun_out/r02_bench_1pipe.json 2>/dev/null; echo "bench 1pipe rc=$?"; head -c 500 gpurun_out/r02_bench_1pipe.json; echo
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-povs > gpurun_out/r02_bench_2pipe_40.json 2>/dev/null; head -c 500 gpurun_out/r02_bench_2pipe_40.json; echo
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --emulate-world 8 > gpurun_out/r02_bench_emu8_2pipe.json 2>/dev/null; head -c 500 gpurun_out/r02_bench_emu8_2pipe.json; echo
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --emulate-world 4 > gpurun_out/r02_bench_emu4_2pipe.json 2>/dev/null; head -c 400 gpurun_out/r02_bench_emu4_2pipe.json; echo
timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation.log 2>&1; tail -8 gpurun_out/r02_animation.log
(time ORACLE_PROFILE=1 timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r02_gputest_3.log 2>&1; echo "suite rc=$?"; grep "oracle\]" gpurun_out/r02_gputest_3.log | head -12; tail -16 gpurun_out/r02_gputest_3.log
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_2pipe -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-povs > $R/gpurun_out/r02_prof_2pipe.log 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_2pipe -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_prof_2pipe_summary.txt 2>&1; done; head -14 gpurun_out/r02_prof_2pipe_summary.txt
cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r02_pmc_fetch -o bench -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-povs > $R/gpurun_out/r02_pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
cd $R; for f in $(find gpurun_out/r02_pmc_fetch -name "*.db"); do python tools/rocpd_summary.py $f --counters 2>&1 | grep -i "trace_stream\|columns" | head -8; done
