mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q --durations=30) > gpurun_out/r02_gputest_1.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_head.json 2> gpurun_out/r02_bench_head.err
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_head -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r02_prof_head.log 2>&1
cd $R; ls -R gpurun_out/r02_prof_head | head -20
for f in $(find gpurun_out/r02_prof_head -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_prof_head_summary.txt 2>&1; done
tail -5 gpurun_out/r02_gputest_1.log; cat gpurun_out/r02_bench_head.json | head -c 1500
