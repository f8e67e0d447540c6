# round 3 wrap-up: whole GPU suite at HEAD, config suite (configs 3-5), kernel trace of the driver's command with its timed region
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -30 > gpurun_out/r03_gpu_suite.log; tail -14 gpurun_out/r03_gpu_suite.log
timeout 600 python tools/config_suite.py 2>&1 | grep "^config" | tee gpurun_out/r03_config_suite.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r03_prof && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r03_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-stages > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_prof.log
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/r03_prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r03_bench_kernel_trace.txt 2>&1
python tools/rocpd_gaps.py $DB 47 > gpurun_out/r03_bench_gaps.txt 2>&1
cat gpurun_out/r03_bench_gaps.txt
python -c "
import json; d=json.load(open('gpurun_out/r03_prof_bench.json')); print('under rocprofv3: %.3f ms/step' % d['ms_per_step'])"
