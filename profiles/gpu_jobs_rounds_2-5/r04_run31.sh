# round 4, last GPU session: the whole GPU suite on the last commit, and the rocprofv3 kernel trace of the driver's command with the shipped kernels
mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_run31_pytest.log; echo "suite: $(tail -1 gpurun_out/r04_run31_pytest.log)"
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktrace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $B > $R/gpurun_out/r04_run31_bench.json 2> /dev/null
cd $R
DB=$(find /tmp/ktrace -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r04_bench_kernel_trace.txt 2>&1; head -12 gpurun_out/r04_bench_kernel_trace.txt
find /tmp/ktrace -name "*stats*" | head -5
for f in $(find /tmp/ktrace -name "*kernel_stats*.csv" | head -1); do head -12 $f > gpurun_out/r04_bench_kernel_stats.csv; done
python -c "
import json; d=json.load(open('gpurun_out/r04_run31_bench.json')); r=d['roofline']; print('under rocprofv3: %.3f ms/step; %d timed launches, avg %.4f ms' % (d['ms_per_step'], r['launches'], r['avg_launch_ms']))"
