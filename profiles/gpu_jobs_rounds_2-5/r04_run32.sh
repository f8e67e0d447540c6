# round 4: kernel timeline of the timed burst (where the idle time between its ~57 launches goes), with and without the per-launch events of the bench
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages"
cd /tmp && export TMPDIR=/tmp
for ev in 0 1; do
  rm -rf /tmp/kt$ev
  if [ $ev = 1 ]; then export BENCH_NO_LAUNCH_TIMING=1; else unset BENCH_NO_LAUNCH_TIMING; fi
  timeout 300 rocprofv3 --kernel-trace -d /tmp/kt$ev -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 $B > $R/gpurun_out/r04_run32_bench$ev.json 2> /dev/null
  DB=$(find /tmp/kt$ev -name "*.db" | head -1)
  python $R/tools/rocpd_timeline.py $DB 60 > $R/gpurun_out/r04_run32_timeline$ev.txt 2>&1
  python -c "
import json; d=json.load(open('$R/gpurun_out/r04_run32_bench$ev.json')); print('launch events %s: %.4f ms/step' % ('off' if $ev else 'on', d['ms_per_step']))"
done
wc -l $R/gpurun_out/r04_run32_timeline*.txt
