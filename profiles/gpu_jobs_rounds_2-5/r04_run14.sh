# round 4, fourteenth GPU session: the tile split's bound with this round's shade stage; a kernel timeline of one rank's burst
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-stages --no-reference-layout"
for spec in "0 20" "8 20" "4 20" "2 20" "8 160" "0 160"; do
  set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $2 --warmup 5 --emulate-world $1 $B > gpurun_out/r04_run14_w$1_s$2.json 2> gpurun_out/r04_run14_w$1_s$2.err
  python -c "
import json; d=json.load(open('gpurun_out/r04_run14_w$1_s$2.json')); print('emulate-world $1, $2 steps: %.4f ms/step' % d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace8 && timeout 300 rocprofv3 --kernel-trace -d /tmp/trace8 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world 8 $B > /dev/null 2>&1
cd $R
DB=$(find /tmp/trace8 -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB 9 > gpurun_out/r04_run14_timeline_w8.txt 2>&1; wc -l gpurun_out/r04_run14_timeline_w8.txt
python tools/rocpd_gaps.py $DB 8 > gpurun_out/r04_run14_gaps_w8.txt 2>&1; head -20 gpurun_out/r04_run14_gaps_w8.txt
