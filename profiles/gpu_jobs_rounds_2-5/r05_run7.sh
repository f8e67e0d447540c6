# round 5, seventh GPU session: what the 0.04 ms per step between the probe's rank (0.249) and bench.py --emulate-world 8 (0.290) is: the per-launch HIP events or the exchange
mkdir -p gpurun_out
B="--gpus 1 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages"
for v in "" "BENCH_NO_LAUNCH_TIMING=1" "BENCH_NO_GATHER=1" "BENCH_NO_LAUNCH_TIMING=1 BENCH_NO_GATHER=1" "BENCH_EMULATE_THROUGH_TORCH=1"; do
  env $v timeout 300 python bench.py $B > gpurun_out/r05_run7.json 2> gpurun_out/r05_run7.err
  python -c "
import json
try:
    d=json.load(open('gpurun_out/r05_run7.json')); print('%-50s %.4f ms/step %.1f Mrays/s' % ('$v', d['ms_per_step'], d['value']))
except Exception as e: print('$v failed', e); print(open('gpurun_out/r05_run7.err').read()[-600:])" | tee -a gpurun_out/r05_emulation_overheads.txt
done
