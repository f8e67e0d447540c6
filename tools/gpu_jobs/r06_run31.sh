# round 6, GPU session 31: which section of the full command leaves one of the nine points of view at 3-4 ms per step (a seating in the making beside it)?
mkdir -p gpurun_out
show() { python -c "
import json,sys
try:
    d=json.load(open('$1')); print('%-34s' % '$2', [(p['ms_per_step'], int(p['seating_in_the_making'])) for p in d['povs']['per_pov']])
except Exception as e: print('$2 failed', e)"; }
C="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config3 --no-reference-layout"
timeout 300 python bench.py $C --no-pmc --no-stages > gpurun_out/r06_run31_a.json 2>/dev/null; show gpurun_out/r06_run31_a.json "no pmc, no stages"
timeout 300 python bench.py $C --no-pmc > gpurun_out/r06_run31_b.json 2>/dev/null; show gpurun_out/r06_run31_b.json "no pmc, stages"
timeout 300 python bench.py $C --no-stages > gpurun_out/r06_run31_c.json 2>/dev/null; show gpurun_out/r06_run31_c.json "pmc, no stages"
timeout 300 python bench.py $C > gpurun_out/r06_run31_d.json 2>/dev/null; show gpurun_out/r06_run31_d.json "pmc, stages"
BENCH_RESEAT_DISTANCE=0 timeout 300 python bench.py $C > gpurun_out/r06_run31_e.json 2>/dev/null; show gpurun_out/r06_run31_e.json "pmc, stages, no re-seating"
