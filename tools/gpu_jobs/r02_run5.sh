mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_materials_svgf.py tests/test_gpu_tlas.py -x -q -k "merged or glass or sponza_render or cornell_render or statistics or feature_toggles or edge_sizes or device_tlas_equals" ) > gpurun_out/r02_run5_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r02_run5_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r02_bench_run5.json 2>/dev/null; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_run5.json')); r=d['roofline']
print(d['value'], d['ms_per_step']); print({k:r.get(k) for k in ('achieved','frac','launches','launch_ms','steady_state','time_share_of_step')})"
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 8 > gpurun_out/r02_emu8_run5.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_emu8_run5.json')); r=d['roofline']; print('emu8', d['value'], d['ms_per_step'], r.get('launches'), r.get('launch_ms'), r.get('steady_state'), r.get('time_share_of_step'))"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 8 > gpurun_out/r02_emu8_20_run5.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_emu8_20_run5.json')); print('emu8 20 steps', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 2 > gpurun_out/r02_emu2_run5.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_emu2_run5.json')); print('emu2 20 steps', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 4 > gpurun_out/r02_emu4_run5.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_emu4_run5.json')); print('emu4 20 steps', d['value'], d['ms_per_step'])"
ANIM_INSTANCES=4000 timeout 400 python tools/animation_bench.py > gpurun_out/r02_animation_4000.log 2>&1; tail -7 gpurun_out/r02_animation_4000.log
ANIM_INSTANCES=1500 timeout 400 python tools/animation_bench.py > gpurun_out/r02_animation_1500.log 2>&1; tail -6 gpurun_out/r02_animation_1500.log
