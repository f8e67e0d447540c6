mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_tlas.py tests/test_gpu_parity.py tests/test_gpu_materials_svgf.py -x -q -k "tlas or merged or tile_split or under_the_tile" ) > gpurun_out/r02_run4_new.log 2>&1; echo "new tests rc=$?"; tail -12 gpurun_out/r02_run4_new.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_run4.json 2> gpurun_out/r02_bench_run4.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_run4.json')); r=d['roofline']
print(d['value'], d['ms_per_step']); print({k:r.get(k) for k in ('achieved','frac','launches','launch_ms','steady_state','time_share_of_step','traffic','traffic_detail','counters','pmc_errors')}); print(d.get('povs',{}).get('ms_per_step_avg'), d.get('povs',{}).get('ms_per_step_stddev')); print(d['cpu_baseline'])"
tail -3 gpurun_out/r02_bench_run4.err
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_run4 -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > $R/gpurun_out/r02_prof_run4.log 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_run4 -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_prof_run4_summary.txt 2>&1; done; head -14 gpurun_out/r02_prof_run4_summary.txt
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 8 > gpurun_out/r02_emu8_run4.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_emu8_run4.json')); r=d['roofline']; print('emu8 merged', d['value'], d['ms_per_step'], r.get('launches'), r.get('launch_ms'), r.get('steady_state'), r.get('time_share_of_step'), d['config']['stage_ms_per_step_one_frame_alone'])"
BENCH_SCHEDULER=slots timeout 600 python bench.py --gpus 1 --steps 40 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 8 > gpurun_out/r02_emu8_slots_run4.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_emu8_slots_run4.json')); print('emu8 slots', d['value'], d['ms_per_step'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_emu8 -o bench -- python $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-pmc --emulate-world 8 > $R/gpurun_out/r02_prof_emu8.log 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_emu8 -name "*.db"); do python tools/rocpd_summary.py $f > gpurun_out/r02_prof_emu8_summary.txt 2>&1; python tools/rocpd_gaps.py $f > gpurun_out/r02_prof_emu8_gaps.txt 2>&1; done; head -16 gpurun_out/r02_prof_emu8_summary.txt; cat gpurun_out/r02_prof_emu8_gaps.txt | tail -12
timeout 400 python tools/animation_bench.py > gpurun_out/r02_animation_run4.log 2>&1; tail -6 gpurun_out/r02_animation_run4.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_anim -o anim -- python $R/tools/animation_bench.py > /dev/null 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_anim -name "*.db"); do python tools/rocpd_summary.py $f 2>&1 | grep -i "build_tlas\|kernel  " ; done
timeout 600 python tools/config_suite.py > gpurun_out/r02_config_suite.log 2>&1; tail -6 gpurun_out/r02_config_suite.log
