mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_tlas.py tests/test_gpu_parity.py -x -q -k "tlas or animated or merged" ) > gpurun_out/r02_run6_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r02_run6_tests.log
ANIM_INSTANCES=1500 timeout 400 python tools/animation_bench.py > gpurun_out/r02_animation_1500.log 2>&1; echo "anim1500 rc=$?"; tail -6 gpurun_out/r02_animation_1500.log
R=$PWD; cd /tmp && export TMPDIR=/tmp && ANIM_INSTANCES=4000 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_anim4000 -o anim -- python $R/tools/animation_bench.py > /dev/null 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_anim4000 -name "*.db"); do python tools/rocpd_summary.py $f 2>&1 | grep -i "build_tlas\|kernel  " ; done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_prof_anim441 -o anim -- python $R/tools/animation_bench.py > $R/gpurun_out/r02_animation_441.log 2>&1
cd $R; for f in $(find gpurun_out/r02_prof_anim441 -name "*.db"); do python tools/rocpd_summary.py $f 2>&1 | grep -i "build_tlas" ; done; tail -5 gpurun_out/r02_animation_441.log
for v in default fetch256 tri3 nw32 mixed4m mixed16m waves4; do
  if [ $v = default ]; then unset GRT_DEVICE_LIB; else export GRT_DEVICE_LIB=$R/gpu-raytracer_amd/csrc/_variants/$v/libgrt_device.so; fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc > gpurun_out/r02_variant_$v.json 2>/dev/null
  timeout 300 python bench.py --gpus 1 --steps 40 --warmup 4 --no-cpu-baseline --no-povs --no-pmc --emulate-world 8 > gpurun_out/r02_variant_emu8_$v.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open('gpurun_out/r02_variant_$v.json')); e=json.load(open('gpurun_out/r02_variant_emu8_$v.json')); r=d['roofline']
print('%-10s N=1 %.3f ms/step frac %.4f steady %.4f | emu8 %.3f ms/step steady %.4f' % ('$v', d['ms_per_step'], r['frac'], r['steady_state']['frac'], e['ms_per_step'], e['roofline']['steady_state']['frac']))"
done
unset GRT_DEVICE_LIB
(time ORACLE_PROFILE=1 timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=8 2>&1 | grep -v "^WARNING" ) > gpurun_out/r02_gputest_6.log 2>&1; echo "suite rc=$?"; grep "oracle\]" gpurun_out/r02_gputest_6.log | head -6; tail -14 gpurun_out/r02_gputest_6.log
