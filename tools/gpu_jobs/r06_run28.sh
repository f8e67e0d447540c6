# round 6, GPU session 28: kernel_svgf_finalize's work done by the last (tiled) a-trous pass -- SVGF tests, then config 3 with and without (variant nofuse)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py tests/test_gpu_rccl.py -m gpu -x -q 2>&1 | tail -2
V=$PWD/gpu-raytracer_amd/csrc/_variants
for name in fused nofuse fused2 nofuse2; do
  lib=""; case $name in fused*) ;; nofuse*) lib="$V/nofuse/libgrt_device.so";; esac
  GRT_DEVICE_LIB=$lib timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-povs --no-pmc --no-reference-layout --no-stages 2>/dev/null > gpurun_out/r06_run28_$name.json
  python -c "
import json; d=json.load(open('gpurun_out/r06_run28_$name.json')); c=d['config3']; print('%-8s step %.4f | config3 %.4f ms per filtered frame, filter %.4f | %s' % ('$name', d['ms_per_step'], c['ms_per_filtered_frame'], c['filter_ms_per_frame'], [(k['kernel'][12:], k['ms_per_frame'], k['frac_unique']) for k in c['kernels']]))"
done
