mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_tlas.py tests/test_gpu_parity.py -x -q -k "tlas or animated" 2>&1 | tail -3
GRT_DEVICE_LIB=$PWD/gpu-raytracer_amd/csrc/_variants/tlastimes/libgrt_device.so timeout 300 python -m pytest tests/test_gpu_tlas.py -x -q -s 2>&1 | grep "kernel_build_tlas" | awk '{k=$2" "$3; if (!(k in seen) || seen[k] < 2) {print; seen[k]++}}' | head -12
GRT_DEVICE_LIB=$PWD/gpu-raytracer_amd/csrc/_variants/tlastimes/libgrt_device.so timeout 200 python tools/animation_bench.py 2>&1 | grep "kernel_build_tlas" | tail -3
GRT_DEVICE_LIB=$PWD/gpu-raytracer_amd/csrc/_variants/tlastimes/libgrt_device.so ANIM_INSTANCES=4000 timeout 200 python tools/animation_bench.py 2>&1 | grep "kernel_build_tlas" | tail -2
timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_441.log 2>&1; grep -i "TLAS" gpurun_out/r02_animation_441.log
ANIM_INSTANCES=4000 timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_4000.log 2>&1; grep -i "TLAS" gpurun_out/r02_animation_4000.log
