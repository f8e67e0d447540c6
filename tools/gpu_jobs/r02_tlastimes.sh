mkdir -p gpurun_out
export GRT_DEVICE_LIB=$PWD/gpu-raytracer_amd/csrc/_variants/tlastimes/libgrt_device.so
timeout 300 python -m pytest tests/test_gpu_tlas.py -x -q -s 2>&1 | grep "kernel_build_tlas\|passed\|failed" | awk '{k=$2" "$3; if (!(k in seen) || seen[k] < 3) {print; seen[k]++}}' | head -40
timeout 200 python tools/animation_bench.py 2>&1 | grep "kernel_build_tlas" | tail -5
ANIM_INSTANCES=4000 timeout 200 python tools/animation_bench.py 2>&1 | grep "kernel_build_tlas" | tail -3
