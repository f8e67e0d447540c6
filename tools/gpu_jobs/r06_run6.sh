# round 6, GPU session 6: the learned seating on device-built flattened trees (read back, seated, rt_update_nodes) -- the device BLAS tests, the static geometry tests
# (the reseat hooks sit in the same update()), then what the trees cost: tools/blas_bench.py (host tree | device tree as built | device tree seated), 32 steps each.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_blas.py tests/test_gpu_static_geometry.py tests/test_gpu_tlas.py -x -q 2>&1 | grep -v WARNING | tail -15
DEVICE_PRESPLIT=0.08 timeout 900 python tools/blas_bench.py 2>&1 | grep -v WARNING | tee gpurun_out/r06_blas_bench.txt
