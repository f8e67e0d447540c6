# round 6, GPU session 7: device BLAS / static geometry / TLAS tests after the seating of device-built trees
timeout 1500 python -m pytest tests/test_gpu_blas.py tests/test_gpu_static_geometry.py tests/test_gpu_tlas.py -x -q 2>&1 | grep -v WARNING | tail -15
