mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_blas.py -q 2>&1 | tail -40 > gpurun_out/r03_run7_blas.log; grep -n "^E \|passed\|failed" gpurun_out/r03_run7_blas.log | head -20
timeout 300 python tools/blas_bench.py 2>&1 | grep -v WARNING | tee gpurun_out/r03_run7_blas_bench.log
