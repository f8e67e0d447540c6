mkdir -p gpurun_out
for s in cornellbox sponza; do timeout 300 python tools/blas_diagnose.py $s 2>&1 | grep -v "WARNING\|Runtime\|h = np\|q = np\|ok = " | head -8; done
timeout 600 python -m pytest tests/test_gpu_blas.py -q 2>&1 | tail -30 > gpurun_out/r03_run8_blas.log; grep -n "^E  \|passed\|failed" gpurun_out/r03_run8_blas.log | head -12
timeout 300 python tools/blas_bench.py 2>&1 | grep -v WARNING | tee gpurun_out/r03_run8_blas_bench.log
