# round 4, thirtieth GPU session: the box pass of the device build over the whole grid (integer atomics per wave): validity + hits, build times
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blas.py tests/test_gpu_tlas.py -x -q 2>&1 | tail -5 > gpurun_out/r04_run30_pytest.log; tail -2 gpurun_out/r04_run30_pytest.log
for ms in 0 1; do
  echo "--- merge_static $ms"
  MERGE_STATIC=$ms timeout 600 python tools/blas_bench.py 2>/dev/null | tail -2
done
