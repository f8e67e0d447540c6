# round 4, twenty-fourth GPU session: the device BLAS build cutting the piece with the largest surface area first (box table of power-of-two runs) against
# round 3's widest-piece-first: validity + hits (tests/test_gpu_blas.py), then build time and traversal cost on the benchmark scene in both layouts
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_blas.py tests/test_gpu_widening.py::test_command_line_render_and_screenshots_match_the_library -x -q 2>&1 | tail -6 > gpurun_out/r04_run24_pytest.log; tail -3 gpurun_out/r04_run24_pytest.log
for ms in 0 1; do
  echo "--- merge_static $ms, largest area first"
  MERGE_STATIC=$ms timeout 600 python tools/blas_bench.py 2>/dev/null | tail -3
  echo "--- merge_static $ms, widest piece first (round 3)"
  MERGE_STATIC=$ms GRT_BLAS_SPLIT_WIDEST=1 timeout 600 python tools/blas_bench.py 2>/dev/null | tail -2
done
