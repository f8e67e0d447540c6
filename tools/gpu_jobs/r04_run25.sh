# round 4, twenty-fifth GPU session: the whole GPU suite on the round's last build (device BLAS rule, the command line's burst), configs 3-5 at full size,
# counters of the filter kernels with and without the LDS tiles
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_run25_pytest.log; echo "suite: $(tail -1 gpurun_out/r04_run25_pytest.log)"
timeout 900 python tools/config_suite.py > gpurun_out/r04_config_suite.txt 2> gpurun_out/r04_config_suite.err; cat gpurun_out/r04_config_suite.txt
timeout 900 python tools/svgf_counters.py > gpurun_out/r04_svgf_counters.txt 2> gpurun_out/r04_svgf_counters.err; cat gpurun_out/r04_svgf_counters.txt
