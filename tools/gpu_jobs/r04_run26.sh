# round 4, twenty-sixth GPU session: counters of the filter kernels with and without the LDS tiles
mkdir -p gpurun_out
timeout 1200 python tools/svgf_counters.py > gpurun_out/r04_svgf_counters.txt 2> gpurun_out/r04_svgf_counters.err; cat gpurun_out/r04_svgf_counters.txt; tail -3 gpurun_out/r04_svgf_counters.err
