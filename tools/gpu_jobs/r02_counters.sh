mkdir -p gpurun_out
timeout 900 python tools/kernel_counters.py --steps 20 --warmup 5 > gpurun_out/r02_kernel_counters.txt 2>&1; cat gpurun_out/r02_kernel_counters.txt
