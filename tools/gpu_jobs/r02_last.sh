mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_tlas.py -x -q 2>&1 | tail -3
timeout 900 python tools/kernel_counters.py --steps 20 --warmup 5 > gpurun_out/r02_kernel_counters.txt 2>&1; cat gpurun_out/r02_kernel_counters.txt
