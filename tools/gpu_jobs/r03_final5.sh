# after the instancing rule: whole GPU suite, config suite
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -60 > gpurun_out/r03_gpu_suite.log; tail -9 gpurun_out/r03_gpu_suite.log

