# round 6, GPU session 21: the whole GPU suite on the build with the endgame regions + primary-ray patches, then the driver's command
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_run21_pytest.log 2>&1; tail -3 gpurun_out/r06_run21_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_run21_bench.json 2> gpurun_out/r06_run21_bench.err; tail -c 600 gpurun_out/r06_run21_bench.err; head -c 700 gpurun_out/r06_run21_bench.json
