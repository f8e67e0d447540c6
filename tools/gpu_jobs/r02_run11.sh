mkdir -p gpurun_out
(time timeout 900 python -m pytest tests/test_gpu_materials_svgf.py tests/test_gpu_full_size.py -x -q -k "svgf" --durations=8) > gpurun_out/r02_run11_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r02_run11_tests.log
CONFIG_ONLY=3 timeout 600 python tools/config_suite.py > gpurun_out/r02_run11_config3.log 2>&1
grep "^config" gpurun_out/r02_run11_config3.log
