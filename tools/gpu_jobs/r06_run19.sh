# round 6, GPU session 19: the drain of the traversal launch wave by wave (probe build: RT_WAVE_CLOCK), whole frame and rank 0 of 8
mkdir -p gpurun_out
export GRT_DEVICE_LIB=$PWD/gpu-raytracer_amd/csrc/_variants/waveclock/libgrt_device.so
timeout 300 python tools/wave_clock_probe.py --world 1 > gpurun_out/r06_wave_clock_world1.txt 2>&1; cat gpurun_out/r06_wave_clock_world1.txt | tail -30
timeout 300 python tools/wave_clock_probe.py --world 8 > gpurun_out/r06_wave_clock_world8.txt 2>&1; cat gpurun_out/r06_wave_clock_world8.txt | tail -30
