# round 4, tenth GPU session: per-kernel counters of the whole step after the shade changes
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 900 python tools/kernel_counters.py --steps 20 --warmup 5 > gpurun_out/r04_run10_kernel_counters.txt 2>gpurun_out/r04_run10_kernel_counters.err; tail -40 gpurun_out/r04_run10_kernel_counters.txt
