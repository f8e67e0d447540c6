# round 4, fifteenth GPU session: kernel timeline of one rank's burst (8-way tile split emulated on one GPU)
mkdir -p gpurun_out
R=$PWD
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-stages --no-reference-layout"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace8 && timeout 300 rocprofv3 --kernel-trace -d /tmp/trace8 -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --emulate-world 8 $B > gpurun_out_b.json 2>/dev/null
cd $R
DB=$(find /tmp/trace8 -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB 100000 | tail -700 > gpurun_out/r04_run15_timeline_w8.txt 2>&1; wc -l gpurun_out/r04_run15_timeline_w8.txt
cat /tmp/gpurun_out_b.json | cut -c1-300
