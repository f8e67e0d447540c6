# round 6, GPU session 35 (the round's last state): kernel timelines of the driver's command on the whole frame and on rank 0's share of an 8-way split (where does a rank's traversal time go?)
mkdir -p gpurun_out
for W in 0 8; do
B="--gpus 1 --steps 20 --warmup 5 --emulate-world $W --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages --no-tile-split-bound"
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/r06_prof35_$W -o bench -- python $GRAFT_REPO_ROOT/bench.py $B > $GRAFT_REPO_ROOT/gpurun_out/r06_run35_$W.json 2>/dev/null; cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r06_prof35_$W -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB 40 kernel_accumulate_group 1.5 > gpurun_out/r06_timeline_final_world$W.txt 2>&1; wc -l gpurun_out/r06_timeline_final_world$W.txt
rm -rf gpurun_out/r06_prof35_$W
done
