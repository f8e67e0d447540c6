# round 5, eighth GPU session: kernel timeline of rank 0's burst under bench.py --emulate-world 8 (where do the exchange's 0.7 ms go?)
mkdir -p gpurun_out
B="--gpus 1 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages"
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/r05_prof8 -o bench -- python $GRAFT_REPO_ROOT/bench.py $B > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r05_prof8 -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB 9 kernel_accumulate_group 1.5 > gpurun_out/r05_rank_timeline.txt 2>&1; wc -l gpurun_out/r05_rank_timeline.txt
rm -rf gpurun_out/r05_prof8
