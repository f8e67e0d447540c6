# round 6, GPU session 36: what sits next to the 40 us launches of a rank's last iterations (kernels + memory copies of the trace, in time order)
mkdir -p gpurun_out
B="--gpus 1 --steps 20 --warmup 5 --emulate-world 8 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages --no-tile-split-bound"
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/r06_prof36 -o bench -- python $GRAFT_REPO_ROOT/bench.py $B > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r06_prof36 -name "*.db" | head -1)
python -c "
import sqlite3,sys; db=sqlite3.connect('$DB'); print([r[0] for r in db.execute(\"select name from sqlite_master where type in ('table','view')\").fetchall()][:60])"
python tools/rocpd_timeline.py $DB 3 kernel_accumulate_group 0.3 | tail -70
rm -rf gpurun_out/r06_prof36
