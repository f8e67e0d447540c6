# round 6, GPU session 8: waves resident per SIMD of the traversal launch, dispatch by dispatch (tools/occupancy_probe.py)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 python $GRAFT_REPO_ROOT/tools/occupancy_probe.py 2>&1 | grep -v WARNING | tee $GRAFT_REPO_ROOT/gpurun_out/r06_occupancy_raw.txt | tail -70
