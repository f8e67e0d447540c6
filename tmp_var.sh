cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
echo "overlap shadows on"; BATCH=4 IN_FLIGHT=1,2,4,8 timeout 300 python tools/rank_emulation.py 8 2>&1 | grep -v WARNING
echo "overlap shadows off"; GRT_OVERLAP_SHADOWS=0 BATCH=4 IN_FLIGHT=1,2,4,8 timeout 300 python tools/rank_emulation.py 8 2>&1 | grep -v WARNING
