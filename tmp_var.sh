cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v WARNING | tail -3
timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | grep -v WARNING | tail -1
timeout 200 python tools/frame_breakdown.py 1 2>&1 | grep -v WARNING | grep "stage" | head -24
