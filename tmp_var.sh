cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "edge_sizes or batches" 2>&1 | grep -v WARNING | tail -8
timeout 200 python tools/svgf_timing.py 2>&1 | grep -v WARNING | tail -3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  DB=$(find gpurun_out/pmc_$c -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB --counters 2>&1 | grep -E "$c" | head -20
  rm -rf gpurun_out/pmc_$c
done
