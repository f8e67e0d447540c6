cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
  echo "rc=$?"
  DB=$(find gpurun_out/pmc_$c -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB --counters 2>&1 | grep -E "$c|columns" | head -20
  rm -rf gpurun_out/pmc_$c
done
