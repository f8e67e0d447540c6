cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v WARNING | tail -3
timeout 600 python bench.py 2>&1 | grep -v WARNING | tail -1 > gpurun_out/bench_r01_default.log
cat gpurun_out/bench_r01_default.log
rm -rf gpurun_out/prof_r01
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r01 -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
DB=$(find gpurun_out/prof_r01 -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r01_final_kernel_trace_stats.txt 2>&1
cat gpurun_out/r01_final_kernel_trace_stats.txt | head -14
grep -v WARNING gpurun_out/prof_bench.log | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
rm -rf gpurun_out/prof_r01
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')
" 2>&1 | grep -v WARNING | tail -3
