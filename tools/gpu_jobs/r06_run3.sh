# round 6, GPU session 3: the L1 roof (tools/microbench/l1_lookup_rate.hip: plain, then under rocprofv3 --pmc for the TCP's own counts), then the new GPU test
# of the skipping walk / the seating, then the trace tests on the shipped build (pop as a loop).
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/microbench/l1_lookup_rate
$B > $GRAFT_REPO_ROOT/gpurun_out/r06_l1_lookup_rate_plain.txt 2>&1; cat $GRAFT_REPO_ROOT/gpurun_out/r06_l1_lookup_rate_plain.txt
for group in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum" "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  rm -rf /tmp/l1prof; timeout 300 rocprofv3 --kernel-trace --pmc $group -d /tmp/l1prof -o l1 -- $B 2048 > /tmp/l1prof.log 2>&1 || tail -5 /tmp/l1prof.log
  python3 - "$group" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob("/tmp/l1prof/**/*.db", recursive=True)
if not dbs: print("no database for", sys.argv[1]); sys.exit(0)
db = sqlite3.connect(dbs[0]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]; name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
dur = {k.split("(")[0]: (n, ns) for k, n, ns in cur.execute("select %s, count(*), sum(end - start) from kernels group by %s" % (name_col, name_col))}
rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
for kernel, counter, n, total in sorted(rows):
    k = kernel.split("(")[0]; d = dur.get(k, (0, 0))
    print("%-44s %-34s dispatches %d sum %.6g  | kernel ns %.6g (%d dispatches)" % (k[:44], counter, n, total, d[1], d[0]))
PY
done > $GRAFT_REPO_ROOT/gpurun_out/r06_l1_lookup_rate_pmc.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r06_l1_lookup_rate_pmc.txt | head -80
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py -x -q -k "seating or bit_exact or flattened or statistics" 2>&1 | tail -5
