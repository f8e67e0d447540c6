# round 4, fifth GPU session: where do the traversal waves spend their cycles? (SQ wait / active breakdown, instruction mix), and
# which TA / TCP counters this rocprofv3 knows
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TD_[A-Z0-9_]+|GRBM_[A-Z0-9_]+)\b" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r04_counters_available.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r04_counters_available.txt
cd $GRAFT_REPO_ROOT
timeout 900 python tools/pmc_pass.py --steps 20 --warmup 5 --groups \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" \
  "SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH" \
  "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
  > gpurun_out/r04_pmc_breakdown.json 2> gpurun_out/r04_pmc_breakdown.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_pmc_breakdown.json"))
print("errors:", d["errors"])
for name, c in d["kernels"].items():
    if not any(k in name for k in ("trace_stream", "material", "sort_stream")): continue
    print(name)
    for k, v in sorted(c.items()): print("   %-40s %6d %.6g" % (k, v[0], v[1]))
PY
