#!/usr/bin/env python3
"""How many waves the traversal launch keeps resident, LAUNCH BY LAUNCH: bench.py's roofline record says 4.9-5.1 waves per SIMD
on average against the 6 the kernel is compiled for. Runs the driver's command under `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES
GRBM_GUI_ACTIVE` (one pass) and prints, per dispatch of the traversal kernel in the timed plan's order: duration, waves launched, waves resident per SIMD
(4 x SQ_WAVE_CYCLES / (1024 SIMDs x cycles): SQ_WAVE_CYCLES counts quad-cycles).   usage (GPU box): python tools/occupancy_probe.py [--steps 20 --warmup 5]"""
import argparse
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    work = tempfile.mkdtemp(prefix="grt_occ_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "-d", work, "-o", "occ", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup), "--no-cpu-baseline", "--no-povs", "--no-pmc", "--no-stages", "--no-config3", "--no-reference-layout"]
    proc = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", BENCH_PMC_CHILD="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    dbs = glob.glob(os.path.join(work, "**", "*.db"), recursive=True)
    if proc.returncode != 0 or not dbs:
        print("rocprofv3 failed:", proc.stdout[-1500:]); return
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    columns = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("counters_collection columns:", columns)
    key = "dispatch_id" if "dispatch_id" in columns else ("correlation_id" if "correlation_id" in columns else None)
    time_cols = [c for c in ("start", "end") if c in columns]
    if key is None:
        print("no per-dispatch key in this rocprofv3's schema"); return
    select = "select %s, kernel_name, counter_name, sum(value)%s from counters_collection where kernel_name like '%%trace_stream%%' group by %s, counter_name order by %s" % (
        key, (", min(start), max(end)" if len(time_cols) == 2 else ""), key, key)
    rows = cur.execute(select).fetchall()
    launches = {}
    for row in rows:
        d = launches.setdefault(row[0], {"kernel": row[1].split("(")[0]})
        d[row[2]] = float(row[3])
        if len(row) > 4:
            d["ns"] = float(row[5] - row[4])
    if launches and "ns" not in next(iter(launches.values())):   # durations from the kernel trace, by order of dispatch
        kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in kcols else [c for c in kcols if "name" in c][0]
        traced = cur.execute("select start, end from kernels where %s like '%%trace_stream%%' order by start" % name_col).fetchall()
        for (k, d), (s, e) in zip(sorted(launches.items()), traced):
            d["ns"] = float(e - s)
    print("%-8s %-40s %10s %10s %12s %14s" % ("launch", "kernel", "ms", "waves", "waves/SIMD", "busy fraction"))
    total_cycles = total_wave = 0.0
    for k, d in sorted(launches.items()):
        cycles = d.get("ns", 0.0) * 2.4
        resident = 4.0 * d.get("SQ_WAVE_CYCLES", 0.0) / (1024.0 * cycles) if cycles else float("nan")
        busy = d.get("SQ_BUSY_CYCLES", 0.0) / max(d.get("GRBM_GUI_ACTIVE", 1.0), 1.0)
        total_cycles += cycles; total_wave += 4.0 * d.get("SQ_WAVE_CYCLES", 0.0)
        print("%-8s %-40s %10.4f %10d %12.2f %14.3f" % (k, d["kernel"][:40], d.get("ns", 0.0) * 1e-6, int(d.get("SQ_WAVES", 0)), resident, busy))
    if total_cycles:
        print("all traversal launches: %.2f waves resident per SIMD (time-weighted)" % (total_wave / (1024.0 * total_cycles)))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
