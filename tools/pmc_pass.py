#!/usr/bin/env python3
"""Hardware counters of the traversal kernel inside the benchmark: runs `bench.py` under `rocprofv3 --kernel-trace --pmc
<counters>` (ONE counter group per pass, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass)
and sums the counters per kernel from the rocpd database. bench.py calls run_passes() on rank 0; stand-alone:

    python tools/pmc_pass.py --steps 20 --warmup 5 [--groups FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES ..."]

HBM bytes: FETCH_SIZE / WRITE_SIZE are the L2's memory-side request counters in KB; on gfx950 a 128-B read request is
tallied as 64 B (guide, "HBM"), other widths are uncalibrated -- so both are calibrated IN THE SAME RUN on kernels whose
bytes are known: kernel_stream_read (1 GiB read by the stream-bandwidth probe of bench.py) and kernel_generate_stream
(28 B written per primary ray)."""
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (8 SQ slots + 1 GRBM per pass: MI355X_MICROARCH.md "rocprofv3 PMC slots")
DEFAULT_GROUPS = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"]


def summarise(db_path):
    """{kernel name: {counter: [dispatches, sum]}} from the rocpd database of one pass."""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    out = {}
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
    for kernel, counter, n, total in rows:
        out.setdefault(kernel.split("(")[0], {})[counter] = [int(n), float(total)]
    try:   # the time the dispatches of a kernel took in this pass (for rates per cycle), from the kernel trace of the same run
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
        for kernel, n, ns in cur.execute("select %s, count(*), sum(end - start) from kernels group by %s" % (name_col, name_col)).fetchall():
            if kernel.split("(")[0] in out:
                out[kernel.split("(")[0]]["_duration_ns"] = [int(n), float(ns)]
    except Exception:
        pass
    return out


def run_pass(counters, bench_args, timeout_s=240):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="grt_pmc_", dir="/tmp")
    cmd = [rocprof, "--kernel-trace", "--pmc"] + counters.split() + ["-d", work, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp", BENCH_PMC_CHILD="1")
    try:
        proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s, start_new_session=True)
    except subprocess.TimeoutExpired:
        shutil.rmtree(work, ignore_errors=True)
        return None, "timed out after %d s" % timeout_s
    dbs = glob.glob(os.path.join(work, "**", "*.db"), recursive=True)
    if proc.returncode != 0 or not dbs:
        shutil.rmtree(work, ignore_errors=True)
        return None, "rocprofv3 exited with %d: %s" % (proc.returncode, proc.stdout[-300:])
    try:
        result = summarise(dbs[0])
    except Exception as e:  # the schema differs between rocprofv3 versions
        result, proc.stdout = None, "cannot read the counter database: %s" % e
    shutil.rmtree(work, ignore_errors=True)
    return result, (None if result is not None else proc.stdout)


def run_passes(steps, warmup, groups=None, extra_args=()):
    """Returns {"kernels": {kernel: {counter: [dispatches, sum]}}, "errors": [...]} over all passes."""
    merged, errors = {}, []
    bench_args = ["--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-povs", "--no-pmc", "--no-stages", "--no-config3", "--no-reference-layout"] + list(extra_args)
    for group in (groups or DEFAULT_GROUPS):
        result, error = run_pass(group, bench_args)
        if result is None:
            errors.append("%s: %s" % (group, error))
            continue
        for kernel, counters in result.items():
            merged.setdefault(kernel, {}).update(counters)
    return {"kernels": merged, "errors": errors}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--groups", nargs="*", default=None)
    a = ap.parse_args()
    print(json.dumps(run_passes(a.steps, a.warmup, a.groups), indent=1))
