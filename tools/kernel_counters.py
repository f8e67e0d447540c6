#!/usr/bin/env python3
"""Per-kernel hardware counters of the whole benchmark step (not only the traversal launch that bench.py prices): runs the
three counter passes of tools/pmc_pass.py and prints, per kernel of the wavefront, dispatches, time, memory-side bytes per
dispatch (FETCH_SIZE / WRITE_SIZE with the calibration factors bench.py measured in profiles/r02_bench.json), the rate
those bytes move at, VALU lane utilisation, the share of a wave's cycles spent issuing VALU, resident waves per SIMD and
their product (how busy the SIMD's vector ALU is). usage (GPU box): python tools/kernel_counters.py [--steps 20 --warmup 5]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_pass  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    fetch_cal, write_cal = 2.0, 0.568
    try:
        detail = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))["roofline"]["traffic_detail"]
        fetch_cal, write_cal = detail["fetch_size_calibration"], detail["write_size_calibration"]
    except Exception:
        pass
    result = pmc_pass.run_passes(a.steps, a.warmup)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(result, open(os.path.join(ROOT, "gpurun_out", "kernel_counters_raw.json"), "w"))
    if result["errors"]:
        print("errors:", result["errors"])
    print("calibration: FETCH_SIZE x %.3f, WRITE_SIZE x %.3f (KB units)" % (fetch_cal, write_cal))
    print("%-34s %6s %10s %10s %10s %9s %8s %8s %8s %8s" % ("kernel", "calls", "total ms", "rd MB/call", "wr MB/call", "GB/s", "lanes", "issue", "waves", "valu"))
    rows = []
    for name, c in result["kernels"].items():
        if "_duration_ns" not in c:
            continue
        calls, ns = c["_duration_ns"]
        rd = c.get("FETCH_SIZE", [0, 0.0])[1] * 1024.0 * fetch_cal
        wr = c.get("WRITE_SIZE", [0, 0.0])[1] * 1024.0 * write_cal
        insts, threads = c.get("SQ_INSTS_VALU", [0, 0.0])[1], c.get("SQ_THREAD_CYCLES_VALU", [0, 0.0])[1]
        wave_cycles, active = c.get("SQ_WAVE_CYCLES", [0, 0.0])[1], c.get("SQ_ACTIVE_INST_VALU", [0, 0.0])[1]
        lanes = threads / (64.0 * insts) if insts else 0.0
        issue = active / wave_cycles if wave_cycles else 0.0
        waves = 4.0 * wave_cycles / (1024.0 * ns * 2.4) if ns else 0.0
        rows.append((ns, name, calls, ns / 1e6, rd / calls / 1e6, wr / calls / 1e6, (rd + wr) / ns if ns else 0.0, lanes, issue, waves, issue * waves))
    for r in sorted(rows, reverse=True):
        print("%-34s %6d %10.3f %10.2f %10.2f %9.0f %8.3f %8.3f %8.2f %8.2f" % ((r[1][-34:],) + tuple(r[2:])))


if __name__ == "__main__":
    main()
