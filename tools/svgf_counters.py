#!/usr/bin/env python3
"""Hardware counters of the SVGF filter kernels inside the benchmark's config-3 section (one rocprofv3 --pmc pass per counter group, as
tools/pmc_pass.py does for the traversal): what the a-trous passes ask of L1 and of the L2 -> L1 path, with and without the LDS tiles
(rt_set_svgf_tiles). usage (GPU box): python tools/svgf_counters.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_pass  # noqa: E402

ARGS = ["--gpus", "1", "--steps", "4", "--warmup", "4", "--no-cpu-baseline", "--no-povs", "--no-pmc", "--no-stages", "--no-reference-layout"]
GROUPS = ["TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum", "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE", "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"]
PIXELS = 1920 * 1080


def main():
    for tiles in (1, 0):
        os.environ["BENCH_SVGF_TILES"] = str(tiles); os.environ["BENCH_PMC_CONFIG3"] = "1"   # (a counter pass of bench.py skips config 3 unless told otherwise)
        merged = {}
        for group in GROUPS:
            result, error = pmc_pass.run_pass(group, ARGS)
            if result is None:
                print("tiles %d, %s: %s" % (tiles, group, error)); continue
            for kernel, counters in result.items():
                merged.setdefault(kernel, {}).update(counters)
        print("svgf_lds_tiles = %d" % tiles)
        print("%-46s %7s %9s %12s %14s %12s %8s %8s %8s" % ("kernel", "calls", "us / call", "L1 acc / px", "L1->L2 rd / px", "valu / px", "lds / px", "vmem / px", "waves"))
        for kernel in sorted(merged):
            if "svgf" not in kernel and "taa" not in kernel:
                continue
            c = merged[kernel]; calls = c.get("_duration_ns", [0, 0.0])[0] or 1
            per = lambda name: c.get(name, [0, 0.0])[1] / calls / PIXELS
            us = c.get("_duration_ns", [0, 0.0])[1] / calls / 1e3
            waves = 4.0 * c.get("SQ_WAVE_CYCLES", [0, 0.0])[1] / (1024.0 * max(c.get("_duration_ns", [0, 1.0])[1], 1.0) * 2.4)
            print("%-46s %7d %9.1f %12.2f %14.2f %12.1f %8.2f %8.2f %8.2f" % (kernel[-46:], calls, us, per("TCP_TOTAL_CACHE_ACCESSES_sum"), per("TCP_TCC_READ_REQ_sum"), 64.0 * per("SQ_INSTS_VALU"), 64.0 * per("SQ_INSTS_LDS"), 64.0 * (per("SQ_INSTS_VMEM_RD") + per("SQ_INSTS_VMEM_WR")), waves))


if __name__ == "__main__":
    main()
