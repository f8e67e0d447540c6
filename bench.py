#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) and ms/frame on Sponza 1920x1080, BVH8,
diffuse + plastic materials, NEE + MIS + Russian roulette, 10 bounces, samples 0..3 (4 spp).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the whole frame = one sample per pixel:
generate -> (trace, sort, shade, shadow-trace) x bounces -> accumulate (Pathtracer::render of the
reference, Pathtracer.cpp:738-855).  The steps are submitted the way the library is meant to be
driven: the 4 samples of a frame as ONE wavefront (rt_render_samples: every launch carries 4 paths
per pixel; bit-identical to 4 single-sample calls) and up to --samples-in-flight such submissions
concurrently; --steps K renders exactly K samples.  With N > 1 the frame is split into row tiles
dealt round-robin to the ranks (each rank holds a full scene replica) and one RCCL all-gather per
completed 4-spp frame rebuilds the float4 frame on every rank: the total work is fixed, so the
scaling is STRONG.  Scene data is resident in HBM before the timed region; nothing crosses
PCIe inside it.

Rank 0 prints ONE JSON line.  `value` counts closest-hit rays (bounce 0 = primary, bounces >= 1 =
secondary) of all ranks per second of max-over-ranks wall time; shadow rays are reported
separately in `config`.  `roofline` prices the dominant kernel (kernel_trace_bvh8) in ALGORITHMIC
bytes (40 B per ray + 80 B per BVH8 node fetched + 48 B per triangle tested + 52 B
per transformed instance entry + 4 B per identity entry, the node / triangle counts measured by the
counting variant of the same kernel on the same rays) over its HIP-event time, against the 8 TB/s
HBM3E peak of MI355X.  `cpu_baseline` times the CPU oracle (a port of the reference's device
traversal, OpenMP over rays) on a bounded sample of the same rays on this box's host cores, and
the reference's own BVH builder (oracle/_ref, compiled verbatim) where it has been built.
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto 4 hardware queues by default; the tracer keeps (samples in flight) x 2 streams busy
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
NUM_BOUNCES = 10
SPP = 4
# The fixed Sponza points of view of the reference's perf harness (Util/PerfTest.h:30-40: position, rotation quaternion);
# its harness times 32 frames at each and reports the average per POV.
SPONZA_POVS = [
    ((18.739738, 10.332139, -10.229103), (0.000000, 0.801883, 0.000000, 0.597480)),
    ((31.355043, 31.696985, 13.222142), (0.000000, 0.387925, 0.000000, -0.921690)),
    ((70.257584, 8.347624, 49.902672), (0.000000, -0.576111, 0.000000, -0.817371)),
    ((24.349691, 51.417969, -10.351927), (0.000000, -0.985181, 0.000000, 0.171514)),
    ((24.349691, 51.417969, -10.351927), (0.000000, -0.245309, 0.000000, -0.969444)),
    ((-15.957721, 62.806641, -43.916168), (0.000000, -0.803925, 0.000000, 0.594729)),
    ((-52.839905, 38.513454, -8.991060), (0.202261, -0.729369, -0.606600, -0.243197)),
    ((-92.179306, 74.721153, 12.197323), (0.009840, 0.621556, 0.007809, -0.783262)),
    ((-129.707321, 17.916590, 43.054050), (0.011467, 0.408287, 0.005129, -0.912762)),
]
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, "Chip-level parameters"); ~6.3-6.6 TB/s is achievable
# ---- the peaks the traversal launch is priced against (roofline.binding); every one from /opt/skills/guides/MI355X_MICROARCH.md ----------
CLOCK_HZ = 2.4e9          # max clock
CUS, SIMDS_PER_CU = 256, 4
# Vector ALUs: four SIMD-32 per CU, a 64-lane wave-instruction issues over 2 cycles ("Wave scheduling"; "Per-instruction cycle constants": v_fma_f32
# 2 cyc) = 32 lane-instructions per SIMD and cycle. 256 x 4 x 32 x 2.4e9 = 7.86e13 lane-instructions per second -- the data sheet's 157.3 TFLOP/s FP32
# counts each fused multiply-add as two. (Rounds 3-4 priced against half of this, 16 lanes per cycle: wrong, the review of round 4 said so.)
VALU_PEAK_LANE_INSTR_PER_S = CUS * SIMDS_PER_CU * 32 * CLOCK_HZ
# What the kernel's own instruction mix could reach at best (profiles/r04_instruction_costs.txt, tools/microbench/valu_rates.hip, measured on this chip):
# fma / mul / add / sub / logic / v_bitop3 / v_mov / right shifts retire a wave in ~2.3 cycles ("fast"), conversions / min / max / compares / selects / left
# shifts / bit-field instructions in ~4.1 ("slow"), and one of each interleaved take 4.5 cycles per PAIR (they overlap).
VALU_CYCLES_FAST, VALU_CYCLES_SLOW, VALU_CYCLES_PAIR = 2.3, 4.1, 4.52
# Static mix of one round of the shipped closest-hit engine (kernel_trace_stream_bvh8_flat_skip, profiles/trace_round_mix.json, tools/isa_loop_mix.py): vector instructions, of which slow
TRACE_ROUND_VALU, TRACE_ROUND_VALU_SLOW = 397, 183
try:   # tools/isa_loop_mix.py --json profiles/trace_round_mix.json, re-run whenever the traversal kernel changes
    _mix = json.load(open(os.path.join(ROOT, "profiles", "trace_round_mix.json")))
    TRACE_ROUND_VALU, TRACE_ROUND_VALU_SLOW = int(_mix["valu"]), int(_mix["valu_slow"])
except Exception:
    pass
# L1 (vector cache, one per CU). The guide gives its capacity and no throughput: the roof is MEASURED (tools/microbench/l1_lookup_rate.hip under rocprofv3 --pmc,
# profiles/r06_l1_lookup_rate.txt): TCP_TOTAL_CACHE_ACCESSES per clock and CU with every access hitting, by access shape -- a lane's 16-byte piece of its own
# 80-byte record (the node fetch) 1.672, of its own 48-byte record (the triangle fetch) 1.978, 64 different 128-byte lines 0.989, a coalesced stream 0.913
# (58.5 B per clock). The traversal launch is priced against the rate of ITS mix of the first two (l1_peak_lookups_per_clock below).
L1_LOOKUPS_PER_CLOCK = {"node": 1.672, "triangle": 1.978, "lines": 0.989, "coalesced": 0.913}
L1_STREAM_GBPS = CUS * 64 * L1_LOOKUPS_PER_CLOCK["coalesced"] * CLOCK_HZ / 1e9    # what the L1's data path streams, whole chip: 35.9 TB/s
L2_PEAK_GBPS = 34500.0   # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate


def l1_peak_lookups_per_clock(node_visits, triangle_tests):
    """Look-ups per clock and CU the L1 sustains on this launch's mix of node fetches (5 per visit and lane) and triangle fetches (3 per test)."""
    node, tri = 5.0 * node_visits, 3.0 * triangle_tests
    if node + tri <= 0:
        return L1_LOOKUPS_PER_CLOCK["node"]
    return (node + tri) / (node / L1_LOOKUPS_PER_CLOCK["node"] + tri / L1_LOOKUPS_PER_CLOCK["triangle"])
L1_TO_L2_REQUEST_BYTES = 64   # TCP_TCC_READ_REQ counts 64-byte requests (a 128-byte line miss is two)


TRACE_KERNEL = ["kernel_trace_stream_bvh8"]   # the dominant kernel's name in the rocprofv3 records: ..._flat when the whole scene is one flattened tree (rt_set_static_geometry)
EXPAND_TEXTURES = 1   # --expand-textures: 0 keeps BC1 blocks compressed on the device, decoded per texel fetch (config expand_block_compressed_textures)
MERGE_STATIC = 1   # --merge-static: 0 stages the scene exactly as the reference does (one BLAS per mesh under the TLAS)


def build_scene(grt):
    """BASELINE config #2: Sponza, every odd diffuse material -> roughplastic alpha 0.3."""
    grt.config_reset()
    grt.config_set(merge_static=MERGE_STATIC, expand_block_compressed_textures=EXPAND_TEXTURES)
    if os.environ.get("BENCH_SLOT_ASSIGNMENT"):   # experiments: how the flattened tree's collapse deals children to octant slots (config static_slot_assignment)
        grt.config_set(static_slot_assignment=int(os.environ["BENCH_SLOT_ASSIGNMENT"]))
    if os.environ.get("BENCH_SLOT_LEARNING_RAYS"):
        grt.config_set(static_slot_learning_rays=int(os.environ["BENCH_SLOT_LEARNING_RAYS"]))
    if os.environ.get("BENCH_SLOT_LEARNING_VIEWPOINT"):
        grt.config_set(static_slot_learning_viewpoint=int(os.environ["BENCH_SLOT_LEARNING_VIEWPOINT"]))
    if os.environ.get("BENCH_RESEAT_DISTANCE"):   # experiments: 0 = the flattened tree is never seated again when the camera travels (config static_reseat_distance)
        grt.config_set(static_reseat_distance=float(os.environ["BENCH_RESEAT_DISTANCE"]))
    if os.environ.get("BENCH_SKIP_BEHIND_HIT"):   # A / B: 0 = the reference's walk node for node (config skip_behind_hit, rt_set_skip_behind_hit)
        grt.config_set(skip_behind_hit=int(os.environ["BENCH_SKIP_BEHIND_HIT"]))
    # the reference's own 19 diffuse maps when build() could install them (assets/_cache, see install_reference_sponza_textures),
    # else the quarter-size maps that travel inside the repository, every texel replicated 4x4
    scene = grt.Scene(grt.scene_path("sponza_reference_maps" if grt.reference_sponza_textures_installed() else "sponza"))
    for i in range(1, scene.material_count, 2):
        if scene.material_type(i) == grt.MATERIAL_DIFFUSE:
            scene.set_material(i, grt.MATERIAL_PLASTIC, None, 0.3)
    grt.config_set(num_bounces=NUM_BOUNCES)
    return scene


def cpu_baseline(grt, pt, scene):
    """CPU leg (rank 0, N = 1 only): bounded sample, ~10-30 s of host work."""
    from oracle import binding as oracle  # checker only: never on the product path
    view = oracle.SceneView(pt)
    # threads: all logical CPUs up to 32 -- the container of a GPU box sees 256 logical CPUs but gets about 16 cores' worth
    # (effective_parallelism below); the traversal loop peaks at 16-32 threads there (8 Mrays/s) and falls to 3.5 Mrays/s at 256
    threads = min(os.cpu_count() or 1, 32)
    n = grt.RT_BATCH_SIZE
    o, d, _ = view.generate(0, 0, n)
    t0 = time.perf_counter()
    hits, stats_primary = view.trace(o, d, threads)
    t_primary = time.perf_counter() - t0
    # diffuse-bounce stand-in rays leaving the primary hit points (seeded)
    rng = np.random.default_rng(1234)
    t = hits[:, 2].view(np.float32)
    ok = hits[:, 1] != 0xffffffff
    org = (o + d * np.where(ok, t, 1.0).astype(np.float32) * np.float32(0.999))[:, ok]
    dirs = rng.normal(size=org.shape).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=0)
    t0 = time.perf_counter()
    _, stats_secondary = view.trace(org, dirs, threads)
    t_secondary = time.perf_counter() - t0
    rays = n + org.shape[1]
    out = {
        "value": round(rays / (t_primary + t_secondary) / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "oracle bvh8_trace (OpenMP, %d threads) over the %d primary rays of batch 0 / sample 0 plus %d seeded diffuse-bounce rays from their hit points" % (threads, n, org.shape[1]),
        "primary_mrays_s": round(n / t_primary / 1e6, 3), "secondary_mrays_s": round(org.shape[1] / t_secondary / 1e6, 3),
    }
    # same rays on the GPU through the C ABI: like-for-like ratio + algorithmic GB/s per ray class
    _, ms_p = grt.trace_rays(pt.ctx, o, d, repeat=5)
    _, ms_s = grt.trace_rays(pt.ctx, org, dirs, repeat=5)
    out["gpu_same_rays"] = {
        "primary_mrays_s": round(n / ms_p / 1e3, 1), "secondary_mrays_s": round(org.shape[1] / ms_s / 1e3, 1),
        "primary_alg_gbps": round(stats_primary.algorithmic_bytes() / (ms_p * 1e-3) / 1e9, 1),
        "secondary_alg_gbps": round(stats_secondary.algorithmic_bytes() / (ms_s * 1e-3) / 1e9, 1),
        "nodes_per_ray": [round(stats_primary.nodes / stats_primary.rays, 2), round(stats_secondary.nodes / stats_secondary.rays, 2)],
        "triangles_per_ray": [round(stats_primary.triangles / stats_primary.rays, 2), round(stats_secondary.triangles / stats_secondary.rays, 2)],
    }
    logical = os.cpu_count() or 1
    out["logical_cpus"] = logical
    out["effective_parallelism"] = round(oracle.effective_parallelism(logical), 1)   # a spin loop on every logical CPU against one
    out["note"] = "a port of the reference's device traversal run ray by ray under OpenMP: a stated baseline, not a tuned CPU tracer" 
    if oracle.ref_lib() is not None:  # the reference's own CPU path: BVH2 + BVH8 build of all 383 Sponza meshes
        scene.wait_until_loaded()
        meshes = [scene.mesh_data_array(m, "triangles", np.float32).reshape(-1, 24) for m in range(scene.mesh_data_count)]
        build = {"kind": "reference", "meshes": len(meshes), "triangles": int(sum(m.shape[0] for m in meshes)),
                 "schedule": "one job per mesh on a pool of hardware_concurrency workers, as AssetManager.cpp:57 does",
                 "product_builder_ms_parallel": round(scene.bvh_build_ms, 1)}
        many = oracle.ref_build_many(meshes, logical)   # hardware_concurrency() workers, as the reference's thread pool
        if many is not None:
            one = oracle.ref_build_many(meshes, 1)
            build.update({"cores": logical, "ms_wall": round(many[0], 1), "ms_wall_1_thread": round(one[0], 1), "bvh2_nodes": many[1], "bvh8_nodes": many[2]})
        else:   # an oracle/_ref built before the pooled entry point existed: mesh after mesh on one core
            ms2 = ms8 = 0.0
            for m in meshes:
                ref = oracle.ref_build(m)
                ms2 += ref["ms_bvh2"]; ms8 += ref["ms_bvh8"]
            build.update({"cores": 1, "ms_sah_bvh2": round(ms2, 1), "ms_bvh8_convert": round(ms8, 1), "ms_wall": round(ms2 + ms8, 1)})
        out["reference_bvh_build"] = build
    return out


def pmc_section(args, rays_per_step, launch_ms, plan, node_visits=0.0, triangle_tests=0.0):
    """roofline.traffic and the counters the scope table asks for next to the fraction, from rocprofv3 --pmc
    passes over this very command (tools/pmc_pass.py). Everything is per traversal launch of the timed region, like `achieved`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_pass
    passes = pmc_pass.run_passes(args.steps, args.warmup, groups=pmc_pass.DEFAULT_GROUPS + ["TCC_HIT_sum TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum"],
                                  extra_args=("--merge-static", str(args.merge_static)))
    kernels = passes["kernels"]
    out = {"pmc_errors": passes["errors"]} if passes["errors"] else {}
    trace = kernels.get(TRACE_KERNEL[0])
    if not trace:
        return dict(out, traffic=None)
    # what the child rendered with the non-counting traversal kernel: warm-up, the two profiled frames, the timed plan
    steps_rendered = max(args.warmup, SPP) + 2 * SPP + args.steps
    launches_timed = max(len(launch_ms), 1)
    scale = args.steps / steps_rendered / launches_timed     # sums over the child -> per launch of the timed region
    read_probe, generate = kernels.get("kernel_stream_read"), kernels.get("kernel_generate_stream")
    fetch_factor = write_factor = None
    if "FETCH_SIZE" in trace and read_probe and read_probe.get("FETCH_SIZE", [0, 0])[1] > 0:
        fetch_factor = (read_probe["FETCH_SIZE"][0] * float(1 << 30)) / (read_probe["FETCH_SIZE"][1] * 1024.0)   # known bytes / counted bytes
    if "WRITE_SIZE" in trace and generate and generate.get("WRITE_SIZE", [0, 0])[1] > 0:
        primary_rays = WIDTH * HEIGHT * steps_rendered
        write_factor = (28.0 * primary_rays) / (generate["WRITE_SIZE"][1] * 1024.0)
    if fetch_factor and write_factor:
        read_bytes = trace["FETCH_SIZE"][1] * 1024.0 * fetch_factor * scale
        written_bytes = trace["WRITE_SIZE"][1] * 1024.0 * write_factor * scale
        out["traffic"] = round(read_bytes + written_bytes)
        out["traffic_detail"] = {"hbm_read_bytes_per_launch": round(read_bytes), "hbm_written_bytes_per_launch": round(written_bytes),
                                 "fetch_size_calibration": round(fetch_factor, 3), "write_size_calibration": round(write_factor, 3),
                                 "hbm_gbps_during_traversal": round((read_bytes + written_bytes) / (float(np.mean(launch_ms)) * 1e-3) / 1e9, 1) if len(launch_ms) else None,
                                 "note": "L2 memory-side requests (Infinity-Cache hits included) of all traversal launches of a rocprofv3 --pmc re-run of this command, scaled to one launch of the timed region; FETCH_SIZE / WRITE_SIZE calibrated in that run on kernel_stream_read (1 GiB) and kernel_generate_stream (28 B per primary ray)"}
    else:
        out["traffic"] = None
    if "SQ_INSTS_VALU" in trace and "SQ_THREAD_CYCLES_VALU" in trace and trace["SQ_INSTS_VALU"][1] > 0:
        c = {k: v[1] for k, v in trace.items()}
        counters = {"valu_lane_utilisation": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]), 3)}
        if c.get("SQ_WAVE_CYCLES"):
            counters["wave_cycles_waiting"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
            counters["wave_cycles_issuing_valu"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"], 3)
        seconds = c.get("_duration_ns", 0.0) * 1e-9      # the traversal launches of the counter pass (their own kernel-trace durations)
        if seconds > 0:
            simd_cycles = CUS * SIMDS_PER_CU * seconds * CLOCK_HZ
            counters["valu_instructions_per_cycle_and_simd"] = round(c["SQ_INSTS_VALU"] / simd_cycles, 4)    # peak 0.5: one 64-lane instruction per 2 cycles (SIMD-32)
            counters["cycles_per_valu_instruction_and_simd"] = round(simd_cycles / c["SQ_INSTS_VALU"], 3)
            counters["waves_resident_per_simd"] = round(4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles, 2)   # SQ_WAVE_CYCLES counts quad-cycles
        counters["valu_thread_instructions_per_ray"] = round(c["SQ_THREAD_CYCLES_VALU"] * scale * launches_timed / args.steps / max(rays_per_step, 1.0), 1)
        out["counters"] = counters
        if seconds > 0:
            # ---- roofline.binding: every unit the launch leans on, each against ITS peak from the guide, every fraction <= 1 -----------------
            lane_util = counters["valu_lane_utilisation"]
            useful = c["SQ_THREAD_CYCLES_VALU"] / seconds                     # active lanes summed over all vector instructions, per second
            issue = c["SQ_INSTS_VALU"] / simd_cycles                          # wave-instructions per cycle and SIMD
            fast, slow = TRACE_ROUND_VALU - TRACE_ROUND_VALU_SLOW, TRACE_ROUND_VALU_SLOW
            best_overlapped = (min(fast, slow) * VALU_CYCLES_PAIR + (fast - slow) * VALU_CYCLES_FAST if fast >= slow else min(fast, slow) * VALU_CYCLES_PAIR + (slow - fast) * VALU_CYCLES_SLOW) / TRACE_ROUND_VALU
            best_serial = (fast * VALU_CYCLES_FAST + slow * VALU_CYCLES_SLOW) / TRACE_ROUND_VALU
            measured_cpi = simd_cycles / c["SQ_INSTS_VALU"]
            binding = {"bound": "valu", "unit": "lane-instr/s", "peak": VALU_PEAK_LANE_INSTR_PER_S, "achieved": float("%.4g" % useful),
                       "frac": round(useful / VALU_PEAK_LANE_INSTR_PER_S, 4),
                       "peak_derivation": "MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 per CU x 32 lanes per cycle (a 64-lane wave-instruction issues over 2 cycles: v_fma_f32 2 cyc) x 2.4 GHz = 7.86e13 lane-instructions/s (= the data sheet's 157.3 TFLOP/s FP32 with a fused multiply-add counted as two)",
                       "lane_utilisation": lane_util, "issue_frac": round(issue / 0.5, 4),
                       "issue_frac_definition": "vector wave-instructions per cycle and SIMD (SQ_INSTS_VALU / (1024 SIMDs x launch cycles at 2.4 GHz)) over the peak of one per 2 cycles; frac = issue_frac x lane_utilisation",
                       "mix_aware": {"round": {"valu": TRACE_ROUND_VALU, "slow_class": TRACE_ROUND_VALU_SLOW, "source": "static mix of one round of the closest-hit engine, profiles/trace_round_mix.json, tools/isa_loop_mix.py"},
                                     "class_cycles": {"fast": VALU_CYCLES_FAST, "slow": VALU_CYCLES_SLOW, "pair_interleaved": VALU_CYCLES_PAIR, "source": "profiles/r04_instruction_costs.txt (tools/microbench/valu_rates.hip on MI355X)"},
                                     "best_cycles_per_instruction": {"classes_overlapping": round(best_overlapped, 3), "classes_serial": round(best_serial, 3)}, "measured_cycles_per_instruction": round(measured_cpi, 3),
                                     "frac_classes_overlapping": round(best_overlapped / measured_cpi * lane_util, 4), "frac_classes_serial": round(best_serial / measured_cpi * lane_util, 4),
                                     "note": "what the round's OWN instruction mix could reach: half of its vector instructions are conversions / min / max / compares / selects (4.1 cycles per wave), the other half multiply-adds and logic (2.3), one of each interleaved takes 4.5 per pair. frac_* = (best cycles per instruction / measured) x lane utilisation"}}
            if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                lookups = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / seconds
                per_clock = l1_peak_lookups_per_clock(node_visits, triangle_tests)
                l1_peak = per_clock * CUS * CLOCK_HZ
                l1 = {"bound": "l1 (vector cache: tag look-ups)", "unit": "look-ups/s (TCP_TOTAL_CACHE_ACCESSES)", "peak": float("%.4g" % l1_peak), "achieved": float("%.4g" % lookups), "frac": round(lookups / l1_peak, 4),
                      "lookups_per_clock_and_cu": round(lookups / (CUS * CLOCK_HZ), 4), "peak_lookups_per_clock_and_cu": round(per_clock, 4),
                      "peak_derivation": "MEASURED, not from the guide (which states the L1's capacity only): tools/microbench/l1_lookup_rate.hip under rocprofv3 --pmc, profiles/r06_l1_lookup_rate.txt -- with every access hitting the unit retires 1.672 look-ups per clock and CU for a lane's 16-byte piece of its own 80-byte record (node fetch), 1.978 for 48-byte records (triangle fetch), 0.989 for 64 different lines, 0.913 for a coalesced stream; the peak here is the rate of THIS launch's mix of node and triangle fetches (5 per node visit, 3 per triangle test, from the counting launches) x 256 CUs x 2.4 GHz",
                      "lookups_per_vector_memory_instruction": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / max(c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0), 1.0), 2) if c.get("SQ_INSTS_VMEM_RD") else None}
                if c.get("TCP_GATE_EN1_sum"):
                    l1["clocked"] = round(c["TCP_GATE_EN1_sum"] / (CUS * seconds * CLOCK_HZ), 4); l1["clocked_definition"] = "TCP_GATE_EN1 (cycles the L1 is clocked, summed over the 256 CUs) / (256 x launch cycles at 2.4 GHz): the unit has a request in flight -- 0.92-0.99 in every shape of the microbenchmark, whatever its rate; NOT a utilisation (rounds 4-5 read it as one)"
                if c.get("TA_TA_BUSY_sum"):
                    l1["address_unit_busy"] = round(c["TA_TA_BUSY_sum"] / (CUS * seconds * CLOCK_HZ), 4)
                binding["l1"] = l1
                if c.get("TCP_TCC_READ_REQ_sum"):
                    l2_gbps = c["TCP_TCC_READ_REQ_sum"] * L1_TO_L2_REQUEST_BYTES / seconds / 1e9
                    binding["l2"] = {"bound": "l2 (read bandwidth)", "unit": "GB/s", "peak": L2_PEAK_GBPS, "achieved": round(l2_gbps, 1), "frac": round(l2_gbps / L2_PEAK_GBPS, 4),
                                     "peak_derivation": "MI355X_MICROARCH.md 'L2 (per XCD)': ~34.5 TB/s aggregate; achieved = TCP_TCC_READ_REQ x 64 B / launch time"}
            if out.get("traffic") and len(launch_ms):
                hbm_gbps = out["traffic"] / (float(np.mean(launch_ms)) * 1e-3) / 1e9
                binding["hbm"] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS, "achieved": round(hbm_gbps, 1), "frac": round(hbm_gbps / HBM_PEAK_GBPS, 4),
                                  "note": "memory-side bytes of the launch (roofline.traffic: FETCH_SIZE + WRITE_SIZE, calibrated) / its HIP-event duration: what HBM + Infinity Cache actually move"}
            fracs = {"valu (lane-instructions)": binding["frac"], "valu issue (wave-instructions)": binding["issue_frac"], "valu, mix-aware": binding["mix_aware"]["frac_classes_serial"]}
            for key in ("l1", "l2", "hbm"):
                if key in binding:
                    fracs[key] = binding[key]["frac"]
            binding["utilisation_by_unit"] = fracs
            binding["closest_to_its_roof"] = max(fracs, key=fracs.get)
            binding["note"] = ("The launch is branchy pointer chasing: no unit is saturated -- vector issue, what the round's instruction mix allows and the L1's look-up rate (against its MEASURED roof) all sit at 0.4-0.5, L2 and HBM at a tenth. "
                               "valu frac = useful lane-instructions against the guide's peak; it is low because (a) the triangle phase of a round runs with a fifth of the wave's lanes (lane_utilisation), (b) half the instructions are of the 4-cycle class (mix_aware), "
                               "(c) a round is two dependent memory round trips (counters.wave_cycles_waiting). DESIGN.md 4.1 has the experiments behind this reading")
            out["binding"] = binding
    if "TCC_HIT_sum" in trace and "TCC_MISS_sum" in trace and (trace["TCC_HIT_sum"][1] + trace["TCC_MISS_sum"][1]) > 0:
        out.setdefault("binding", {})["l2_hit_rate"] = round(trace["TCC_HIT_sum"][1] / (trace["TCC_HIT_sum"][1] + trace["TCC_MISS_sum"][1]), 4)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in trace and trace["TCP_TOTAL_CACHE_ACCESSES_sum"][1] > 0 and "TCP_TCC_READ_REQ_sum" in trace:
        out.setdefault("binding", {})["l1_hit_rate"] = round(1.0 - trace["TCP_TCC_READ_REQ_sum"][1] / trace["TCP_TOTAL_CACHE_ACCESSES_sum"][1], 4)
    out["kernels"] = kernels
    return out


# ---- per-stage rooflines ------------------------------------------------------------------------------
# ALGORITHMIC bytes of the stages besides traversal, per unit of work, as the scope table states them:
#   generate    28 B written per primary ray
#   sort        32 B read per ray (+ 28 B of path state from bounce 1 on), 60 B written per ray that reaches a material queue
#   shade       per queue entry read 32 B (60 B from bounce 1 on) + 96 B triangle + 48 B instance transform + 32 B material
#               + one trilinear albedo lookup on textured hits (8 taps x 8-byte BC1 blocks = 64 B; the extra probes of the
#               anisotropic lookup at bounce 0 are NOT counted), written 52 B per continuation ray + 44 B per shadow ray
#               + 3 x 16 B of albedo / normal / position at bounce 0
#   accumulate  48 B per pixel and AOV (sample read, accumulator read + write)
#   SVGF / TAA  every tap the kernels request, enumerated from the tap loops (reference SVGF.h:130-609, TAA.h:10-172);
#               `unique` next to it is the compulsory traffic (every image read / written once per kernel)
SVGF_TAP_BYTES = {   # kernel: (bytes requested per pixel, bytes per pixel if every image moved once); kernels as of round 4 (the a-trous taps are what the ALGORITHM asks for: the tiled kernel serves most of them from LDS)
    # reads radiance pair 32, g-buffer 16 + 8, 4 history (normal, depth) taps, 4 x 3 history taps, history length r/w; writes the pair, the
    # moments, the decoded (normal, depth), the variance pair, and for pixels with >= 4 frames of history the variance pass's copies
    "svgf_reproject": (32 + 16 + 8 + 4 * 16 + 4 * 48 + 8 + 32 + 16 + 16 + 8 + 40, 32 + 16 + 8 + 16 + 48 + 8 + 32 + 16 + 16 + 8 + 40),
    "svgf_variance":  (4, 4),             # only pixels with a history shorter than 4 frames have work (48 taps x 80 B each); the others read their history length
    "svgf_atrous":    (9 * 8 + 32 + 16 + 2 * 4 + 8 * (32 + 16) + 32 + 8 + 5, 32 + 16 + 8 + 32 + 8 + 5),   # 3x3 variance pairs, centre, 8 taps of (direct, indirect, normal + depth); + the history copy of pass 2
    "svgf_finalize":  (32 + 16 + 16 + 16 + 16 + 16 + 16 + 16 + 16 + 8, 184),
    # round 4: kernel_taa_finalize is folded into kernel_taa (reads the frame, its motion vector, 16 history + 8 neighbour taps; writes the next history, the
    # displayed image and the cleared motion vector)
    "taa":            (16 + 8 + 16 * 16 + 8 * 16 + 16 + 16 + 8, 16 + 8 + 16 + 16 + 16 + 8),
}


def stage_rooflines(grt, ctx, counters_per_sample, plan, steps, stream_gbps, textured_fraction=1.0):
    """[{stage, launches, ms_per_step, algorithmic_bytes_per_step, achieved, frac}] from the mode-3 launch timings of a repeat of
    the timed plan (rt_set_profiling(ctx, 3): HIP events around every launch) and the queue sizes per bounce."""
    def total(kind):
        t = grt.launch_timings(ctx, kind).astype(np.float64)
        return len(t), float(t.sum())
    nb = NUM_BOUNCES
    per_step = {"generate": 0.0, "sort": 0.0, "material_diffuse": 0.0, "material_plastic": 0.0, "material_dielectric": 0.0, "material_conductor": 0.0, "accumulate": 0.0}
    samples = [(first + j) % SPP for first, count, _ in plan for j in range(count)]
    for sidx in samples:
        c = counters_per_sample[sidx]
        entries = {"material_diffuse": c.diffuse, "material_plastic": c.plastic, "material_dielectric": c.dielectric, "material_conductor": c.conductor}
        per_step["generate"] += 28.0 * c.trace[0]
        for b in range(nb):
            shaded = sum(e[b] for e in entries.values())
            per_step["sort"] += c.trace[b] * (32.0 + (28.0 if b else 0.0)) + 60.0 * shaded
            nxt = c.trace[b + 1] if b + 1 < nb else 0
            for name, e in entries.items():
                if not e[b]:
                    continue
                share = e[b] / max(shaded, 1)
                texels = 64.0 * textured_fraction if name in ("material_diffuse", "material_plastic") else 0.0
                per_step[name] += e[b] * ((32.0 if b == 0 else 60.0) + 96.0 + 48.0 + 32.0 + texels + (48.0 if b == 0 else 0.0)) + share * (52.0 * nxt + 44.0 * c.shadow[b])
        per_step["accumulate"] += 48.0 * WIDTH * HEIGHT
    stages = []
    for name, nbytes in per_step.items():
        launches, ms = total(name)
        if launches == 0:
            continue
        gbps = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        stages.append({"stage": name, "launches": launches, "ms_per_step": round(ms / steps, 4), "algorithmic_bytes_per_step": round(nbytes / steps),
                       "achieved": round(gbps, 1), "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4), "ratio_to_measured_stream": round(gbps / stream_gbps, 4)})
    return stages


def config3_section(grt, scene, device, stream_gbps, frames=64):
    """BASELINE config 3 (Sponza 1080p, SVGF with 6 a-trous passes + TAA, one sample per filtered frame, the camera moving
    between frames): ms per filtered frame in the merged wavefront, and every filter kernel priced against the stream bandwidth."""
    svgf_tiles = int(os.environ.get("BENCH_SVGF_TILES", "1"))   # 0: the a-trous passes load every tap from the images (rt_set_svgf_tiles), for comparison
    grt.config_set(enable_svgf=1, enable_taa=1, num_atrous_iterations=6, svgf_lds_tiles=svgf_tiles)
    pt = grt.Pathtracer(scene, WIDTH, HEIGHT, device=device)
    try:
        pt.update()
        lib, ctx = grt.device_lib(), pt.ctx
        def frame(index):   # one filtered frame = one sample (rt_render_sample with SVGF on filters the frame when it has passed its last bounce)
            status = lib.rt_render_sample(ctx, index % SPP)
            if status != 0:
                raise RuntimeError(lib.rt_last_error(ctx).decode())
        for f in range(8):
            frame(f)
        lib.rt_synchronize(ctx)
        t0 = time.perf_counter()
        for f in range(frames):
            frame(f)
        lib.rt_synchronize(ctx)
        ms_frame = (time.perf_counter() - t0) / frames * 1e3
        c = pt.counters()
        rays = int(sum(c.trace[:NUM_BOUNCES]))
        grt.set_profiling(ctx, 3)
        for f in range(frames):
            frame(f)
        lib.rt_synchronize(ctx)
        px = WIDTH * HEIGHT
        kernels, filter_ms = [], 0.0
        for name, (tap_bytes, unique_bytes) in SVGF_TAP_BYTES.items():
            t = grt.launch_timings(ctx, name).astype(np.float64)
            if not len(t):
                continue
            per_frame = float(t.sum()) / frames
            filter_ms += per_frame
            passes = len(t) / frames
            gbps = tap_bytes * px * passes / (per_frame * 1e-3) / 1e9
            kernels.append({"kernel": "kernel_" + name, "launches_per_frame": round(passes, 2), "ms_per_frame": round(per_frame, 4),
                            "tap_bytes_per_pixel": tap_bytes, "unique_bytes_per_pixel": unique_bytes,
                            "tap_gbps": round(gbps, 1), "unit": "GB/s", "tap_ratio": round(gbps / stream_gbps, 4),
                            "frac_unique": round(unique_bytes * px * passes / (per_frame * 1e-3) / 1e9 / stream_gbps, 4)})
        if not any(k["kernel"] == "kernel_svgf_finalize" for k in kernels):
            # round 6: the last a-trous pass runs tiled and finalizes its pixels itself (svgf_finalize_pixel): it does not write the filtered pair and its variance mirror (40 B) and
            # moves kernel_svgf_finalize's bytes minus the pair and the (normal, depth) that kernel read back (184 - 48) -- spread over the frame's passes here
            for k in kernels:
                if k["kernel"] == "kernel_svgf_atrous" and k["launches_per_frame"] > 0:
                    unique = k["unique_bytes_per_pixel"] + (184 - 48 - 40) / k["launches_per_frame"]
                    k["unique_bytes_per_pixel"] = round(unique, 1); k["finalizes_in_its_last_pass"] = True
                    k["frac_unique"] = round(unique * px * k["launches_per_frame"] / (k["ms_per_frame"] * 1e-3) / 1e9 / stream_gbps, 4)
        trace_ms = float(grt.launch_timings(ctx, "trace").sum()) / frames
        grt.set_profiling(ctx, False)
        unique_total = sum(k["unique_bytes_per_pixel"] * k["launches_per_frame"] for k in kernels) * px
        return {"workload": "Sponza 1920x1080, SVGF (6 a-trous passes, spatial variance) + TAA, 1 sample per filtered frame, %d frames back to back in the merged wavefront (static camera; the moving-camera frames are parity-tested in tests/test_gpu_full_size.py)" % frames,
                "ms_per_filtered_frame": round(ms_frame, 3), "rays_per_frame": rays, "mrays_s": round(rays / ms_frame / 1e3, 1),
                "filter_ms_per_frame": round(filter_ms, 4), "traversal_ms_per_frame": round(trace_ms, 4),
                "filter_frac_of_stream_unique_bytes": round(unique_total / (filter_ms * 1e-3) / 1e9 / stream_gbps, 4) if filter_ms > 0 else None,
                "kernels": kernels, "svgf_lds_tiles": svgf_tiles,
                "note": "tap bytes = every tap the kernel requests (SVGF.h / TAA.h tap loops; most are served by LDS / L1: tap_ratio = tap bytes over the measured stream-read bandwidth is a re-use figure and exceeds 1), unique bytes = each image once: frac_unique = unique bytes over the measured stream-read bandwidth is the utilisation"}
    finally:
        pt.close()
        grt.config_set(enable_svgf=0, enable_taa=0)


def burst_section(grt, device, steps, warmup, config, workload):
    """The driver's frame loop -- `steps` samples as 4-sample frames, declared as ONE burst of up to 8 frames exactly like the timed region of main() -- on a
    scene staged with other settings, in the same process on the same GPU; wall time around submit + drain."""
    import ctypes
    scene = build_scene(grt)
    grt.config_set(**config)
    pt = grt.Pathtracer(scene, WIDTH, HEIGHT, device=device)
    try:
        pt.update()
        lib, ctx = grt.device_lib(), pt.ctx
        lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.rt_synchronize.argtypes = [ctypes.c_void_p]
        frames = (steps + SPP - 1) // SPP
        def submit(count):
            base = grt.submissions_completed(ctx)
            for _ in range(count):
                if lib.rt_render_samples(ctx, 0, SPP) != 0:
                    raise RuntimeError(lib.rt_last_error(ctx).decode())
            while grt.submissions_completed(ctx) - base < count:
                grt.advance(ctx)
            lib.rt_synchronize(ctx)
        submit(max((warmup + SPP - 1) // SPP, 1))
        lib.rt_render_samples(ctx, 0, 1); c = pt.counters(); rays = int(sum(c.trace[:NUM_BOUNCES]))   # closest-hit rays of one sample
        lib.rt_synchronize(ctx)
        if frames > 1:
            grt.set_frame_pipelining(ctx, True)
            grt.set_stream_batch(ctx, min(frames, 8) * SPP * WIDTH * HEIGHT)
        submit(frames)                                                                              # untimed: the burst's queues are sized on first use
        grt.set_profiling(ctx, 2)
        t0 = time.perf_counter()
        submit(frames)
        elapsed = time.perf_counter() - t0
        trace_ms = float(grt.launch_timings(ctx, 0).sum())
        grt.set_profiling(ctx, False)
        done = frames * SPP
        return {"workload": workload, "steps": done, "ms_per_step": round(elapsed / done * 1e3, 3), "mrays_s": round(rays * done / elapsed / 1e6, 1),
                "traversal_ms_per_step": round(trace_ms / done, 4), "rays_per_step": rays, "skip_behind_hit": bool(pt.skip_behind_hit)}
    finally:
        pt.close(); scene.close()


def reference_layout_section(grt, device, steps, warmup):
    """The same frame loop on the REFERENCE'S acceleration-structure layout (config merge_static 0: one CWBVH per mesh under a
    CWBVH TLAS, 384 instance entries), in the same process on the same GPU, so that the line the driver records carries both
    layouts: the headline is measured on the flattened tree, a layout the reference does not have."""
    global MERGE_STATIC
    keep = MERGE_STATIC
    MERGE_STATIC = 0
    try:
        return burst_section(grt, device, steps, warmup, {}, "the same scene, camera, samples and frame loop (one declared burst) with merge_static = 0: one CWBVH per mesh under a CWBVH TLAS, walked the reference's way (the reference's layout, Integrator.cpp:101-283)")
    finally:
        MERGE_STATIC = keep


def viewpoint_free_section(grt, device, steps, warmup):
    """The headline's tree is seated by 1 M sample rays of which a quarter are paths from the benchmark's own camera (config static_slot_learning_viewpoint = 1: a
    renderer seats the tree it is about to look at). This is the same frame loop on the tree seated WITHOUT any camera rays: what a viewpoint the seating
    never saw pays."""
    return burst_section(grt, device, steps, warmup, dict(static_slot_learning_viewpoint=0), "the same scene, camera, samples and frame loop (one declared burst); the flattened tree's children seated by sample rays from free-space points and surfaces only (static_slot_learning_viewpoint = 0: no ray from the benchmark's camera)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)   # 16 frames of 4 spp: the 9 fill / drain iterations of the wavefront are a small part of the launches
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the hardware-counter passes (rocprofv3 --pmc re-runs of this benchmark: HBM traffic, VALU busy; N = 1 only)")
    ap.add_argument("--no-config3", action="store_true", help="skip the SVGF + TAA frames of BASELINE config 3 (N = 1 only)")
    ap.add_argument("--burst", type=int, default=1, help="1 (default): the run's submissions are declared as bursts of up to 8 (rt_set_stream_batch) whose frames share iterations; 0: they follow one another through the wavefront")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-stage rooflines (a repeat of the timed plan with events around every launch)")
    ap.add_argument("--no-reference-layout", action="store_true", help="skip the pass over the reference's acceleration-structure layout (merge_static 0; N = 1 only)")
    ap.add_argument("--no-tile-split-bound", action="store_true", help="skip the one-GPU bound of the tile split (child runs of this command with --emulate-world 2 / 4 / 8; N = 1 only)")
    ap.add_argument("--no-povs", action="store_true", help="skip the sweep over the reference's 9 fixed Sponza points of view (N = 1 only)")
    ap.add_argument("--emulate-world", type=int, default=0, help="debug: render only rank 0's tiles of an N-GPU split on one GPU (no collective), to exercise the N > 1 code path")
    ap.add_argument("--merge-static", type=int, default=1, help="1 (default): the 382 instances of Sponza that stand still with the identity transform are flattened into one bottom-level tree (config merge_static); 0: one BLAS per mesh under the TLAS, the reference's layout")
    ap.add_argument("--expand-textures", type=int, default=1, help="1 (default): BC1 textures are decoded once, at upload, into 64-byte blocks of texels (rt_set_texture_expansion); 0: the 8-byte blocks stay compressed and every texel fetch decodes one")
    ap.add_argument("--exchange", choices=["native", "torch"], default="native", help="N > 1: who runs the per-frame all-gather. native (default): the library's own frame exchange, ncclAllGather on the context's stream through rt_comm_init_rank / rt_all_gather_framebuffer (torch.distributed only carries the 128-byte communicator id and the timing reductions); torch: dist.all_gather_into_tensor on torch's stream around rt_pack_pixels / rt_unpack_pixels. Falls back to torch when the library cannot set up its communicator")
    ap.add_argument("--batch", type=int, default=SPP, help="samples per pixel per submission (rt_render_samples), 1..%d" % SPP)
    ap.add_argument("--samples-in-flight", type=int, default=0, help="samples per pixel rendered concurrently (rt_set_samples_in_flight)")
    args = ap.parse_args()
    global MERGE_STATIC, EXPAND_TEXTURES
    EXPAND_TEXTURES = args.expand_textures
    MERGE_STATIC = args.merge_static

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist
    import gpu_raytracer_amd as grt
    import importlib
    parallel = importlib.import_module("gpu_raytracer_amd.parallel")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # debug only: BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 runs the N-rank flow (rendezvous, tile split,
    # stream hand-over, reductions) with every rank on GPU 0 and the gather staged through the host
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    scene = build_scene(grt)
    pt = grt.Pathtracer(scene, WIDTH, HEIGHT, device=local_rank)
    pt.update()
    if pt.static_geometry_whole_scene:
        TRACE_KERNEL[0] = "kernel_trace_stream_bvh8_flat_skip" if pt.skip_behind_hit else "kernel_trace_stream_bvh8_flat"   # the engine variant without TLAS / instance code (and its skipping walk: rt_set_skip_behind_hit)
    flatten_build_s = pt.static_geometry_build_seconds if pt.static_geometry_members else 0.0
    closed = False
    lib = grt.device_lib()
    ctx = pt.ctx
    scheduler = os.environ.get("BENCH_SCHEDULER", "merged")     # "slots": the per-submission launch chains, for comparison
    grt.set_scheduler(ctx, scheduler)
    split_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    if args.samples_in_flight <= 0:   # slot scheduler only (profiles/r01_sample_batching.txt): the smaller a rank's share, the more submissions
        args.samples_in_flight = 3 if split_world == 1 else (4 if split_world == 2 else 8)
    grt.set_samples_in_flight(ctx, args.samples_in_flight)
    split = parallel.TileSplit(rank, split_world, WIDTH, HEIGHT)
    device = torch.device("cuda", local_rank)
    merged = scheduler == "merged"
    if merged and split_world > 1:
        grt.set_frame_pipelining(ctx, True)   # pack / unpack follow the frames completed so far; later frames keep the wavefront full

    # this rank's tiles, rendered as scan-order pixel ranges; the gather buffers live in torch
    packed = torch.zeros((split.local_pixels, 4), dtype=torch.float32, device=device)
    gathered = torch.zeros((split_world * split.local_pixels, 4), dtype=torch.float32, device=device)
    import ctypes
    lib.rt_pack_pixels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_unpack_pixels.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_synchronize.argtypes = [ctypes.c_void_p]
    lib.rt_stream_wait_for_context.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rt_context_wait_for_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.rt_set_pixel_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_render_samples.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]

    def check(status):
        if status != 0:
            raise RuntimeError(lib.rt_last_error(ctx).decode())

    if split_world == 1:
        check(lib.rt_set_pixel_range(ctx, 0, WIDTH * HEIGHT))
    else:
        check(lib.rt_set_pixel_tiles(ctx, split.tile_pixels, rank, split_world))

    # The library's own frame exchange (include/gpu_raytracer_amd.h: rt_comm_*): rank 0 makes the communicator id, torch.distributed
    # hands the 128 bytes to the others (its only part in the data path), every rank joins with ncclCommInitRank. From then on a
    # completed frame is exchanged by ONE call that packs, all-gathers over RCCL on the context's stream and unpacks.
    exchange = "torch"
    # (ranks that share a GPU -- BENCH_SHARE_GPU, a one-GPU test box -- can only take it with a collective library that accepts a device twice: the loopback
    # stand-in of the test suite, GRT_COLLECTIVE_LIBRARY=tests/support/libloopback_ccl.so; RCCL itself refuses)
    if world > 1 and args.exchange == "native" and ((backend == "nccl" and not os.environ.get("BENCH_SHARE_GPU")) or os.environ.get("GRT_COLLECTIVE_LIBRARY")):
        lib.rt_comm_unique_id.argtypes = [ctypes.c_void_p]
        lib.rt_comm_init_rank.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        lib.rt_all_gather_framebuffer.argtypes = [ctypes.c_void_p]
        uid = ctypes.create_string_buffer(128)
        made = lib.rt_comm_unique_id(uid) if rank == 0 else 0
        carrier = torch.tensor(list(uid.raw) + [made], dtype=torch.int32, device=device)
        dist.broadcast(carrier, 0)
        values = carrier.cpu().tolist()
        joined = 1
        if values[128] == 0:
            joined = lib.rt_comm_init_rank(ctx, bytes(bytearray(v & 0xff for v in values[:128])), rank, world)
        verdict = torch.tensor([joined], dtype=torch.int32, device=device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MAX)     # every rank or none
        if int(verdict.item()) == 0:
            exchange = "native"
        else:
            if joined == 0:
                lib.rt_comm_destroy.argtypes = [ctypes.c_void_p]; lib.rt_comm_destroy(ctx)
            if rank == 0:
                sys.stderr.write("bench: the library's RCCL communicator could not be set up (%s); exchanging through torch.distributed\n" % lib.rt_last_error(ctx).decode(errors="replace"))

    def exchange_frame():
        """The one data-path collective: this rank's accumulated tiles -> every rank's final framebuffer. Stream-ordered in
        both directions, the host does not block: pack after the previous all_gather has read `packed`, all_gather after
        the pack, unpack after the all_gather; later frames are already being traced meanwhile."""
        if split_world == 1 or os.environ.get("BENCH_NO_GATHER"):
            return
        if exchange == "native":
            check(lib.rt_all_gather_framebuffer(ctx))
            return
        if world == 1 and not os.environ.get("BENCH_EMULATE_THROUGH_TORCH"):
            # --emulate-world: what the native exchange does on this rank minus the collective itself -- pack this rank's tiles and unpack the whole frame, both on the
            # context's stream, no hand-over to another stream (the torch path below costs the emulation 0.4 ms of host round trips per burst that a real rank, which
            # exchanges through rt_all_gather_framebuffer, does not have: profiles/r04_tile_split_rank_timeline.txt)
            check(lib.rt_pack_pixels(ctx, gathered.data_ptr(), split.tile_pixels, rank, split_world, split.tiles_per_rank))
            check(lib.rt_unpack_pixels(ctx, gathered.data_ptr(), split.tile_pixels, split_world, split.tiles_per_rank))
            return
        torch_stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.rt_context_wait_for_stream(ctx, torch_stream))
        check(lib.rt_pack_pixels(ctx, packed.data_ptr(), split.tile_pixels, rank, split_world, split.tiles_per_rank))
        check(lib.rt_stream_wait_for_context(ctx, torch_stream))
        if world > 1 and backend == "nccl":
            dist.all_gather_into_tensor(gathered, packed)
        elif world > 1:
            host = torch.empty(gathered.shape, dtype=gathered.dtype)
            dist.all_gather_into_tensor(host, packed.cpu())
            gathered.copy_(host)
        else:
            gathered[:split.local_pixels].copy_(packed)   # --emulate-world: stand-in for the collective
        if world > 1:   # the gathered tiles become the final framebuffer of this rank (every rank ends up with the whole frame)
            check(lib.rt_context_wait_for_stream(ctx, torch_stream))
            check(lib.rt_unpack_pixels(ctx, gathered.data_ptr(), split.tile_pixels, split_world, split.tiles_per_rank))

    def submissions(steps):
        """[(first sample index, count, completes a frame)] covering exactly `steps` samples, frame by frame."""
        out, k = [], 0
        while k < steps:
            first = k % SPP
            count = min(SPP - first, steps - k, args.batch)
            k += count
            out.append((first, count, first + count == SPP or k == steps))
        return out

    def run(plan):
        """Submits the plan; a frame is exchanged as soon as its last submission is complete. Under the merged scheduler a
        submission completes num_bounces - 1 iterations after it was made (rt_submissions_completed tells), so the loop
        ends by advancing the wavefront without new samples; under the slot scheduler the exchange is simply ordered
        behind the submission."""
        if not merged:
            for first, count, frame_complete in plan:
                check(lib.rt_render_samples(ctx, first, count))
                if frame_complete:
                    exchange_frame()
            return
        base = grt.submissions_completed(ctx)
        handled = 0
        def drain_completions():
            nonlocal handled
            done = grt.submissions_completed(ctx) - base
            while handled < done:
                if plan[handled][2]:
                    exchange_frame()
                handled += 1
        for first, count, _ in plan:
            check(lib.rt_render_samples(ctx, first, count))
            drain_completions()
        while handled < len(plan):
            grt.advance(ctx)
            drain_completions()

    # ---- untimed priming + warm-up: queues, sample frames and streams are allocated on first use (GBs of hipMalloc)
    run(submissions(max(args.warmup, SPP)))
    check(lib.rt_synchronize(ctx))

    # ---- untimed statistics pass: rays per sample and the work counters of the trace kernels ------
    rays_per_sample, shadow_per_sample, alg_closest_per_sample, alg_shadow_per_sample, trace_rays_stat, counters_per_sample = [], [], [], [], [], []
    grt.set_trace_statistics(ctx, True)
    for s in range(SPP):
        check(lib.rt_render_samples(ctx, s, 1))
        c = pt.counters()
        stats = grt.get_trace_statistics(ctx)
        counters_per_sample.append(c)
        rays_per_sample.append(sum(c.trace[:NUM_BOUNCES])); shadow_per_sample.append(sum(c.shadow[:NUM_BOUNCES]))
        alg_closest_per_sample.append(stats["closest"]["algorithmic_bytes"]); alg_shadow_per_sample.append(stats["shadow"]["algorithmic_bytes"])
        trace_rays_stat.append(stats)
    # ... and, for the merged scheduler, of every traversal launch of the plan that is about to be timed
    plan = submissions(args.steps)
    # The run is a batch job: its frames are submitted back to back and nothing is read before the last one. Declared to the library
    # (rt_set_stream_batch), up to 8 of them enter the wavefront together and walk their bounces side by side -- num_bounces iterations for
    # the lot -- instead of one after the other through len(plan) + num_bounces - 1 iterations whose first and last num_bounces - 1 are
    # partly filled (20 steps: 10 iterations instead of 14; measured 1.49 -> 1.42 ms per step, profiles/r04_burst.txt). Every launch then
    # carries ONE bounce of 8 frames instead of every bounce of one: 32 steps as one burst run at 1.365 ms per step where the pipelined steady
    # state of a long run reaches 1.39.
    burst = 0
    if merged and args.burst and len(plan) > 1:   # (every rank of a tile split declares its share the same way: its small submissions shared iterations already, up to 8.3 M paths)
        burst = min(len(plan), 8)
        grt.set_frame_pipelining(ctx, True)
        grt.set_stream_batch(ctx, sum(count for _, count, _ in plan[:burst]) * (WIDTH * HEIGHT if split_world == 1 else split.local_pixels))
    launch_bytes = None
    if merged:
        run(plan)
        history = grt.trace_statistics_history(ctx).astype(np.int64)
        rows = np.vstack([np.zeros((1, 10), np.int64), history])
        per_launch = rows[1:] - rows[:-1]
        launch_bytes = np.array([sum(grt.algorithmic_bytes(r)) for r in per_launch], np.float64)
        launch_closest_bytes = np.array([grt.algorithmic_bytes(r)[0] for r in per_launch], np.float64)
        launch_rays = per_launch[:, 4] + per_launch[:, 9]
    grt.set_trace_statistics(ctx, False)

    # ---- profiled pass (HIP events per stage): kernel time of the stages of one 4-spp frame, nothing else in flight
    grt.set_profiling(ctx, True)
    stage_ms = {}
    for rep in range(2):
        for first, count, _ in submissions(SPP):
            check(lib.rt_render_samples(ctx, first, count))
        c = pt.counters()
        if rep == 1:
            for k in ("ms_generate", "ms_trace", "ms_sort", "ms_shade", "ms_shadow", "ms_post"):
                stage_ms[k] = getattr(c, k) / SPP
    alone_trace_ms = stage_ms["ms_trace"] * SPP
    grt.set_profiling(ctx, False)

    # ---- timed region: exactly K steps -------------------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    check(lib.rt_synchronize(ctx))
    # HIP events around EVERY traversal launch of the timed region, each on the stream the launch runs on -- at N = 1, the configuration the roofline record is
    # quoted on. A rank of a tile split runs launches an N-th the size: the two event packets per launch are 0.38 ms of its 6 ms burst at N = 8 (measured:
    # 0.303 -> 0.284 ms per step, profiles/r05_tile_split_bound.txt), so there the timed region runs bare and an untimed repeat behind it carries the events.
    events_in_timed_region = split_world == 1 and not os.environ.get("BENCH_NO_LAUNCH_TIMING")
    grt.set_profiling(ctx, 2 if events_in_timed_region else 0)
    t0 = time.perf_counter()
    run(plan)
    check(lib.rt_synchronize(ctx))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if not events_in_timed_region and not os.environ.get("BENCH_NO_LAUNCH_TIMING"):
        grt.set_profiling(ctx, 2)
        run(plan)
        check(lib.rt_synchronize(ctx))
    launch_ms = grt.launch_timings(ctx, 0).astype(np.float64)
    shadow_launch_ms = grt.launch_timings(ctx, 1).astype(np.float64)
    grt.set_profiling(ctx, False)

    # ---- untimed repeat of the plan with HIP events around EVERY launch: what each stage of the step costs
    stages_raw = None
    if merged and world == 1 and split_world == 1 and not args.no_stages:
        grt.set_profiling(ctx, 3)
        run(plan)
        check(lib.rt_synchronize(ctx))
        stages_raw = True   # (the timings are read below, next to the stream bandwidth they are priced against)

    rays_plan = float(sum(rays_per_sample[(first + j) % SPP] for first, count, _ in plan for j in range(count)))
    shadow_plan = float(sum(shadow_per_sample[(first + j) % SPP] for first, count, _ in plan for j in range(count)))
    local = torch.tensor([elapsed, rays_plan, shadow_plan], dtype=torch.float64, device=device)
    if world > 1:
        mx = local.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = local.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0]); rays_plan = float(sm[1]); shadow_plan = float(sm[2])

    if rank == 0:
        value = rays_plan / elapsed / 1e6
        alg_closest_plan = float(sum(alg_closest_per_sample[(first + j) % SPP] for first, count, _ in plan for j in range(count)))
        alg_shadow_plan = float(sum(alg_shadow_per_sample[(first + j) % SPP] for first, count, _ in plan for j in range(count)))
        def spread(values):
            v = np.sort(np.asarray(values, np.float64))
            return {"min": round(float(v[0]), 4), "median": round(float(v[len(v) // 2]), 4), "max": round(float(v[-1]), 4)} if len(v) else None
        # bound / achieved / peak / unit / frac: the unit of the dominant kernel that sits closest to its roof, filled in from the hardware counters of the same
        # command (pmc_section: roofline.binding); every fraction in the record is <= 1. SURVEY 8d's figure -- algorithmic bytes over launch time over the HBM
        # peak -- is cache-served bytes here (a 31 MB tree) and lives under its own name, algorithmic_bytes_over_hbm_peak.
        roofline = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None}
        if merged and len(launch_ms) == len(launch_bytes):
            # the dominant kernel: ONE fused traversal launch per iteration (closest-hit rays of every submission in flight +
            # the shadow rays of the previous iteration); bytes per launch from the counting variant of the same launches
            achieved = launch_bytes.sum() / (launch_ms.sum() * 1e-3) / 1e9
            big = launch_rays >= 0.5 * launch_rays.max()        # the steady-state launches (fill and drain iterations excluded)
            per_launch_gbps = launch_bytes / np.maximum(launch_ms, 1e-6) / 1e6
            roofline.update({
                "kernel": TRACE_KERNEL[0], "algorithmic_gbps": round(achieved, 1), "algorithmic_bytes_over_hbm_peak": round(achieved / HBM_PEAK_GBPS, 4),
                "algorithmic_bytes_over_hbm_peak_is": "SURVEY 8d's figure: ALGORITHMIC bytes / launch time / the 8 TB/s HBM peak. The bytes are served by the caches (binding.l1_hit_rate, l2_hit_rate), so this is a yardstick that can exceed 1, NOT a utilisation; the utilisations (all <= 1) are roofline.frac (the binding unit's), hbm_frac (counter bytes over the same peak), algorithmic_bytes_over_l1_stream_rate and the rest of roofline.binding",
                "algorithmic_bytes_over_l1_stream_rate": round(achieved / L1_STREAM_GBPS, 4),
                "launches": int(len(launch_ms)), "algorithmic_bytes_per_launch": round(float(launch_bytes.mean())), "avg_launch_ms": round(float(launch_ms.mean()), 4),
                "launch_ms": spread(launch_ms), "launch_gbps": spread(per_launch_gbps),
                "steady_state": {"launches": int(big.sum()), "algorithmic_gbps": round(float(launch_bytes[big].sum() / (launch_ms[big].sum() * 1e-3) / 1e9), 1),
                                 "ratio_to_hbm_peak_cache_served": round(float(launch_bytes[big].sum() / (launch_ms[big].sum() * 1e-3) / 1e9 / HBM_PEAK_GBPS), 4),
                                 "algorithmic_bytes_over_l1_stream_rate": round(float(launch_bytes[big].sum() / (launch_ms[big].sum() * 1e-3) / 1e9 / L1_STREAM_GBPS), 4),
                                 "launch_gbps": spread(per_launch_gbps[big]), "rays_per_launch": int(launch_rays[big].mean())},
                "closest_hit_share_of_bytes": round(float(launch_closest_bytes.sum() / launch_bytes.sum()), 3),
                "time_share_of_step": round(float(launch_ms.sum() / (elapsed * 1e3)), 3),
                "note": ("" if events_in_timed_region else "N > 1: the launch durations come from an untimed repeat of the timed plan (the timed region itself runs without per-launch events). ") + "achieved = sum of the algorithmic bytes (SURVEY 8d; counted by the counting variant on the same launches) of ALL traversal launches of the timed region / sum of their HIP-event durations; one launch at a time is resident (single stream), so a duration is the kernel's own. steady_state = launches with at least half the rays of the largest. Working set (%.1f MB nodes + %.1f MB triangle positions%s) is L2 / Infinity-Cache resident: algorithmic bytes >> DRAM traffic" % (pt.array("bvh8_nodes").size / 1e6, pt.array("triangles").size // 24 * 48 / 1e6, ", the per-mesh trees and the flattened tree over copies of their triangles; rays only touch the latter" if pt.static_geometry_members else ""),
            })
        else:
            total_ms = float(launch_ms.sum()) if len(launch_ms) else float("nan")
            achieved = alg_closest_plan / (total_ms * 1e-3) / 1e9
            roofline.update({"kernel": "kernel_trace_bvh8", "algorithmic_gbps": round(achieved, 1), "algorithmic_bytes_over_hbm_peak": round(achieved / HBM_PEAK_GBPS, 4), "launches": int(len(launch_ms)),
                             "launch_ms": spread(launch_ms), "shadow_kernel": {"kernel": "kernel_trace_shadow_bvh8", "launches": int(len(shadow_launch_ms)), "launch_ms": spread(shadow_launch_ms),
                             "achieved": round(alg_shadow_plan / (max(float(shadow_launch_ms.sum()), 1e-9) * 1e-3) / 1e9, 1)},
                             "note": "slot scheduler: launches of different submissions queue behind each other, the event durations include that wait"})
        roofline.update({
            "traversal_share": {"algorithmic_gbps_over_the_whole_step": round((alg_closest_plan + alg_shadow_plan) / elapsed / 1e9, 1),
                                "algorithmic_bytes_over_hbm_peak": round((alg_closest_plan + alg_shadow_plan) / elapsed / 1e9 / HBM_PEAK_GBPS, 4),
                                "note": "all traversal bytes (closest + shadow) / wall time of the timed region: a lower bound that charges every other kernel to the traversal"},
            "bytes_per_ray": round(sum(alg_closest_per_sample) / max(sum(rays_per_sample), 1), 1),
            "bytes_per_shadow_ray": round(sum(alg_shadow_per_sample) / max(sum(shadow_per_sample), 1), 1),
            "nodes_per_ray": round(sum(s["closest"]["nodes"] for s in trace_rays_stat) / max(sum(rays_per_sample), 1), 2),
            "triangles_per_ray": round(sum(s["closest"]["triangles"] for s in trace_rays_stat) / max(sum(rays_per_sample), 1), 2),
            "nodes_per_shadow_ray": round(sum(s["shadow"]["nodes"] for s in trace_rays_stat) / max(sum(shadow_per_sample), 1), 2),
            "triangles_per_shadow_ray": round(sum(s["shadow"]["triangles"] for s in trace_rays_stat) / max(sum(shadow_per_sample), 1), 2),
            "frame_alone_trace_ms_per_step": round(alone_trace_ms / SPP, 4),
            "measured_stream_read_gbps": round(grt.measure_stream_bandwidth(ctx, 1 << 30, 5), 1),
        })
        stream_gbps = roofline["measured_stream_read_gbps"]
        if roofline.get("algorithmic_gbps"):   # the denominator SURVEY 8d names: what a streaming read reaches on THIS GPU
            roofline["algorithmic_bytes_over_measured_stream_cache_served"] = round(roofline["algorithmic_gbps"] / stream_gbps, 4)
        if stages_raw:
            trace_ms = float(grt.launch_timings(ctx, "trace").sum())
            stages = stage_rooflines(grt, ctx, counters_per_sample, plan, args.steps, stream_gbps)
            if merged and launch_bytes is not None:
                stages.insert(1, {"stage": "traversal", "launches": int(len(launch_bytes)), "ms_per_step": round(trace_ms / args.steps, 4),
                                  "algorithmic_bytes_per_step": round(float(launch_bytes.sum()) / args.steps),
                                  "algorithmic_gbps": round(float(launch_bytes.sum()) / (trace_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                  "algorithmic_bytes_over_hbm_peak": round(float(launch_bytes.sum()) / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                  "algorithmic_bytes_over_l1_stream_rate": round(float(launch_bytes.sum()) / (trace_ms * 1e-3) / 1e9 / L1_STREAM_GBPS, 4)})
            roofline["stages"] = stages
            roofline["stages_note"] = "a repeat of the timed plan with HIP events around every launch (rt_set_profiling 3); frac = algorithmic bytes (SURVEY 8d formulas, bench.py stage_rooflines) / stage time / the 8 TB/s HBM peak, ratio_to_measured_stream = the same over what a 1 GiB streaming read reaches on this GPU (a read-only probe: a read + write stage such as accumulate can pass it); the traversal's bytes are cache-served, its entry carries a labelled ratio and the L1 fraction instead of frac; the sort kernel is a gather chain, the material kernels are bound by instruction issue: their fraction is a yardstick, not the limit"
            grt.set_profiling(ctx, False)
        result = {
            "metric": "Mrays/s (primary+secondary) + ms/frame, Sponza 1920x1080 4spp BVH8", "value": round(value, 1), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": ("Crytek Sponza as the reference ships it: geometry and the 19 diffuse maps of Data/Sponza (fixed seeds, fixed camera)" if grt.reference_sponza_textures_installed() else "synthetic"),
            "config": {
                "workload": "Sponza (Crytek, 262 687 triangles, 384 instances) 1920x1080, samples 0..3 (4 spp), BVH8/CWBVH, diffuse + roughplastic(odd materials, alpha 0.3), NEE+MIS+RR, 10 bounces, constant white sky, mipmapping on, 19 diffuse textures at the reference's dimensions (1024x1024 + mips, BC1 block-compressed as the reference does by default, decoded once at upload (rt_set_texture_expansion); " + ("the reference's own texture files" if grt.reference_sponza_textures_installed() else "texels replicated 4x4 from the quarter-size maps that travel with the repo") + "), the 5 maps missing upstream are the reference's 1x1 fallback texel",
                "step": "one sample per pixel for the whole frame; the 4 samples of a frame are one submission (rt_render_samples); the submissions feed one merged wavefront (see burst: declared as a burst they enter it together and every launch carries one bounce of all of them; one by one, every launch carries the rays of all submissions in flight)",
                "scheduler": scheduler,
                "acceleration_structure": (("%d of %d instances (all that stand still) flattened into one CWBVH of %d triangle copies" + (", no TLAS: rays start inside the tree (rt_set_static_geometry)" if pt.static_geometry_whole_scene else ", one TLAS leaf beside the other instances") + ", hits reported as the scene's own instances and triangles (rt_upload_triangle_aliases); tree built on the host in %.2f s (SAH object + spatial splits, all threads, then the children of every node seated in the octant slots by what 1 M seeded sample rays -- a quarter camera paths, half from free-space points, a quarter surface rays -- say: config static_slot_learning_rays, host/SlotOrder.cpp; at scene load, not in the timed region); --merge-static 0 runs the reference's layout")
                                           % (pt.static_geometry_members, scene.mesh_count, int((pt.array("alias_mesh_ids") >= 0).sum()), pt.static_geometry_build_seconds)) if pt.static_geometry_members else "one CWBVH per mesh under a CWBVH TLAS (the reference's layout)",
                "rays_per_step": round(rays_plan / args.steps), "shadow_rays_per_step": round(shadow_plan / args.steps),
                "mrays_s_including_shadow": round((rays_plan + shadow_plan) / elapsed / 1e6, 1),
                "ms_per_4spp_frame": round(elapsed / args.steps * SPP * 1e3, 3),
                "emulated_world": args.emulate_world, "ranks": (dist.get_world_size() if world > 1 else 1),
                "parallelism": ("tile-split x%d + one RCCL all-gather of the accumulated float4 frame per %d-spp frame, unpacked into every rank's framebuffer; exchange: %s" % (world, SPP, "the library's own (rt_all_gather_framebuffer: ncclAllGather on the context's stream)" if exchange == "native" else "torch.distributed all_gather_into_tensor around rt_pack_pixels / rt_unpack_pixels")) if world > 1 else "single GPU",
                "samples_per_submission": args.batch, "submissions_in_flight": ("num_bounces (merged wavefront)" if merged else args.samples_in_flight),
                # rt_set_frame_pipelining: the submissions of a tile split are small, up to 8 of them share one iteration of the wavefront
                "submissions_per_iteration": (burst or (min(8, -(-WIDTH * HEIGHT * SPP // max(1, split.local_pixels * args.batch))) if (merged and split_world > 1) else 1)),
                "burst": ("the run's submissions are declared as bursts of %d (rt_set_stream_batch): each burst enters the wavefront together and takes %d iterations; --burst 0 lets submissions follow one another (%d iterations for as many)" % (burst, NUM_BOUNCES, burst + NUM_BOUNCES - 1)) if burst else "off",
                "stage_ms_per_step_one_frame_alone": {k: round(v, 3) for k, v in stage_ms.items()},
            },
            "roofline": roofline,
        }
        if world == 1 and split_world == 1 and not args.no_povs:
            # the reference's perf harness (Util/PerfTest.h): the same frame loop at its 9 fixed points of view
            povs = []
            camera_before = scene.get_camera()
            for position, rotation in SPONZA_POVS:
                scene.set_camera(position, rotation); pt.update()
                run(submissions(2 * SPP)); check(lib.rt_synchronize(ctx))     # untimed: first frames from the new camera
                grt.set_trace_statistics(ctx, False)
                pov_plan = submissions(4 * SPP)
                seating_before = (pt.reseat_pending, pt.reseats_completed)
                t0 = time.perf_counter()
                run(pov_plan)
                check(lib.rt_synchronize(ctx))
                ms = (time.perf_counter() - t0) / (4 * SPP) * 1e3
                check(lib.rt_render_samples(ctx, 0, 1)); c = pt.counters()
                povs.append({"ms_per_step": round(ms, 3), "rays_per_step": int(sum(c.trace[:NUM_BOUNCES])), "mrays_s": round(sum(c.trace[:NUM_BOUNCES]) / ms / 1e3, 1),
                             "seating_in_the_making": bool(seating_before[0]), "seatings_completed": int(seating_before[1])})   # (the tree is seated again beside the frame loop once the camera has travelled: DESIGN.md 4.6)
            scene.set_camera(tuple(camera_before[0]), tuple(camera_before[1])); pt.update()
            ms_all = np.array([p["ms_per_step"] for p in povs]); mr_all = np.array([p["mrays_s"] for p in povs])
            result["povs"] = {"source": "Util/PerfTest.h:30-40 (povs_sponza), 16 steps each", "per_pov": povs,
                              "ms_per_step_avg": round(float(ms_all.mean()), 3), "ms_per_step_stddev": round(float(ms_all.std()), 3),
                              "mrays_s_avg": round(float(mr_all.mean()), 1), "mrays_s_stddev": round(float(mr_all.std()), 1)}
        # (after the sweep: the oracle's 32 OpenMP threads linger on the host cores for a while, and a frame loop that shares them with the seating worker of the sweep
        # once showed one point of view at 4.7 ms per step for 1.25: profiles/r06_bench_run27.json)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(grt, pt, scene)
        if world == 1 and split_world == 1 and merged and not args.no_config3 and (not os.environ.get("BENCH_PMC_CHILD") or os.environ.get("BENCH_PMC_CONFIG3")):   # (tools/svgf_counters.py profiles this section)
            pt.close(); pt = None   # (its queues and sample frames go back first)
            result["config3"] = config3_section(grt, scene, local_rank, stream_gbps)
        if world == 1 and split_world == 1 and merged and args.merge_static and not args.no_reference_layout and not os.environ.get("BENCH_PMC_CHILD"):
            if pt is not None:
                pt.close(); pt = None
            result["flatten_build_s"] = round(float(flatten_build_s), 3)
            result["reference_layout"] = reference_layout_section(grt, local_rank, args.steps, args.warmup)
            result["seating_without_viewpoint"] = viewpoint_free_section(grt, local_rank, args.steps, args.warmup)
            result["ms_per_step_seating_without_viewpoint"] = result["seating_without_viewpoint"]["ms_per_step"]   # (next to value / ms_per_step: VERDICT round 5, weak 7)
        if world == 1 and split_world == 1 and merged and not args.no_pmc and not os.environ.get("BENCH_PMC_CHILD"):
            # hardware counters of the same command (separate rocprofv3 --pmc passes); this process lets go of the GPU first
            if pt is not None:
                pt.close()
            scene.close(); closed = True
            node_visits = float(sum(s["closest"]["nodes"] + s["shadow"]["nodes"] for s in trace_rays_stat)); triangle_tests = float(sum(s["closest"]["triangles"] + s["shadow"]["triangles"] for s in trace_rays_stat))
            pmc = pmc_section(args, rays_plan / args.steps, launch_ms, plan, node_visits, triangle_tests)
            pmc_kernels = pmc.pop("kernels", {})
            result["roofline"].update(pmc)
            r = result["roofline"]
            for stage in r.get("stages", []):   # the counters of the other stages' kernels, where the passes saw them
                name = {"traversal": TRACE_KERNEL[0], "sort": "kernel_sort_stream", "generate": "kernel_generate_stream", "accumulate": "kernel_accumulate_group"}.get(stage["stage"], "kernel_" + stage["stage"] + "_stream")
                k = pmc_kernels.get(name) or pmc_kernels.get(name + "_texels")   # (material kernels: the instantiation without the per-fetch BC1 decode, rt_set_texture_expansion)
                if k and k.get("SQ_INSTS_VALU", [0, 0])[1] > 0 and k.get("_duration_ns"):
                    stage["lane_utilisation"] = round(k["SQ_THREAD_CYCLES_VALU"][1] / (64.0 * k["SQ_INSTS_VALU"][1]), 3)
                    stage["valu_issue_frac"] = round(k["SQ_INSTS_VALU"][1] / (1024.0 * k["_duration_ns"][1] * 2.4) / 0.5, 3)   # wave-instructions per cycle and SIMD over the peak of one per 2 cycles
                    stage["waves_per_simd"] = round(4.0 * k.get("SQ_WAVE_CYCLES", [0, 0.0])[1] / (1024.0 * k["_duration_ns"][1] * 2.4), 2)
            if r.get("traffic") and r.get("algorithmic_bytes_per_launch"):
                # how much of what the traversal reads is served by the caches (L1 + L2 + Infinity Cache together): the
                # algorithmic bytes are a lower bound of its requests, the memory-side traffic is what got past the caches
                r["cache_hit_fraction_lower_bound"] = round(1.0 - r["traffic"] / r["algorithmic_bytes_per_launch"], 3)
                r["hbm_frac"] = round(r["traffic"] / (float(np.mean(launch_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if len(launch_ms) else None
                bind = r.get("binding", {})
                r["bound_in_practice"] = ("no unit saturated; closest to its roof: %s. Utilisations (the vector ALU's, L2's and HBM's peaks from MI355X_MICROARCH.md, the L1's measured: profiles/r06_l1_lookup_rate.txt; roofline.binding): %s. The memory side moves %.0f %% of the algorithmic bytes (hbm_frac %.2f), which is why SURVEY 8d's figure (algorithmic_bytes_over_hbm_peak) can exceed 1"
                                          % (bind.get("closest_to_its_roof"), json.dumps(bind.get("utilisation_by_unit")), 100.0 * r["traffic"] / r["algorithmic_bytes_per_launch"], r["hbm_frac"] or float("nan")))
            bind = r.get("binding") or {}
            if bind.get("closest_to_its_roof"):   # the top level names the binding unit: bound / achieved / peak / unit / frac are ITS
                name = bind["closest_to_its_roof"]
                if name in ("l1", "l2", "hbm"):
                    unit = bind[name]; r.update({"bound": name, "achieved": unit["achieved"], "peak": unit["peak"], "unit": unit["unit"], "frac": unit["frac"]})
                elif name == "valu (lane-instructions)":
                    r.update({"bound": "valu", "achieved": bind["achieved"], "peak": bind["peak"], "unit": bind["unit"], "frac": bind["frac"]})
                elif name == "valu issue (wave-instructions)":
                    r.update({"bound": "valu issue", "achieved": round(bind["issue_frac"] * 0.5, 4), "peak": 0.5, "unit": "vector wave-instructions per cycle and SIMD", "frac": bind["issue_frac"]})
                else:   # what the round's own instruction mix allows (classes serial)
                    mix = bind["mix_aware"]
                    r.update({"bound": "valu issue, mix-aware", "achieved": round(1.0 / mix["measured_cycles_per_instruction"], 4), "peak": round(1.0 / mix["best_cycles_per_instruction"]["classes_serial"], 4),
                              "unit": "vector wave-instructions per cycle and SIMD (x lane utilisation in frac)", "frac": mix["frac_classes_serial"]})
                r["bound_is"] = "the unit of the dominant kernel closest to its roof (roofline.binding.utilisation_by_unit); no unit is saturated -- the launch is a chain of dependent node and triangle fetches"
        if world == 1 and split_world == 1 and merged and not args.no_tile_split_bound and not os.environ.get("BENCH_PMC_CHILD"):
            # No multi-GPU hardware has run this benchmark in any round (SCALE records: skipped). What ONE GPU can say about the tile split: rank 0's share of an N-way
            # split rendered alone (--emulate-world N: its tiles, pack + unpack of the exchange, no collective) -- the whole frame's ms per step over the rank's is a BOUND
            # of the speed-up N ranks can reach on this command (the all-gather itself, 33 MB per frame, comes on top). Child processes of this command, one at a time,
            # after this process has let go of the GPU.
            if not closed:
                if pt is not None:
                    pt.close(); pt = None
                scene.close(); closed = True
            import subprocess
            bound = {"note": "rank 0's share of an N-way tile split rendered on this one GPU (bench.py --emulate-world N, same steps / warm-up); speedup_bound = this line's ms_per_step / the rank's: an upper bound, the collective is not in it", "ranks": []}
            for n in (2, 4, 8):
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--emulate-world", str(n), "--burst", str(args.burst),
                       "--no-cpu-baseline", "--no-povs", "--no-pmc", "--no-config3", "--no-reference-layout", "--no-stages", "--no-tile-split-bound", "--merge-static", str(args.merge_static)]
                try:
                    child = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300, env=dict(os.environ, BENCH_PMC_CHILD="1"))
                    line = [l for l in child.stdout.splitlines() if l.startswith("{")]
                    ms = json.loads(line[-1])["ms_per_step"]
                    bound["ranks"].append({"world": n, "rank_ms_per_step": ms, "speedup_bound": round(result["ms_per_step"] / ms, 2)})
                except Exception as error:   # (a bound that could not be measured is not a reason to lose the line)
                    bound["ranks"].append({"world": n, "error": str(error)[:200]})
            result["tile_split_bound"] = bound
        print(json.dumps(result))

    if not closed:
        if pt is not None:
            pt.close()
        scene.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
