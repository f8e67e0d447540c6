#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel, read off the compiler's assembly.

    python tools/isa_loop_mix.py [--kernel NAME] [--unit kernels_trace] [--flags "-DX=1"] [--min-valu 100] [--listing]

Compiles gpu-raytracer_amd/csrc/<unit>.hip for gfx950 with the Makefile's flags (device code only, -S), finds the kernel's
function body and prints, per loop the compiler annotated (`; =>This Inner Loop Header` / `in Loop: Header=...`), the number of
vector / scalar / branch / wait / memory / LDS instructions and how many vector instructions belong to the class that takes
~4.1 cycles per wave on MI355X (conversions, min / max, compares, selects, left shifts, bit-field, integer multiply, division
helpers; profiles/r04_instruction_costs.txt) -- the rest retire in ~2.3. GPU minutes are scarce: this prices a change of the
traversal round on the CPU. bench.py's roofline.binding.mix_aware takes its two constants from the loop of the closest-hit engine
of kernel_trace_stream_bvh8_flat (the loop with the most vector instructions that loads nodes: five global_load_dwordx4)."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ~2.3 cycles per wave-instruction (profiles/r04_instruction_costs.txt); everything else vector is counted as the 4-cycle class
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_mov_b64", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
        "v_bitop3_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_nop"}
TRANSCENDENTAL = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if base.startswith("v_"):
        if base in FAST:
            return "valu_fast"
        if base in TRANSCENDENTAL:
            return "valu_transcendental"
        return "valu_slow"
    if base.startswith("s_waitcnt"):
        return "waitcnt"
    if base.startswith("s_cbranch") or base == "s_branch":
        return "branch"
    if base.startswith("s_load") or base.startswith("s_buffer_load"):
        return "smem"
    if base.startswith("s_"):
        return "salu"
    if base.startswith("global_") or base.startswith("buffer_") or base.startswith("flat_") or base.startswith("scratch_"):
        return "vmem"
    if base.startswith("ds_"):
        return "lds"
    return "other"


def compile_unit(unit, flags):
    out = "/tmp/isa_loop_mix_%s.s" % unit
    per_file = ["-fno-slp-vectorize"] if unit in ("kernels_shade", "kernels_post") else []   # the Makefile's HIPFLAGS_<unit>
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
           "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "gpu-raytracer_amd", "csrc", unit + ".hip")] + per_file + flags.split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def kernel_body(lines, kernel):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_?Z?\w*%s\w*:" % re.escape(kernel), l) and not l.startswith("."))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loops(body):
    """{header label: [instruction lines]}: a block belongs to the innermost loop its comment names."""
    current, out = None, {}
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label = m.group(1)
            rest = l[m.end():]
            if "Loop Header" in rest:
                current = label.lstrip(".L")
            else:
                m2 = re.search(r"in Loop: Header=(BB\d+_\d+)", rest)
                current = m2.group(1) if m2 else None
            continue
        m = re.match(r"^; %bb\.\d+:\s*(;.*)?$", l)
        if m:
            m2 = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            current = m2.group(1) if m2 else None
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if current:
            out.setdefault(current, []).append(s)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="kernel_trace_stream_bvh8_flat")
    ap.add_argument("--unit", default="kernels_trace")
    ap.add_argument("--flags", default="")
    ap.add_argument("--min-valu", type=int, default=100)
    ap.add_argument("--listing", action="store_true", help="print the instructions of the loops shown")
    ap.add_argument("--json", default="", help="write {valu, valu_slow, ...} of the FIRST loop shown (the closest-hit engine's round) to this file: bench.py reads profiles/trace_round_mix.json")
    a = ap.parse_args()
    path = compile_unit(a.unit, a.flags)
    lines = open(path).read().split("\n")
    body = kernel_body(lines, a.kernel)
    meta = [l.strip() for l in lines if re.search(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):", l) or re.search(r"\.name:\s+\S*%s\S*$" % re.escape(a.kernel), l)]
    print("%s (%s.hip %s): %d lines of assembly" % (a.kernel, a.unit, a.flags or "default flags", len(body)))
    # the metadata block of this kernel: the four figures in front of its .name line
    for i, l in enumerate(lines):
        if re.search(r"\.name:\s+\S*%s\s*$" % re.escape(a.kernel), l) or re.search(r"\.name:\s+_Z\d+%s\w*$" % re.escape(a.kernel), l):
            block = lines[max(0, i - 40):i + 30]
            print("  " + ", ".join(x.strip() for x in block if re.search(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count):", x)))
            break
    first = None
    for header, instrs in loops(body).items():
        counts = {}
        for s in instrs:
            c = classify(s.split()[0])
            counts[c] = counts.get(c, 0) + 1
        valu = counts.get("valu_fast", 0) + counts.get("valu_slow", 0) + counts.get("valu_transcendental", 0)
        if valu < a.min_valu:
            continue
        node_loads = sum(1 for s in instrs if s.startswith("global_load_dwordx4"))
        print("  loop %-10s valu %4d (slow class %4d, transcendental %2d) salu %3d branch %3d waitcnt %3d vmem %3d (dwordx4 loads %2d) lds %3d smem %d"
              % (header, valu, counts.get("valu_slow", 0) + counts.get("valu_transcendental", 0), counts.get("valu_transcendental", 0), counts.get("salu", 0), counts.get("branch", 0),
                 counts.get("waitcnt", 0), counts.get("vmem", 0), node_loads, counts.get("lds", 0), counts.get("smem", 0)))
        if first is None:
            first = {"kernel": a.kernel, "loop": header, "valu": valu, "valu_slow": counts.get("valu_slow", 0) + counts.get("valu_transcendental", 0), "salu": counts.get("salu", 0), "branch": counts.get("branch", 0),
                     "vmem": counts.get("vmem", 0), "dwordx4_loads": node_loads, "lds": counts.get("lds", 0), "flags": a.flags}
        if a.listing:
            for s in instrs:
                print("      " + s)
    if a.json and first:
        import json
        json.dump(first, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
