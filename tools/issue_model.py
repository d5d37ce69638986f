#!/usr/bin/env python3
"""Static instruction-issue model of the walk loop (csrc/vrc_walk.h) x measured block frequencies -> the issue floor that
bench.py reports as roofline.issue.floor_ms.

    python tools/issue_model.py [--freq profiles/r03/walk_block_frequencies_c3.json] [--out profiles/issue_model.json]

1. The asm text of the loop is taken from vrc_walk.h itself (the VRC_WALK_ASM macro with the variant parts substituted), split
   into its four blocks (header / descend / advance / pop) and every instruction is put in an issue class
   (tools/micro/op_issue.hip, gen_op_mix.py; profiles/r03/op_issue.txt, op_mix.txt, op_pairs.txt):
     A  v_add / sub / mul / fma_f32, v_mov        simple; co-issues beside anything but a class-P instruction
     B  v_xor / and / or, v_add / sub_u32, v_lshrrev, v_ashrrev      simple; co-issues beside A, B, S
     S  v_cndmask, v_addc (SGPR operand)          simple; co-issues beside A, B
     Q  compares, v_min / max / min3, v_lshlrev, v_lshl_or / lshl_add / add3 / or3 / and_or / bfe, v_ffbh, v_cvt ...
                                                  slow class: ~4.3 cycles of its unit each, only an A beside it
     P  v_pk_*_f32, v_mul_u32_u24, v_mad_*24, v_readlane / writelane      nothing beside it (none in the loop)
2. tests/tools/path_schedule.py --dump gives how often a wave issues each block per stage (from the oracle's op traces of the
   C3 frame; a data file, bench.py never touches the oracle for it).
3. Per tile-sample: VALU / SALU / slow-class instructions of the walks = sum over stages and blocks.

The floor (bench.py): a SIMD cannot issue faster than one VALU wave-instruction per 2 cycles (MI355X_MICROARCH.md: a wave64
instruction takes two passes of the SIMD-32) nor retire slow-class instructions faster than one per ~4.3 cycles, and a CU's one
scalar unit serves its four SIMDs at ~1 instruction per cycle:
    floor = max(2.0 * VALU per SIMD, 4.3 * slow + 4.7 * P per SIMD, 1.1 * (SALU + branches) per CU)   -- floor_cycles() below.
The fastest any instruction stream of plain VALU instructions issued in the microbenchmarks is one per 2.56 cycles
(`measured_ceiling`), which bench.py reports beside the architectural figure."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WALK = os.path.join(ROOT, "cpuvoxelraycaster_amd", "csrc", "vrc_walk.h")

CLASS_A = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32"}
CLASS_B = {"v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32"}
CLASS_S = {"v_cndmask_b32", "v_addc_co_u32", "v_subb_co_u32"}
CLASS_P = {"v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mov_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_i32_i24",
           "v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_mul_lo_u32", "v_mul_hi_u32"}
CYC_VALU_ARCH = 2.0        # MI355X_MICROARCH.md: wave64 VALU instruction = 2 passes of the SIMD-32
CYC_VALU_MEASURED = 2.56   # fastest plain-VALU stream, 8 waves per SIMD, at 2.39-2.40 GHz (profiles/r03/op_pairs.txt: "F add:mov")
CYC_SLOW = 4.3             # one slow-class instruction (profiles/r03/op_mix.txt: v_lshl_or_b32 alone 4.28-4.32)
CYC_P = 4.7
CYC_SALU_PER_CU = 1.1      # profiles/r03/op_mix.txt "salu only": 4.45 per SIMD stream = 1.1 cycles per instruction of the CU's scalar unit


def opcode(line):
    m = re.match(r"\s*([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|buffer_[a-z0-9_]+|global_[a-z0-9_]+)", line)
    if not m:
        return None
    op = m.group(1)
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_")):
        return "vmem"
    if op in CLASS_A:
        return "A"
    if op in CLASS_B:
        return "B"
    if op in CLASS_S:
        return "S"
    if op in CLASS_P:
        return "P"
    return "Q"


def preprocess(src, defined=()):
    """the lines the compiler sees for the product's flags: #ifdef / #ifndef / #else / #endif resolved against `defined` (the
    product build defines none of the A/B switches), so that a macro with two variants is read in the shipped one whatever
    their order in the file (round-5 advice)"""
    out, stack = [], []                # stack of (taking this branch, any enclosing branch skipped)
    for line in src.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|else|endif)\b\s*(\w+)?", line)
        if m and m.group(1) in ("ifdef", "ifndef"):
            take = (m.group(2) in defined) == (m.group(1) == "ifdef")
            stack.append(take)
        elif m and m.group(1) == "else":
            stack[-1] = not stack[-1]
        elif m and m.group(1) == "endif":
            stack.pop()
        elif all(stack):
            out.append(line)
    assert not stack, "unbalanced #if in the walk header"
    return "\n".join(out)


def macro_body(src, name):
    """text of a (possibly function-like) #define, continuation lines joined"""
    m = re.search(r"^#define\s+" + re.escape(name) + r"(\([^)]*\))?[ \t]+(.*?)(?<!\\)\n", src, re.S | re.M)
    if not m:
        raise KeyError(name)
    return m.group(2)


def strings_of(text):
    """the asm text a macro body contributes: its string literals, in order, with nested macro names left in place"""
    out = []
    for tok in re.finditer(r'"((?:[^"\\]|\\.)*)"|\b(VRC_WALK_[A-Z_0-9]+|SPANPART|STARTPART|COUNTPART|BIASPART|ROWSH|ROWDOWN)\b|/\*(.*?)\*/', text, re.S):
        if tok.group(1) is not None:
            out.append(("str", tok.group(1).replace("\\n", "\n")))
        elif tok.group(2) is not None:
            out.append(("macro", tok.group(2)))
        else:
            out.append(("comment", tok.group(3)))
    return out


def expand(src, lod_mode, count_lane):
    sub = {"STARTPART": "VRC_WALK_START_BELOW","SPANPART": "VRC_WALK_SPAN_NOLOD" if lod_mode == 0 else "VRC_WALK_SPAN_LOD",
           "TMAXPART": "VRC_WALK_TMAX_NOLOD" if lod_mode == 0 else "VRC_WALK_TMAX_LOD",
           "COUNTPART": "VRC_WALK_COUNT_LANE" if count_lane else "VRC_WALK_COUNT_WAVE",
           "BIASPART": "VRC_WALK_BIAS_ADD" if lod_mode == 2 else "VRC_WALK_BIAS_NONE"}

    def rec(text):
        res = []
        for kind, val in strings_of(text):
            if kind == "str":
                res.append(val)
            elif kind == "comment":
                if "----" in val:
                    res.append("\n;BLOCK " + val.strip(" -").split()[0] + "\n")
            else:
                name = sub.get(val, val)
                if name in ("ROWSH", "ROWDOWN"):          # macro arguments that are string literals (the stack's row stride)
                    res.append("11" if name == "ROWSH" else "0xfffff800")
                else:
                    res.extend(rec(macro_body(src, name)))
        return res
    return "".join(rec(macro_body(src, "VRC_WALK_ASM")))


def block_counts(lod_mode=0, count_lane=False):
    src = preprocess(open(WALK).read())
    text = expand(src, lod_mode, count_lane)
    blocks = {"prologue": {}, "header": {}, "descend": {}, "advance": {}, "pop": {}, "start": {}, "epilogue": {}}
    in_start = False                  # the second half of the pop block is also the entry of rays that start below the root
    cur = "prologue"
    for line in text.split("\n"):
        line = line.strip()
        if line.startswith(";BLOCK"):
            cur = line.split()[1]
            continue
        if re.match(r"Lwalk_loop", line):
            cur = "header"
            continue
        if re.match(r"Lwalk_next", line):
            cur = "header"            # the two loop-closing scalar instructions run every round
            in_start = False
            continue
        if re.match(r"Lwalk_start", line):
            in_start = True
            continue
        if re.match(r"Lwalk_done", line):
            cur = "epilogue"
            continue
        if re.match(r"Lwalk_adv", line):
            cur = "advance"
            continue
        op = opcode(line)
        if op is None:
            continue
        c = classify(op)
        blocks[cur][c] = blocks[cur].get(c, 0) + 1
        if in_start:
            blocks["start"][c] = blocks["start"].get(c, 0) + 1
    return blocks


def per_tile_sample(freq, counts_nolod, counts_lod):
    """instructions of the four walks per tile-sample, by class"""
    tot = {}
    rounds = 0.0
    for stage, f in freq["stages"].items():
        counts = counts_lod if stage.startswith("GI") else counts_nolod   # the GI ray and its shadow ray are cast with a LOD coefficient
        for blk in ("header", "descend", "advance", "pop", "start"):
            n = f.get(blk + "s" if blk == "start" else blk, 0) / freq["tiles"]
            for c, k in counts[blk].items():
                tot[c] = tot.get(c, 0.0) + n * k
        rounds += f["header"] / freq["tiles"]
    tot["rounds"] = rounds
    return tot


def floor_cycles(valu, slow, p, scalar):
    """Cycles (of one SIMD / of one CU's scalar unit) below which the instruction streams cannot be issued, per resource; counts are
    wave-instructions per SIMD (VALU, slow, P) and per CU (scalar = SALU + branch instructions of the CU's four SIMDs)."""
    return {"valu_arch": CYC_VALU_ARCH * valu, "valu_measured_ceiling": CYC_VALU_MEASURED * valu,
            "slow_unit": CYC_SLOW * slow + CYC_P * p, "scalar_unit": CYC_SALU_PER_CU * scalar}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--freq", default=os.path.join(ROOT, "profiles", "r05", "walk_block_frequencies_c3_quad.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "issue_model.json"))
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    nolod, lod = block_counts(0, False), block_counts(1, False)
    freq = json.load(open(a.freq))
    pts = per_tile_sample(freq, nolod, lod)
    doc = {"_comment": "tools/issue_model.py: per-block instruction classes of the walk loop (parsed from csrc/vrc_walk.h) x block issue "
                       "frequencies (tests/tools/path_schedule.py --dump); bench.py uses it only for the kernel sources it was made from",
           "source_hash": bench.kernel_source_hash(), "frequencies": os.path.relpath(a.freq, ROOT), "workload": freq["workload"],
           "blocks_no_lod": nolod, "blocks_lod": lod,
           "walks_per_tile_sample": {k: round(v, 2) for k, v in pts.items()},
           # live lanes per issued loop header, stage by stage (a header is issued when any of the wave's 64 lanes iterates): what
           # roofline.issue.lane_utilisation averages over.  In the quadrant layout the primary and shadow stages' lanes are one
           # pixel's `spp` identical walks side by side (raycaster.hpp:131,153 cast them once per sample), so their high figures
           # are three-quarters copies at 4 spp; the GI stages' rays are all distinct
           "lane_utilisation_by_stage": {{"primary": "primary", "shadow": "shadow", "GI": "gi", "GI shadow": "gi_shadow"}[k]:
                                         round(v["lane_iterations"] / (64.0 * v["header"]), 4) for k, v in freq["stages"].items()},
           "cycles": {"valu_arch": CYC_VALU_ARCH, "valu_measured_ceiling": CYC_VALU_MEASURED, "slow": CYC_SLOW, "p": CYC_P,
                      "salu_per_cu": CYC_SALU_PER_CU}}
    json.dump(doc, open(a.out, "w"), indent=1)
    for name, b in (("no LOD", nolod), ("LOD", lod)):
        print(name, {k: v for k, v in b.items() if k not in ("prologue", "epilogue")})
    v = sum(pts.get(c, 0) for c in "ABSQP")
    print(f"per tile-sample: rounds {pts['rounds']:.1f}, VALU {v:.0f} (A {pts.get('A', 0):.0f} B {pts.get('B', 0):.0f} S {pts.get('S', 0):.0f} "
          f"Q {pts.get('Q', 0):.0f} P {pts.get('P', 0):.0f}), SALU {pts.get('salu', 0):.0f} + branch {pts.get('branch', 0):.0f}, LDS {pts.get('lds', 0):.0f}, VMEM {pts.get('vmem', 0):.0f}")


if __name__ == "__main__":
    main()
