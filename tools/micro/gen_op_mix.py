#!/usr/bin/env python3
"""Generates tools/micro/op_mix.hip: issue cost of MIXED instruction streams on gfx950 at 8 waves per SIMD.

op_issue.hip showed that costs of single-opcode streams are not additive in mixed streams; this measures patterns
(fast : slow ratios, pairs of different slow classes, SALU / LDS / exec-write interleaving) so that the walk's
per-iteration cost can be modelled (bench.py roofline.issue.floor_ms) and its instruction selection decided.
Every pattern is a list of ops; op j runs on chain j % 4 (registers v[100+4c .. 103+4c]) and the list is repeated to 32
instructions per loop trip.  Data toggles (lane-dependent start values, results fed back) so that the power governor
sees a busy chip."""
import sys

OPS = {
    # fast class (2.7)
    "add":   "v_add_f32 {d}, {a}, {d}",
    "xor":   "v_xor_b32 {d}, {a}, {d}",
    "mul":   "v_mul_f32 {d}, {a}, {d}",
    "fma":   "v_fma_f32 {d}, {a}, {b}, {d}",
    "mov":   "v_mov_b32 {d}, {a}",
    "subu":  "v_sub_u32 {d}, {a}, {d}",
    "lshrc": "v_lshrrev_b32 {d}, 3, {a}",
    "lshlc": "v_lshlrev_b32 {d}, 3, {a}",
    "ashrc": "v_ashrrev_i32 {d}, 31, {a}",
    "and":   "v_and_b32 {d}, {a}, {d}",
    "or":    "v_or_b32 {d}, {a}, {d}",
    "andl":  "v_and_b32 {d}, 0x10100, {a}",
    # slow class
    "lshlv": "v_lshlrev_b32 {d}, {a}, {d}",
    "lshrv": "v_lshrrev_b32 {d}, {a}, {d}",
    "minf":  "v_min_f32 {d}, {a}, {d}",
    "maxf":  "v_max_f32 {d}, {a}, {d}",
    "minu":  "v_min_u32 {d}, {a}, {d}",
    "min3":  "v_min3_f32 {d}, {a}, {b}, {d}",
    "lor":   "v_lshl_or_b32 {d}, {a}, {b}, {d}",
    "lorc":  "v_lshl_or_b32 {d}, {a}, 1, {d}",
    "ladd":  "v_lshl_add_u32 {d}, {a}, 11, {d}",
    "add3":  "v_add3_u32 {d}, {a}, {b}, {d}",
    "bfe":   "v_bfe_u32 {d}, {a}, {b}, 1",
    "ffbh":  "v_ffbh_u32 {d}, {a}",
    "pkmul": "v_pk_mul_f32 v[{p0}:{p1}], v[{p2}:{p3}], v[{p0}:{p1}]",
    "pkadd": "v_pk_add_f32 v[{p0}:{p1}], v[{p2}:{p3}], v[{p0}:{p1}]",
    "cmp":   "v_cmp_lt_f32_e32 vcc, {a}, {d}",
    "cmps":  "v_cmp_lt_f32_e64 s[20:21], {a}, {d}",
    "cmpu":  "v_cmp_ne_u32_e32 vcc, 0, {d}",
    "cnd32": "v_cndmask_b32_e32 {d}, {a}, {d}, vcc",
    "cnd0":  "v_cndmask_b32_e32 {d}, 0, {a}, vcc",
    "cnd64": "v_cndmask_b32_e64 {d}, {a}, {d}, s[20:21]",
    "addc":  "v_addc_co_u32_e32 {d}, vcc, {a}, {d}, vcc",
    "addcs": "v_addc_co_u32_e64 {d}, s[22:23], {a}, {d}, s[20:21]",
    "sgpr":  "v_and_b32 {d}, s24, {d}",
    "rcp":   "v_rcp_f32 {d}, {a}",
    "cvt":   "v_cvt_f32_u32 {d}, {a}",
    # scalar / memory / control
    "sand":  "s_and_b64 s[22:23], s[20:21], vcc",
    "sandn": "s_andn2_b64 s[22:23], s[22:23], s[20:21]",
    "sadd":  "s_add_u32 s25, s25, 1",
    "sexec": "s_and_b64 exec, exec, s[26:27]",
    "smovx": "s_mov_b64 exec, s[26:27]",
    "dsw":   "ds_write_b64 v120, v[{p0}:{p1}]",
    "dsr":   "ds_read_b64 v[116:117], v120",
    "swl":   "s_waitcnt lgkmcnt(0)",
    "gld":   "global_load_dwordx2 v[118:119], v121, s[28:29]",
    "swv":   "s_waitcnt vmcnt(0)",
    "bcnt":  "s_bcnt1_i32_b64 s25, exec",
    "nop":   "s_nop 0",
}

FASTSET = {"add", "xor", "mul", "fma", "mov", "subu", "lshrc", "lshlc", "ashrc", "and", "or", "andl"}

PATTERNS = [
    ("F add", ["add"]), ("F xor", ["xor"]), ("F fma", ["fma"]), ("F lshr const", ["lshrc"]), ("F ashr const", ["ashrc"]),
    ("F and literal", ["andl"]), ("F mov", ["mov"]), ("F or", ["or"]),
    ("S lshl_or", ["lor"]), ("S lshl_or const", ["lorc"]), ("S min_f32", ["minf"]), ("S lshl var", ["lshlv"]), ("S lshr var", ["lshrv"]),
    ("S cmp vcc", ["cmp"]), ("S cmp sgpr", ["cmps"]), ("S pk_mul", ["pkmul"]), ("S min3", ["min3"]), ("S ffbh", ["ffbh"]),
    ("S sgpr operand", ["sgpr"]), ("S bfe", ["bfe"]), ("T rcp", ["rcp"]),
    ("cmp + cnd32 (1:1)", ["cmp", "cnd32"]), ("cmp + 3 cnd32", ["cmp", "cnd32", "cnd32", "cnd32"]), ("cmp + 3 cnd0", ["cmp", "cnd0", "cnd0", "cnd0"]),
    ("cmps + 3 cnd64", ["cmps", "cnd64", "cnd64", "cnd64"]), ("cmp + addc", ["cmp", "addc"]), ("cmps + addcs", ["cmps", "addcs"]),
    ("S F  lor:add 1:1", ["lor", "add"]), ("S F F  lor:add 1:2", ["lor", "add", "add"]), ("S F F F  lor:add 1:3", ["lor", "add", "add", "add"]),
    ("S S F F", ["lor", "lor", "add", "add"]), ("S S S F", ["lor", "lor", "lor", "add"]),
    ("S F  cmp:add 1:1", ["cmp", "add"]), ("S F F F cmp:add 1:3", ["cmp", "add", "add", "add"]),
    ("S F  minf:xor 1:1", ["minf", "xor"]), ("S F  pkmul:add 1:1", ["pkmul", "add"]),
    ("S S' lor + cmp", ["lor", "cmp"]), ("S S' lor + minf", ["lor", "minf"]), ("S S' pkmul + cmp", ["pkmul", "cmp"]), ("S S' minf + cmp", ["minf", "cmp"]),
    ("S S' lor + pkmul", ["lor", "pkmul"]), ("S S' lshlv + cmp", ["lshlv", "cmp"]),
    ("T F rcp + add", ["rcp", "add"]), ("T S rcp + lor", ["rcp", "lor"]), ("T F F F rcp + 3 add", ["rcp", "add", "add", "add"]),
    ("F + salu 1:1 add, s_add", ["add", "sadd"]), ("F + 2 salu", ["add", "sadd", "sadd"]), ("S + salu lor, s_add", ["lor", "sadd"]),
    ("cmp + s_and + add", ["cmp", "sand", "add"]), ("cmp + s_and + s_andn2 + add", ["cmp", "sand", "sandn", "add"]),
    ("add + exec and (1:1)", ["add", "sexec"]), ("add add add + exec mov", ["add", "add", "add", "smovx"]),
    ("add x3 + ds_write_b64", ["add", "add", "add", "dsw"]), ("add x3 + ds_read_b64", ["add", "add", "add", "dsr"]),
    ("add x7 + ds_read_b64 + wait", ["add", "add", "add", "dsr", "add", "add", "add", "swl"]),
    ("add x7 + global_load + wait", ["add", "add", "add", "gld", "add", "add", "add", "swv"]),
    ("add x3 + global_load", ["add", "add", "add", "gld"]),
    ("add + s_nop (1:1)", ["add", "nop"]), ("salu only s_add", ["sadd"]), ("s_bcnt + s_add + add", ["bcnt", "sadd", "add"]),
    ("F F lshrc + subu", ["lshrc", "subu"]), ("F F and + or", ["and", "or"]),
]


def emit(ops):
    lines = []
    n = 0
    while n < 32:
        for op in ops:
            c = n % 4
            base = 100 + 4 * c
            lines.append(OPS[op].format(d=f"v{base}", a=f"v{base + 1}", b=f"v{base + 2}",
                                        p0=base, p1=base + 1, p2=base + 2, p3=base + 3))
            n += 1
            if n >= 32 and len(ops) <= 4:
                break
        if len(ops) > 4 and n >= 32:
            break
    return lines, n


def main(path):
    out = []
    out.append("// GENERATED by tools/micro/gen_op_mix.py -- do not edit.  See that file for what this measures.\n")
    out.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n")
    out.append('#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115",'
               '"v116","v117","v118","v119","v120","v121","vcc","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","memory"\n')
    out.append("__shared__ unsigned long long lds_buf[1024];\n")
    out.append("template <int MODE> __global__ void __launch_bounds__(1024) k(float* out, const float* in, int iters, unsigned long long* clk)\n{\n")
    out.append("    const float f0 = 1.0f + 0.001f * threadIdx.x, f1 = 0.999f + 1e-6f * threadIdx.x;\n")
    out.append("    lds_buf[threadIdx.x] = threadIdx.x;\n")
    out.append('    asm volatile("s_mov_b64 s[28:29], %2\\n v_mov_b32 v121, 0\\n v_lshlrev_b32 v120, 3, %3\\n"\n')
    for r in range(100, 120):
        src = "%0" if r % 2 == 0 else "%1"
        out.append(f'                 "v_add_f32 v{r}, {src}, {src}\\n"\n')
    out.append('                 "s_mov_b64 s[20:21], exec\\n s_mov_b64 s[22:23], exec\\n s_mov_b64 s[26:27], exec\\n s_mov_b32 s24, 3\\n s_mov_b32 s25, 0\\n s_mov_b64 vcc, exec"\n')
    out.append('                 :: "v"(f0), "v"(f1), "s"(in), "v"(threadIdx.x) : CLOB);\n')
    out.append("    unsigned long long t0 = 0, r0 = 0, t1 = 0, r1 = 0;\n")
    out.append('    asm volatile("s_memtime %0\\n s_memrealtime %1\\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");\n')
    out.append("    for (int i = 0; i < iters; ++i) {\n")
    counts = []
    for idx, (name, ops) in enumerate(PATTERNS):
        lines, n = emit(ops)
        counts.append(n)
        kw = "if" if idx == 0 else "else if"
        body = "\\n ".join(lines)
        out.append(f'        {kw} (MODE == {idx}) {{ asm volatile("{body}" ::: CLOB); }}\n')
    out.append("    }\n")
    out.append('    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n s_memtime %0\\n s_memrealtime %1\\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");\n')
    out.append('    float r; asm volatile("v_add_f32 %0, v100, v104\\n v_add_f32 %0, %0, v108\\n v_add_f32 %0, %0, v112\\n v_add_f32 %0, %0, v116\\n v_add_f32 %0, %0, v118" : "=v"(r) :: CLOB);\n')
    out.append("    out[blockIdx.x * blockDim.x + threadIdx.x] = r;\n")
    out.append("    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }\n}\n\n")
    out.append("static double g_base = 0.0;\n")
    out.append("template <int MODE> void run(const char* name, int ninstr, int nvalu)\n{\n")
    out.append("    static float *out = nullptr, *in = nullptr; static unsigned long long* clk = nullptr;\n")
    out.append("    if (!out) { (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&in, 1 << 20); (void)hipMemset(in, 0, 1 << 20); (void)hipMalloc(&clk, 16); }\n")
    out.append("    const int iters = 4000, waves_per_simd = 8;\n")
    out.append("    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);\n")
    out.append("    k<MODE><<<512, 1024>>>(out, in, 10, clk);\n    double best = 1e30; unsigned long long h[2] = {0, 0};\n")
    out.append("    for (int rep = 0; rep < 3; ++rep) {\n        (void)hipEventRecord(e0); k<MODE><<<512, 1024>>>(out, in, iters, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);\n")
    out.append("        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) { best = ms; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }\n    }\n")
    out.append("    const double ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;   // s_memrealtime counts at 100 MHz\n")
    out.append("    const double per_trip = best * 1e6 / ((double)iters * waves_per_simd) * 2.4;   // nominal cycles of SIMD issue per loop trip\n")
    out.append("    if (MODE == 0) g_base = per_trip / ninstr;\n")
    out.append('    printf("%-34s %2d instr (%2d VALU)  %7.1f cyc/trip  %5.2f per instr  %5.2f per VALU  x%.2f of add   memtime clock %.2f GHz\\n", name, ninstr, nvalu, per_trip,\n')
    out.append("           per_trip / ninstr, nvalu ? per_trip / nvalu : 0.0, nvalu ? per_trip / nvalu / g_base : 0.0, ghz);\n")
    out.append("    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);\n}\n\n")
    out.append("int main()\n{\n    setvbuf(stdout, nullptr, _IONBF, 0);\n")
    for idx, (name, ops) in enumerate(PATTERNS):
        lines, n = emit(ops)
        nv = sum(1 for l in lines if l.startswith("v_"))
        out.append(f'    run<{idx}>("{name}", {n}, {nv});\n')
    out.append("    return 0;\n}\n")
    with open(path, "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tools/micro/op_mix.hip")
