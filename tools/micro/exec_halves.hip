// Does gfx950 skip a pass of the SIMD-32 when one half (or three quarters) of a wave64's EXEC mask is zero?  (Round 5: the frame
// kernel's GI stages run with 44-52 % of their lanes live; if an instruction whose upper 32 lanes are all dark cost one pass instead
// of two, packing the live lanes of a stage into one half of the wave would be worth their exchange.)  Same harness as op_issue.hip:
// 32 independent instructions per loop trip on named registers, 8 waves per SIMD, cycles of SIMD issue per wave64 instruction at a
// nominal 2.4 GHz -- for different EXEC masks, set once before the loop.
// hipcc --offload-arch=gfx950 -O3 tools/micro/exec_halves.hip -o /tmp/exec_halves && /tmp/exec_halves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
             "v114", "v115", "vcc", "s20", "s21", "s22", "s23"
#define REP8(x) x x x x x x x x

template <int OP>
__global__ void k(float* out, int iters, uint64_t mask)
{
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 2.0\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 4.0\n"
                 "v_mov_b32 v104, 1.0\n v_mov_b32 v105, 2.0\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 4.0\n"
                 "v_mov_b32 v108, 1.0\n v_mov_b32 v109, 2.0\n v_mov_b32 v110, 0.5\n v_mov_b32 v111, 4.0\n"
                 "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 2.0\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 4.0\n" ::: CLOB);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)mask), hi = __builtin_amdgcn_readfirstlane((uint32_t)(mask >> 32));
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 s22, %0\n s_mov_b32 s23, %1\n s_mov_b64 exec, s[22:23]" ::"s"(lo), "s"(hi) : CLOB);
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 v100, v101, v102\n v_add_f32 v104, v105, v106\n v_add_f32 v108, v109, v110\n v_add_f32 v112, v113, v114" ::: CLOB);) }
        else if (OP == 1) { REP8(asm volatile("v_min_f32 v100, v101, v102\n v_min_f32 v104, v105, v106\n v_min_f32 v108, v109, v110\n v_min_f32 v112, v113, v114" ::: CLOB);) }
        else if (OP == 2) { REP8(asm volatile("v_fma_f32 v100, v101, v102, v103\n v_xor_b32 v104, v105, v106\n v_cmp_lt_f32 vcc, v109, v110\n v_cndmask_b32 v112, v113, v114, vcc" ::: CLOB);) }
        else if (OP == 4) { REP8(asm volatile("v_cmp_lt_f32 vcc, v101, v102\n v_cmp_lt_f32 vcc, v105, v106\n v_cmp_lt_f32 vcc, v109, v110\n v_cmp_lt_f32 vcc, v113, v114" ::: CLOB);) }
        else if (OP == 5) { REP8(asm volatile("v_xor_b32 v100, v101, v102\n v_xor_b32 v104, v105, v106\n v_xor_b32 v108, v109, v110\n v_xor_b32 v112, v113, v114" ::: CLOB);) }
        else if (OP == 6) { REP8(asm volatile("v_cndmask_b32 v100, v101, v102, vcc\n v_cndmask_b32 v104, v105, v106, vcc\n v_cndmask_b32 v108, v109, v110, vcc\n v_cndmask_b32 v112, v113, v114, vcc" ::: CLOB);) }
        else if (OP == 3) { REP8(asm volatile("v_lshl_or_b32 v100, v101, 1, v102\n v_lshl_or_b32 v104, v105, 1, v106\n v_lshl_or_b32 v108, v109, 1, v110\n v_lshl_or_b32 v112, v113, 1, v114" ::: CLOB);) }
    }
    asm volatile("s_mov_b64 exec, s[20:21]" ::: CLOB);
    float r;
    asm volatile("v_add_f32 %0, v100, v104\n v_add_f32 %0, %0, v108\n v_add_f32 %0, %0, v112" : "=v"(r)::CLOB);
    if (iters < 0) out[threadIdx.x] = r;
}

// Is the few-lanes cost time of the SIMD's VALU (other waves wait) or only of the wave itself?  Waves 0-3, 8-11 of a block
// (one per SIMD each) run with mask A, waves 4-7, 12-15 with mask B; mask 0 = the wave leaves at once.
template <int OP>
__global__ void kmix(float* out, int iters, uint64_t mask_a, uint64_t mask_b)
{
    const uint64_t mask = ((threadIdx.x >> 8) & 1) ? mask_b : mask_a;
    if (__builtin_amdgcn_readfirstlane((uint32_t)(mask | (mask >> 32))) == 0) return;
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 2.0\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 4.0\n"
                 "v_mov_b32 v104, 1.0\n v_mov_b32 v105, 2.0\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 4.0\n"
                 "v_mov_b32 v108, 1.0\n v_mov_b32 v109, 2.0\n v_mov_b32 v110, 0.5\n v_mov_b32 v111, 4.0\n"
                 "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 2.0\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 4.0\n" ::: CLOB);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)mask), hi = __builtin_amdgcn_readfirstlane((uint32_t)(mask >> 32));
    asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 s22, %0\n s_mov_b32 s23, %1\n s_mov_b64 exec, s[22:23]" ::"s"(lo), "s"(hi) : CLOB);
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_add_f32 v100, v101, v102\n v_add_f32 v104, v105, v106\n v_add_f32 v108, v109, v110\n v_add_f32 v112, v113, v114" ::: CLOB);) }
        else if (OP == 1) { REP8(asm volatile("v_min_f32 v100, v101, v102\n v_min_f32 v104, v105, v106\n v_min_f32 v108, v109, v110\n v_min_f32 v112, v113, v114" ::: CLOB);) }
        else { REP8(asm volatile("v_fma_f32 v100, v101, v102, v103\n v_xor_b32 v104, v105, v106\n v_cmp_lt_f32 vcc, v109, v110\n v_cndmask_b32 v112, v113, v114, vcc" ::: CLOB);) }
    }
    asm volatile("s_mov_b64 exec, s[20:21]" ::: CLOB);
    float r;
    asm volatile("v_add_f32 %0, v100, v104\n v_add_f32 %0, %0, v108\n v_add_f32 %0, %0, v112" : "=v"(r)::CLOB);
    if (iters < 0) out[threadIdx.x] = r;
}

template <int OP>
void runmix(const char* op, const char* name, uint64_t mask_a, uint64_t mask_b)
{
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 1 << 24);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kmix<OP><<<256 * 2, 1024>>>(out, 10, mask_a, mask_b);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kmix<OP><<<256 * 2, 1024>>>(out, iters, mask_a, mask_b);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // SIMD cycles per loop-trip instruction slot (32 instructions per trip), all waves of the SIMD together
    printf("%-28s %-52s %7.3f ms = %6.2f SIMD cycles per instruction of the loop (8 wave slots)\n", op, name, best, best * 1e6 / ((double)iters * 32) * 2.4);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int OP>
void sweepmix(const char* op)
{
    const uint64_t F = ~0ull, L = 1ull, Z = 0ull, T = 0xfffull;
    runmix<OP>(op, "8 waves all 64 lanes", F, F);
    runmix<OP>(op, "8 waves lane 0 only", L, L);
    runmix<OP>(op, "4 waves all 64 lanes + 4 waves lane 0 only", F, L);
    runmix<OP>(op, "4 waves all 64 lanes (4 left at once)", F, Z);
    runmix<OP>(op, "4 waves lane 0 only (4 left at once)", L, Z);
    runmix<OP>(op, "4 waves all 64 lanes + 4 waves lanes 0..11", F, T);
    runmix<OP>(op, "4 waves lanes 0..11 (4 left at once)", T, Z);
}

static double g_base = 0.0;

template <int OP>
void run(const char* op, const char* name, uint64_t mask)
{
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 1 << 24);
    const int iters = 4000, ninstr = 32, waves_per_simd = 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * 2, 1024>>>(out, 10, mask);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<OP><<<256 * 2, 1024>>>(out, iters, mask);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double cyc = best * 1e6 / ((double)iters * ninstr * waves_per_simd) * 2.4;
    if (mask == ~0ull) g_base = cyc;
    printf("%-34s %-36s %6.2f cycles  x%.2f of all 64 lanes\n", op, name, cyc, cyc / g_base);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int OP>
void sweep(const char* op)
{
    run<OP>(op, "all 64 lanes", ~0ull);
    run<OP>(op, "lanes 0..31", 0x00000000ffffffffull);
    run<OP>(op, "lanes 32..63", 0xffffffff00000000ull);
    run<OP>(op, "lanes 0..15", 0x000000000000ffffull);
    run<OP>(op, "lanes 0..15 and 32..47", 0x0000ffff0000ffffull);
    run<OP>(op, "lanes 16..31", 0x00000000ffff0000ull);
    run<OP>(op, "every other lane", 0x5555555555555555ull);
    run<OP>(op, "lane 0", 1ull);
    run<OP>(op, "lane 0 and lane 63", 0x8000000000000001ull);
    // how few is few: k live lanes from lane 0, and the same number spread over the wave
    static char names[32][40];
    int n = 0;
    for (int kk : {2, 3, 4, 5, 6, 7, 8, 10, 12, 14}) {
        snprintf(names[n], sizeof names[n], "lanes 0..%d", kk - 1);
        run<OP>(op, names[n++], (1ull << kk) - 1ull);
    }
    run<OP>(op, "lanes 0, 16, 32, 48", 0x0001000100010001ull);
    run<OP>(op, "lanes 0, 8, 16, ... 56 (8 lanes)", 0x0101010101010101ull);
    run<OP>(op, "lanes 0, 4, 8, ... 60 (16 lanes)", 0x1111111111111111ull);
    run<OP>(op, "lane 37", 1ull << 37);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    sweepmix<0>("v_add_f32");
    sweepmix<1>("v_min_f32 (slow class)");
    sweepmix<2>("fma / xor / cmp / cndmask");
    sweep<0>("v_add_f32");
    sweep<1>("v_min_f32 (slow class)");
    sweep<2>("fma / xor / cmp / cndmask");
    sweep<3>("v_lshl_or_b32 (slow class)");
    sweep<4>("v_cmp_lt_f32 vcc (slow class)");
    sweep<5>("v_xor_b32");
    sweep<6>("v_cndmask_b32 vcc");
    return 0;
}
