// Issue cost of VALU instruction classes on gfx950 (decides how vrc_walk.h is written).  Measured, 8 waves/SIMD, cycles per
// instruction per SIMD at a nominal 2.4 GHz: 2-source VGPR-only VOP1/VOP2 (add, mul, xor, shifts, mov, literals) ~2.5-2.75;
// anything that reads or writes an SGPR / VCC (v_cmp, v_cndmask, v_addc, SGPR operands) or has three sources (v_min3, v_add3,
// v_lshl_add, v_bfe) ~4.35; v_pk_add_f32 / v_pk_mul_f32 ~4.8 (for two flops).  e32 vs e64 encoding makes no difference.
// hipcc --offload-arch=gfx950 -O3 tools/micro/pk_issue.hip -o /tmp/pk_issue && /tmp/pk_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* cyc)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a = {1.0f + threadIdx.x, 2.0f}, b = {0.5f, 0.25f}, c = {3.0f, 4.0f}, d = {5.0f, 6.0f};
    float e = 7.0f, f = 8.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 2 independent chains of plain adds: 32 v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
        } else if (MODE == 1) { // 16 v_pk_add_f32 (same flops as mode 0)
            REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if (MODE == 2) { // 16 v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if (MODE == 3) { // 2 independent pk chains: 32 v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));)
        } else if (MODE == 4) { // 32 v_cndmask (VOP3 with SGPR mask)
            REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %2, vcc\n v_cndmask_b32_e64 %1, %1, %3, vcc" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y) : "vcc");)
        } else if (MODE == 5) { // 32 v_min3_f32
            REP16(asm volatile("v_min3_f32 %0, %0, %2, %3\n v_min3_f32 %1, %1, %2, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
        } else if (MODE == 6) { // 32 v_add3_u32 / lshl_add
            REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %2\n v_add3_u32 %1, %1, %2, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
        } else if (MODE == 7) { // 32 v_cmp (e64 to SGPR)
            REP16(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %2\n v_cmp_lt_f32_e32 vcc, %1, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y) : "vcc", "s20", "s21");)
        } else if (MODE == 9) {  // same op, 64-bit encoding
            REP16(asm volatile("v_add_f32_e64 %0, %0, %2\n v_add_f32_e64 %1, %1, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
        } else if (MODE == 10) { // VOP2 with a 32-bit literal
            REP16(asm volatile("v_and_b32_e32 %0, 0x10100, %0\n v_and_b32_e32 %1, 0x10100, %1" : "+v"(e), "+v"(f));)
        } else if (MODE == 11) { // VOP2 with an SGPR operand
            REP16(asm volatile("v_and_b32_e32 %0, s20, %0\n v_and_b32_e32 %1, s20, %1" : "+v"(e), "+v"(f) : : "s20");)
        } else if (MODE == 12) { // e32 compare -> e32 select -> e32 add-with-carry (the per-axis triple)
            REP16(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %2\n v_cndmask_b32_e32 %1, 0, %3, vcc\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y) : "vcc");)
        } else if (MODE == 13) { // the same triple in 64-bit encodings
            REP16(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %2\n v_cndmask_b32_e64 %1, 0, %3, s[20:21]\n v_addc_co_u32_e64 %0, s[22:23], %0, %0, s[20:21]" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y) : "vcc", "s20", "s21", "s22", "s23");)
        } else if (MODE == 14) { // VOP1 / VOP2 integer mix
            REP16(asm volatile("v_xor_b32_e32 %0, %0, %2\n v_lshrrev_b32_e32 %1, %2, %1" : "+v"(e), "+v"(f) : "v"(b.x));)
        } else if (MODE == 15) { // v_mov
            REP16(asm volatile("v_mov_b32_e32 %0, %2\n v_mov_b32_e32 %1, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
        } else if (MODE == 17) { // plain adds with only the low 32 lanes enabled: does a SIMD-32 skip the empty half?
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 exec_hi, 0" : : : "s20", "s21");
            REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
            asm volatile("s_mov_b64 exec, s[20:21]" : : : "s20", "s21");
        } else if (MODE == 18) { // ... with every other lane enabled
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 exec_lo, 0x55555555\n s_mov_b32 exec_hi, 0x55555555" : : : "s20", "s21");
            REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
            asm volatile("s_mov_b64 exec, s[20:21]" : : : "s20", "s21");
        } else if (MODE == 19) { // 3-source op with only the low 32 lanes
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 exec_hi, 0" : : : "s20", "s21");
            REP16(asm volatile("v_min3_f32 %0, %0, %2, %3\n v_min3_f32 %1, %1, %2, %3" : "+v"(e), "+v"(f) : "v"(b.x), "v"(b.y));)
            asm volatile("s_mov_b64 exec, s[20:21]" : : : "s20", "s21");
        } else if (MODE == 8) { // 32 v_addc_co_u32
            REP16(asm volatile("v_addc_co_u32_e64 %0, s[20:21], %0, %0, vcc\n v_addc_co_u32_e64 %1, s[20:21], %1, %1, vcc" : "+v"(e), "+v"(f) : : "vcc", "s20", "s21");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y + c.x + c.y + d.x + e + f;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int ninstr)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int iters = 4000;
        const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;   // 4 SIMDs per CU
        const int blocks_per_cu = (256 * waves_per_simd) / threads;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<256 * blocks_per_cu, threads>>>(out, 10, cyc);
        hipEventRecord(e0);
        k<MODE><<<256 * blocks_per_cu, threads>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        // per-SIMD issue cost per instruction in ns -> cycles at 2.4 GHz
        const double per_instr_ns = ms * 1e6 / ((double)iters * ninstr * waves_per_simd);
        printf("%-34s waves/SIMD %d: %.2f cycles per instruction per SIMD (at 2.4 GHz), wave clock %.1f cyc/instr\n", name, waves_per_simd,
               per_instr_ns * 2.4, (double)h / ((double)iters * ninstr));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    run<0>("v_add_f32 x32 (2 chains)", 32);
    run<1>("v_pk_add_f32 x16 (1 chain)", 16);
    run<2>("v_pk_mul_f32 x16 (1 chain)", 16);
    run<3>("v_pk_add_f32 x32 (2 chains)", 32);
    run<4>("v_cndmask_b32 x32", 32);
    run<5>("v_min3_f32 x32", 32);
    run<6>("v_lshl_add_u32 / v_add3_u32 x32", 32);
    run<7>("v_cmp_lt_f32 x32", 32);
    run<8>("v_addc_co_u32 x32", 32);
    run<9>("v_add_f32_e64 x32", 32);
    run<10>("v_and_b32_e32 literal x32", 32);
    run<11>("v_and_b32_e32 sgpr x32", 32);
    run<12>("cmp/cndmask/addc e32 x48", 48);
    run<13>("cmp/cndmask/addc e64 x48", 48);
    run<14>("v_xor / v_lshrrev e32 x32", 32);
    run<15>("v_mov_b32 x32", 32);
    run<17>("v_add_f32 x32, low 32 lanes", 32);
    run<18>("v_add_f32 x32, even lanes", 32);
    run<19>("v_min3_f32 x32, low 32 lanes", 32);
    return 0;
}
