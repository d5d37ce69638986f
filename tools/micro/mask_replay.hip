// What do the walk loop's thin EXEC masks cost?  tools/micro/exec_halves.hip found that on gfx950 a VALU instruction issued with
// 8 or fewer lanes in EXEC (wherever they are) is SLOWER than with 9 or more: simple ops x1.24, the slow class (compares, min /
// max, shift-ors ...) x3.2 when every wave of the SIMD runs that way, less when mixed with full waves.  The frame kernel's walk
// loop issues its descend / advance / pop blocks under the mask of the lanes that take that path: 21 % of its instructions run
// with <= 8 lanes (tests/tools/path_schedule.py --dump-masks: the block / lane-count sequence of the C3 frame's walks from the
// oracle's traces).  This replays that sequence -- the loop's four blocks with their real instruction mix on dummy registers, no
// memory -- once with the recorded lane counts and once with every count below 9 raised to 9: the difference is what a walk loop
// that never issued a thin mask would gain at most (VALU time only).
// hipcc --offload-arch=gfx950 -O3 tools/micro/mask_replay.hip -o /tmp/mask_replay && /tmp/mask_replay profiles/r05/walk_mask_sequence.u32
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
             "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "vcc", "s20", "s21", "s22", "s23"

// px v100 py v101 pz v102 | tcx..z v103-105 | tox..z v106-108 | cx cy cz v109-111 | tc v112 | m v113 t v114 t2 v115 kind v116
// co v117 mirror v118 sf v119 tmin v120 pid v121 sp v122 normal v123 node v124:125 stkb v126
#define B_HEADER                                                                         \
    "v_xor_b32 v116, v117, v118\n v_mul_f32 v109, v100, v103\n v_mul_f32 v110, v101, v104\n v_mul_f32 v111, v102, v105\n" \
    "v_lshrrev_b32 v113, v116, v124\n v_sub_f32 v109, v109, v106\n v_sub_f32 v110, v110, v107\n v_sub_f32 v111, v111, v108\n" \
    "v_and_b32 v114, 0x10100, v113\n v_min3_f32 v112, v109, v110, v111\n v_cmp_ne_u32 vcc, 0, v114\n v_min_f32 v115, 1.0, v112\n" \
    "v_cmp_le_f32 vcc, v120, v115\n v_cmp_lt_u32 vcc, 0x100, v114\n"
#define B_DESCEND                                                                        \
    "v_mul_f32 v119, 0.5, v119\n v_add3_u32 v121, v121, v125, v116\n v_mul_f32 v113, v103, v119\n v_mul_f32 v114, v104, v119\n" \
    "v_mul_f32 v115, v105, v119\n v_add_f32 v113, v113, v109\n v_add_f32 v114, v114, v110\n v_add_f32 v115, v115, v111\n" \
    "v_sub_f32 v113, v120, v113\n v_sub_f32 v114, v120, v114\n v_sub_f32 v115, v120, v115\n v_add_u32 v122, 0xfffff800, v122\n" \
    "v_ashrrev_i32 v113, 31, v113\n v_ashrrev_i32 v114, 31, v114\n v_ashrrev_i32 v115, 31, v115\n v_and_b32 v117, 1, v113\n" \
    "v_and_or_b32 v117, v114, 2, v117\n v_and_or_b32 v117, v115, 4, v117\n v_and_b32 v113, v119, v113\n v_and_b32 v114, v119, v114\n" \
    "v_and_b32 v115, v119, v115\n v_add_f32 v100, v100, v113\n v_add_f32 v101, v101, v114\n v_add_f32 v102, v102, v115\n"
#define B_ADVANCE                                                                        \
    "v_cmp_le_f32 s[20:21], v109, v112\n v_cmp_le_f32 s[22:23], v110, v112\n v_cmp_le_f32 vcc, v111, v112\n v_mov_b32 v120, v112\n" \
    "v_cndmask_b32 v123, 0, 1, vcc\n v_cndmask_b32 v116, 0, v119, s[20:21]\n v_addc_co_u32 v123, s[20:21], v123, v123, s[22:23]\n" \
    "v_cndmask_b32 v113, 0, v119, s[22:23]\n v_cndmask_b32 v115, 0, v119, vcc\n v_addc_co_u32 v123, s[22:23], v123, v123, s[20:21]\n" \
    "v_sub_f32 v100, v100, v116\n v_sub_f32 v101, v101, v113\n v_sub_f32 v102, v102, v115\n v_xor_b32 v117, v117, v123\n" \
    "v_and_b32 v114, v117, v123\n v_cmp_ne_u32 vcc, 0, v114\n"
#define B_POP                                                                            \
    "v_add_f32 v109, v100, v116\n v_add_f32 v110, v101, v113\n v_add_f32 v111, v102, v115\n v_xor_b32 v109, v109, v100\n" \
    "v_xor_b32 v110, v110, v101\n v_xor_b32 v111, v111, v102\n v_or3_b32 v114, v109, v110, v111\n v_ffbh_u32 v114, v114\n" \
    "v_sub_u32 v114, 31, v114\n v_cmp_le_u32 vcc, 23, v114\n v_lshl_add_u32 v122, v114, 11, v126\n v_lshl_add_u32 v119, v114, 23, v126\n" \
    "v_lshlrev_b32 v115, v114, -1\n v_bfe_u32 v109, v100, v114, 1\n v_bfe_u32 v110, v101, v114, 1\n v_bfe_u32 v111, v102, v114, 1\n" \
    "v_and_b32 v100, v100, v115\n v_and_b32 v101, v101, v115\n v_and_b32 v102, v102, v115\n v_lshl_or_b32 v117, v110, 1, v109\n" \
    "v_lshl_or_b32 v117, v111, 2, v117\n"

template <int PAD>
__global__ void replay(const uint32_t* __restrict__ seq, uint32_t n, uint32_t entries, float* out)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t pos = (wave * 9973u) % n;
    asm volatile("v_mov_b32 v100, 1.25\n v_mov_b32 v101, 1.5\n v_mov_b32 v102, 1.75\n v_mov_b32 v103, 2.0\n v_mov_b32 v104, 3.0\n"
                 "v_mov_b32 v105, 4.0\n v_mov_b32 v106, 1.0\n v_mov_b32 v107, 2.0\n v_mov_b32 v108, 0.5\n v_mov_b32 v109, 0\n"
                 "v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"
                 "v_mov_b32 v116, 0\n v_mov_b32 v117, 3\n v_mov_b32 v118, 5\n v_mov_b32 v119, 0.5\n v_mov_b32 v120, 0\n v_mov_b32 v121, 7\n"
                 "v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v124, 0x1ff00\n v_mov_b32 v125, 9\n v_mov_b32 v126, 64\n" ::: CLOB);
    uint32_t e = seq[pos];
    for (uint32_t i = 0; i < entries; ++i) {
        pos = pos + 1 == n ? 0 : pos + 1;
        const uint32_t next = seq[pos];                           // (uniform: a scalar load, in flight during the block)
        uint32_t lanes = e & 255u;
        const uint32_t b = e >> 8;
        if (PAD && lanes < (uint32_t)PAD) lanes = PAD;
        const uint64_t mask = lanes >= 64u ? ~0ull : (1ull << lanes) - 1ull;
        if (b == 0) asm volatile("s_mov_b64 exec, %0\n" B_HEADER "s_mov_b64 exec, -1\n" ::"s"(mask) : CLOB);
        else if (b == 1) asm volatile("s_mov_b64 exec, %0\n" B_DESCEND "s_mov_b64 exec, -1\n" ::"s"(mask) : CLOB);
        else if (b == 2) asm volatile("s_mov_b64 exec, %0\n" B_ADVANCE "s_mov_b64 exec, -1\n" ::"s"(mask) : CLOB);
        else asm volatile("s_mov_b64 exec, %0\n" B_POP "s_mov_b64 exec, -1\n" ::"s"(mask) : CLOB);
        e = next;
    }
    float r;
    asm volatile("v_add_f32 %0, v100, v101\n v_add_f32 %0, %0, v102" : "=v"(r)::CLOB);
    if (entries == 0xffffffffu) out[threadIdx.x] = r;
}

#include "mask_replay_gen.inc"
// The same blocks as straight-line code (tools/micro/gen_mask_replay.py): no sequence fetch, no dispatch.
template <int VARIANT>
__global__ void replay_line(uint32_t trips, float* out)
{
    asm volatile("v_mov_b32 v100, 1.25\n v_mov_b32 v101, 1.5\n v_mov_b32 v102, 1.75\n v_mov_b32 v103, 2.0\n v_mov_b32 v104, 3.0\n"
                 "v_mov_b32 v105, 4.0\n v_mov_b32 v106, 1.0\n v_mov_b32 v107, 2.0\n v_mov_b32 v108, 0.5\n v_mov_b32 v109, 0\n"
                 "v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"
                 "v_mov_b32 v116, 0\n v_mov_b32 v117, 3\n v_mov_b32 v118, 5\n v_mov_b32 v119, 0.5\n v_mov_b32 v120, 0\n v_mov_b32 v121, 7\n"
                 "v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v124, 0x1ff00\n v_mov_b32 v125, 9\n v_mov_b32 v126, 64\n" ::: CLOB);
    // (waves of a SIMD drift apart by themselves; a different start delay per wave makes sure they begin apart)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t d = 0; d < wave * 37u; ++d) asm volatile("s_nop 7" ::: CLOB);
    for (uint32_t i = 0; i < trips; ++i) {
        if (VARIANT == 0) asm volatile(GEN_RECORDED "s_mov_b64 exec, -1\n" ::: CLOB);
        else if (VARIANT == 1) asm volatile(GEN_PAD9 "s_mov_b64 exec, -1\n" ::: CLOB);
        else if (VARIANT == 2) asm volatile(GEN_PAD16 "s_mov_b64 exec, -1\n" ::: CLOB);
        else asm volatile(GEN_ALL64 "s_mov_b64 exec, -1\n" ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v100, v101\n v_add_f32 %0, %0, v102" : "=v"(r)::CLOB);
    if (trips == 0xffffffffu) out[threadIdx.x] = r;
}

template <int VARIANT>
double run_line(float* out, int waves_per_simd, const char* name, double base)
{
    const uint32_t trips = 60;
    const int threads = waves_per_simd * 4 * 64 / 2;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    replay_line<VARIANT><<<512, threads>>>(2, out);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        replay_line<VARIANT><<<512, threads>>>(trips, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("straight-line, %d waves per SIMD  %-40s %8.3f ms  %+6.1f %%\n", waves_per_simd, name, best, base > 0 ? (best / base - 1.0) * 100.0 : 0.0);
    return best;
}

template <int PAD>
double run(const uint32_t* seq, uint32_t n, float* out, int waves_per_simd, const char* name, double base)
{
    const uint32_t entries = 20000;
    const int threads = waves_per_simd * 4 * 64 / 2;              // two workgroups per CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    replay<PAD><<<512, threads>>>(seq, n, 200, out);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        replay<PAD><<<512, threads>>>(seq, n, entries, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%d waves per SIMD  %-46s %8.3f ms  %+6.1f %%\n", waves_per_simd, name, best, base > 0 ? (best / base - 1.0) * 100.0 : 0.0);
    return best;
}

int main(int argc, char** argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* path = argc > 1 ? argv[1] : "profiles/r05/walk_mask_sequence.u32";
    FILE* f = fopen(path, "rb");
    if (!f) { printf("cannot open %s\n", path); return 1; }
    std::vector<uint32_t> h;
    uint32_t v;
    while (fread(&v, 4, 1, f) == 1) h.push_back(v);
    fclose(f);
    size_t thin = 0;
    for (uint32_t e : h) thin += (e & 255u) <= 8u;
    printf("%zu block issues, %.1f %% of them with <= 8 lanes\n", h.size(), 100.0 * thin / h.size());
    uint32_t* seq; float* out;
    (void)hipMalloc(&seq, h.size() * 4); (void)hipMalloc(&out, 1 << 20);
    (void)hipMemcpy(seq, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("straight-line window: %d block issues, %.1f %% of them with <= 8 lanes\n", GEN_ENTRIES, GEN_THIN * 100.0);
    for (int w : {4, 6, 8}) {
        const double b = run_line<0>(out, w, "recorded lane counts", 0.0);
        run_line<1>(out, w, "counts below 9 raised to 9", b);
        run_line<2>(out, w, "counts below 16 raised to 16", b);
        run_line<3>(out, w, "every block with all 64 lanes", b);
    }
    for (int w : {6, 8}) {
        const double b = run<0>(seq, (uint32_t)h.size(), out, w, "recorded lane counts", 0.0);
        run<9>(seq, (uint32_t)h.size(), out, w, "counts below 9 raised to 9", b);
        run<16>(seq, (uint32_t)h.size(), out, w, "counts below 16 raised to 16", b);
        run<64>(seq, (uint32_t)h.size(), out, w, "every block with all 64 lanes", b);
    }
    return 0;
}
