// Issue cost per opcode on gfx950, 8 waves per SIMD (round 3: which of the walk's instructions are "slow class", and what
// could replace them).  Every test is 4 independent chains x 8 = 32 instructions per loop trip on explicitly named registers
// (v100..v115), so that operand banks are what the test says and not what the allocator chose.  Reported: cycles of SIMD
// issue per wave64 instruction at a nominal 2.4 GHz, and the ratio to v_add_f32 measured in the same process.
// hipcc --offload-arch=gfx950 -O3 tools/micro/op_issue.hip -o /tmp/op_issue && /tmp/op_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
             "v114", "v115", "v116", "v117", "v118", "v119", "vcc", "s20", "s21", "s22", "s23", "s24", "s25"
#define REP8(x) x x x x x x x x

// One "quad" = 4 instructions, one per chain.
#define T(ID, QUAD) \
    else if (MODE == ID) { REP8(asm volatile(QUAD ::: CLOB);) }

template <int MODE>
__global__ void k(float* out, int iters)
{
    asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 2.0\n v_mov_b32 v102, 0.5\n v_mov_b32 v103, 4.0\n"
                 "v_mov_b32 v104, 1.0\n v_mov_b32 v105, 2.0\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 4.0\n"
                 "v_mov_b32 v108, 1.0\n v_mov_b32 v109, 2.0\n v_mov_b32 v110, 0.5\n v_mov_b32 v111, 4.0\n"
                 "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 2.0\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 4.0\n"
                 "v_mov_b32 v116, 1.0\n v_mov_b32 v117, 2.0\n v_mov_b32 v118, 0.5\n v_mov_b32 v119, 4.0\n"
                 "s_mov_b64 s[20:21], exec\n s_mov_b64 s[22:23], exec\n s_mov_b32 s24, 3\n s_mov_b64 vcc, exec" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(asm volatile("v_add_f32 v100, v101, v102\n v_add_f32 v104, v105, v106\n v_add_f32 v108, v109, v110\n v_add_f32 v112, v113, v114" ::: CLOB);) }
        // ---- banks: sources in the same bank (reg % 4), destination bank
        T(1, "v_add_f32 v100, v104, v108\n v_add_f32 v101, v105, v109\n v_add_f32 v102, v106, v110\n v_add_f32 v103, v107, v111\n")
        T(2, "v_add_f32 v100, v100, v101\n v_add_f32 v104, v104, v105\n v_add_f32 v108, v108, v109\n v_add_f32 v112, v112, v113\n")
        // ---- three VGPR sources
        T(3, "v_fma_f32 v100, v101, v102, v103\n v_fma_f32 v104, v105, v106, v107\n v_fma_f32 v108, v109, v110, v111\n v_fma_f32 v112, v113, v114, v115\n")
        T(4, "v_fma_f32 v100, v104, v108, v112\n v_fma_f32 v101, v105, v109, v113\n v_fma_f32 v102, v106, v110, v114\n v_fma_f32 v103, v107, v111, v115\n")
        T(5, "v_fmac_f32 v100, v101, v102\n v_fmac_f32 v104, v105, v106\n v_fmac_f32 v108, v109, v110\n v_fmac_f32 v112, v113, v114\n")
        T(6, "v_min3_f32 v100, v101, v102, v103\n v_min3_f32 v104, v105, v106, v107\n v_min3_f32 v108, v109, v110, v111\n v_min3_f32 v112, v113, v114, v115\n")
        T(7, "v_min3_f32 v100, v101, v101, v102\n v_min3_f32 v104, v105, v105, v106\n v_min3_f32 v108, v109, v109, v110\n v_min3_f32 v112, v113, v113, v114\n")
        T(8, "v_med3_f32 v100, v101, v102, v103\n v_med3_f32 v104, v105, v106, v107\n v_med3_f32 v108, v109, v110, v111\n v_med3_f32 v112, v113, v114, v115\n")
        T(9, "v_lshl_or_b32 v100, v101, 1, v102\n v_lshl_or_b32 v104, v105, 1, v106\n v_lshl_or_b32 v108, v109, 1, v110\n v_lshl_or_b32 v112, v113, 1, v114\n")
        T(10, "v_lshl_or_b32 v100, v101, v102, v103\n v_lshl_or_b32 v104, v105, v106, v107\n v_lshl_or_b32 v108, v109, v110, v111\n v_lshl_or_b32 v112, v113, v114, v115\n")
        T(11, "v_lshl_add_u32 v100, v101, 11, v102\n v_lshl_add_u32 v104, v105, 11, v106\n v_lshl_add_u32 v108, v109, 11, v110\n v_lshl_add_u32 v112, v113, 11, v114\n")
        T(12, "v_add3_u32 v100, v101, v102, v103\n v_add3_u32 v104, v105, v106, v107\n v_add3_u32 v108, v109, v110, v111\n v_add3_u32 v112, v113, v114, v115\n")
        T(13, "v_or3_b32 v100, v101, v102, v103\n v_or3_b32 v104, v105, v106, v107\n v_or3_b32 v108, v109, v110, v111\n v_or3_b32 v112, v113, v114, v115\n")
        T(14, "v_and_or_b32 v100, v101, v102, v103\n v_and_or_b32 v104, v105, v106, v107\n v_and_or_b32 v108, v109, v110, v111\n v_and_or_b32 v112, v113, v114, v115\n")
        T(15, "v_bfe_u32 v100, v101, v102, 1\n v_bfe_u32 v104, v105, v106, 1\n v_bfe_u32 v108, v109, v110, 1\n v_bfe_u32 v112, v113, v114, 1\n")
        T(16, "v_bfi_b32 v100, v101, v102, v103\n v_bfi_b32 v104, v105, v106, v107\n v_bfi_b32 v108, v109, v110, v111\n v_bfi_b32 v112, v113, v114, v115\n")
        T(17, "v_alignbit_b32 v100, v101, v102, v103\n v_alignbit_b32 v104, v105, v106, v107\n v_alignbit_b32 v108, v109, v110, v111\n v_alignbit_b32 v112, v113, v114, v115\n")
        T(18, "v_perm_b32 v100, v101, v102, v103\n v_perm_b32 v104, v105, v106, v107\n v_perm_b32 v108, v109, v110, v111\n v_perm_b32 v112, v113, v114, v115\n")
        T(19, "v_mad_u32_u24 v100, v101, v102, v103\n v_mad_u32_u24 v104, v105, v106, v107\n v_mad_u32_u24 v108, v109, v110, v111\n v_mad_u32_u24 v112, v113, v114, v115\n")
        T(20, "v_xad_u32 v100, v101, v102, v103\n v_xad_u32 v104, v105, v106, v107\n v_xad_u32 v108, v109, v110, v111\n v_xad_u32 v112, v113, v114, v115\n")
        T(21, "v_add_lshl_u32 v100, v101, v102, 3\n v_add_lshl_u32 v104, v105, v106, 3\n v_add_lshl_u32 v108, v109, v110, 3\n v_add_lshl_u32 v112, v113, v114, 3\n")
        // ---- SGPR / VCC / EXEC traffic
        T(22, "v_cmp_lt_f32_e32 vcc, v100, v101\n v_cmp_lt_f32_e32 vcc, v104, v105\n v_cmp_lt_f32_e32 vcc, v108, v109\n v_cmp_lt_f32_e32 vcc, v112, v113\n")
        T(23, "v_cmp_lt_f32_e64 s[20:21], v100, v101\n v_cmp_lt_f32_e64 s[22:23], v104, v105\n v_cmp_lt_f32_e64 s[20:21], v108, v109\n v_cmp_lt_f32_e64 s[22:23], v112, v113\n")
        T(24, "v_cmp_lt_u32_e32 vcc, v100, v101\n v_cmp_lt_u32_e32 vcc, v104, v105\n v_cmp_lt_u32_e32 vcc, v108, v109\n v_cmp_lt_u32_e32 vcc, v112, v113\n")
        T(25, "v_cndmask_b32_e32 v100, v101, v102, vcc\n v_cndmask_b32_e32 v104, v105, v106, vcc\n v_cndmask_b32_e32 v108, v109, v110, vcc\n v_cndmask_b32_e32 v112, v113, v114, vcc\n")
        T(26, "v_cndmask_b32_e64 v100, 0, v102, s[20:21]\n v_cndmask_b32_e64 v104, 0, v106, s[22:23]\n v_cndmask_b32_e64 v108, 0, v110, s[20:21]\n v_cndmask_b32_e64 v112, 0, v114, s[22:23]\n")
        T(27, "v_addc_co_u32_e32 v100, vcc, v101, v102, vcc\n v_addc_co_u32_e32 v104, vcc, v105, v106, vcc\n v_addc_co_u32_e32 v108, vcc, v109, v110, vcc\n v_addc_co_u32_e32 v112, vcc, v113, v114, vcc\n")
        T(28, "v_add_co_u32_e32 v100, vcc, v101, v102\n v_add_co_u32_e32 v104, vcc, v105, v106\n v_add_co_u32_e32 v108, vcc, v109, v110\n v_add_co_u32_e32 v112, vcc, v113, v114\n")
        T(29, "v_and_b32_e32 v100, s24, v101\n v_and_b32_e32 v104, s24, v105\n v_and_b32_e32 v108, s24, v109\n v_and_b32_e32 v112, s24, v113\n")
        T(30, "v_mul_f32_e32 v100, s24, v101\n v_mul_f32_e32 v104, s24, v105\n v_mul_f32_e32 v108, s24, v109\n v_mul_f32_e32 v112, s24, v113\n")
        T(31, "v_readfirstlane_b32 s24, v100\n v_readfirstlane_b32 s25, v104\n v_readfirstlane_b32 s24, v108\n v_readfirstlane_b32 s25, v112\n")
        T(32, "v_readlane_b32 s24, v100, 3\n v_readlane_b32 s25, v104, 5\n v_readlane_b32 s24, v108, 7\n v_readlane_b32 s25, v112, 9\n")
        T(33, "v_writelane_b32 v100, s24, 3\n v_writelane_b32 v104, s24, 5\n v_writelane_b32 v108, s24, 7\n v_writelane_b32 v112, s24, 9\n")
        // ---- plain two-source integer / float
        T(34, "v_mov_b32_e32 v100, v101\n v_mov_b32_e32 v104, v105\n v_mov_b32_e32 v108, v109\n v_mov_b32_e32 v112, v113\n")
        T(35, "v_lshlrev_b32_e32 v100, v101, v102\n v_lshlrev_b32_e32 v104, v105, v106\n v_lshlrev_b32_e32 v108, v109, v110\n v_lshlrev_b32_e32 v112, v113, v114\n")
        T(36, "v_xor_b32_e32 v100, v101, v102\n v_xor_b32_e32 v104, v105, v106\n v_xor_b32_e32 v108, v109, v110\n v_xor_b32_e32 v112, v113, v114\n")
        T(37, "v_min_f32_e32 v100, v101, v102\n v_min_f32_e32 v104, v105, v106\n v_min_f32_e32 v108, v109, v110\n v_min_f32_e32 v112, v113, v114\n")
        T(38, "v_min_u32_e32 v100, v101, v102\n v_min_u32_e32 v104, v105, v106\n v_min_u32_e32 v108, v109, v110\n v_min_u32_e32 v112, v113, v114\n")
        T(39, "v_sub_u32_e32 v100, v101, v102\n v_sub_u32_e32 v104, v105, v106\n v_sub_u32_e32 v108, v109, v110\n v_sub_u32_e32 v112, v113, v114\n")
        T(40, "v_ffbh_u32_e32 v100, v101\n v_ffbh_u32_e32 v104, v105\n v_ffbh_u32_e32 v108, v109\n v_ffbh_u32_e32 v112, v113\n")
        T(41, "v_mul_u32_u24_e32 v100, v101, v102\n v_mul_u32_u24_e32 v104, v105, v106\n v_mul_u32_u24_e32 v108, v109, v110\n v_mul_u32_u24_e32 v112, v113, v114\n")
        T(42, "v_mul_lo_u32 v100, v101, v102\n v_mul_lo_u32 v104, v105, v106\n v_mul_lo_u32 v108, v109, v110\n v_mul_lo_u32 v112, v113, v114\n")
        T(43, "v_mul_hi_u32 v100, v101, v102\n v_mul_hi_u32 v104, v105, v106\n v_mul_hi_u32 v108, v109, v110\n v_mul_hi_u32 v112, v113, v114\n")
        T(44, "v_cvt_f32_u32_e32 v100, v101\n v_cvt_f32_u32_e32 v104, v105\n v_cvt_f32_u32_e32 v108, v109\n v_cvt_f32_u32_e32 v112, v113\n")
        T(45, "v_rcp_f32_e32 v100, v101\n v_rcp_f32_e32 v104, v105\n v_rcp_f32_e32 v108, v109\n v_rcp_f32_e32 v112, v113\n")
        T(46, "v_rsq_f32_e32 v100, v101\n v_rsq_f32_e32 v104, v105\n v_rsq_f32_e32 v108, v109\n v_rsq_f32_e32 v112, v113\n")
        // ---- packed / 64-bit
        T(47, "v_pk_mul_f32 v[100:101], v[104:105], v[108:109]\n v_pk_mul_f32 v[102:103], v[106:107], v[110:111]\n v_pk_mul_f32 v[112:113], v[104:105], v[108:109]\n v_pk_mul_f32 v[114:115], v[106:107], v[110:111]\n")
        T(48, "v_pk_add_f32 v[100:101], v[104:105], v[108:109]\n v_pk_add_f32 v[102:103], v[106:107], v[110:111]\n v_pk_add_f32 v[112:113], v[104:105], v[108:109]\n v_pk_add_f32 v[114:115], v[106:107], v[110:111]\n")
        T(49, "v_pk_fma_f32 v[100:101], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[102:103], v[106:107], v[110:111], v[114:115]\n v_pk_fma_f32 v[116:117], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[118:119], v[106:107], v[110:111], v[114:115]\n")
        T(50, "v_pk_mov_b32 v[100:101], v[104:105], v[108:109]\n v_pk_mov_b32 v[102:103], v[106:107], v[110:111]\n v_pk_mov_b32 v[112:113], v[104:105], v[108:109]\n v_pk_mov_b32 v[114:115], v[106:107], v[110:111]\n")
        T(51, "v_mov_b64 v[100:101], v[104:105]\n v_mov_b64 v[102:103], v[106:107]\n v_mov_b64 v[112:113], v[108:109]\n v_mov_b64 v[114:115], v[110:111]\n")
        T(52, "v_lshlrev_b64 v[100:101], 3, v[104:105]\n v_lshlrev_b64 v[102:103], 3, v[106:107]\n v_lshlrev_b64 v[112:113], 3, v[108:109]\n v_lshlrev_b64 v[114:115], 3, v[110:111]\n")
        T(53, "v_pk_add_f32 v[100:101], v[104:105], v[108:109] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[102:103], v[106:107], v[110:111] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[112:113], v[104:105], v[108:109] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 v[114:115], v[106:107], v[110:111] neg_lo:[0,1] neg_hi:[0,1]\n")
        T(54, "v_pk_add_u16 v100, v101, v102\n v_pk_add_u16 v104, v105, v106\n v_pk_add_u16 v108, v109, v110\n v_pk_add_u16 v112, v113, v114\n")
        T(55, "v_pk_lshlrev_b16 v100, v101, v102\n v_pk_lshlrev_b16 v104, v105, v106\n v_pk_lshlrev_b16 v108, v109, v110\n v_pk_lshlrev_b16 v112, v113, v114\n")
        // ---- SDWA / DPP
        T(56, "v_add_u32_sdwa v100, v101, v102 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa v104, v105, v106 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa v108, v109, v110 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:DWORD\n v_add_u32_sdwa v112, v113, v114 dst_sel:DWORD src0_sel:BYTE_0 src1_sel:DWORD\n")
        T(57, "v_mov_b32_dpp v100, v101 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v104, v105 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v108, v109 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v112, v113 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
        T(58, "v_cmp_lt_f32_sdwa vcc, v100, v101 src0_sel:DWORD src1_sel:DWORD\n v_cmp_lt_f32_sdwa vcc, v104, v105 src0_sel:DWORD src1_sel:DWORD\n v_cmp_lt_f32_sdwa vcc, v108, v109 src0_sel:DWORD src1_sel:DWORD\n v_cmp_lt_f32_sdwa vcc, v112, v113 src0_sel:DWORD src1_sel:DWORD\n")
        // ---- compares that write EXEC
        T(59, "v_cmpx_le_f32_e32 vcc, v100, v100\n v_cmpx_le_f32_e32 vcc, v104, v104\n v_cmpx_le_f32_e32 vcc, v108, v108\n v_cmpx_le_f32_e32 vcc, v112, v112\n")
        // ---- mixes as they occur in the walk
        T(60, "v_cmp_le_f32_e32 vcc, v100, v101\n v_cndmask_b32_e32 v104, 0, v105, vcc\n v_sub_f32_e32 v108, v108, v104\n v_xor_b32_e32 v112, v113, v114\n")
        T(61, "v_sub_f32_e32 v100, v101, v102\n v_lshrrev_b32_e32 v104, 31, v100\n v_lshl_or_b32 v108, v104, v106, v108\n v_xor_b32_e32 v112, v113, v114\n")
        // two-source VOP2 with an inline constant / a literal
        T(62, "v_and_b32_e32 v100, 0x10100, v101\n v_and_b32_e32 v104, 0x10100, v105\n v_and_b32_e32 v108, 0x10100, v109\n v_and_b32_e32 v112, 0x10100, v113\n")
        T(63, "v_mul_f32_e32 v100, 0.5, v101\n v_mul_f32_e32 v104, 0.5, v105\n v_mul_f32_e32 v108, 0.5, v109\n v_mul_f32_e32 v112, 0.5, v113\n")
        T(64, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
        T(65, "v_nop\n v_nop\n v_nop\n v_nop\n")
        T(66, "v_add_f32_e64 v100, v101, v102 clamp\n v_add_f32_e64 v104, v105, v106 clamp\n v_add_f32_e64 v108, v109, v110 clamp\n v_add_f32_e64 v112, v113, v114 clamp\n")
        T(67, "v_add_f32_e64 v100, |v101|, -v102\n v_add_f32_e64 v104, |v105|, -v106\n v_add_f32_e64 v108, |v109|, -v110\n v_add_f32_e64 v112, |v113|, -v114\n")
        T(68, "v_cvt_pkrtz_f16_f32 v100, v101, v102\n v_cvt_pkrtz_f16_f32 v104, v105, v106\n v_cvt_pkrtz_f16_f32 v108, v109, v110\n v_cvt_pkrtz_f16_f32 v112, v113, v114\n")
        T(69, "v_mbcnt_lo_u32_b32 v100, -1, v101\n v_mbcnt_lo_u32_b32 v104, -1, v105\n v_mbcnt_lo_u32_b32 v108, -1, v109\n v_mbcnt_lo_u32_b32 v112, -1, v113\n")
        T(70, "v_cmp_class_f32_e32 vcc, v100, v101\n v_cmp_class_f32_e32 vcc, v104, v105\n v_cmp_class_f32_e32 vcc, v108, v109\n v_cmp_class_f32_e32 vcc, v112, v113\n")
        T(71, "v_ldexp_f32 v100, v101, v102\n v_ldexp_f32 v104, v105, v106\n v_ldexp_f32 v108, v109, v110\n v_ldexp_f32 v112, v113, v114\n")
        T(72, "v_max_f32_e32 v100, v101, v102\n v_fma_f32 v104, v105, v106, v107\n v_max_f32_e32 v108, v109, v110\n v_fma_f32 v112, v113, v114, v115\n")
        T(73, "v_fma_f32 v100, v101, 0.5, v103\n v_fma_f32 v104, v105, 0.5, v107\n v_fma_f32 v108, v109, 0.5, v111\n v_fma_f32 v112, v113, 0.5, v115\n")
        T(74, "v_mad_i32_i24 v100, v101, v102, v103\n v_mad_i32_i24 v104, v105, v106, v107\n v_mad_i32_i24 v108, v109, v110, v111\n v_mad_i32_i24 v112, v113, v114, v115\n")
        T(75, "v_sad_u32 v100, v101, v102, v103\n v_sad_u32 v104, v105, v106, v107\n v_sad_u32 v108, v109, v110, v111\n v_sad_u32 v112, v113, v114, v115\n")
        T(76, "v_lerp_u8 v100, v101, v102, v103\n v_lerp_u8 v104, v105, v106, v107\n v_lerp_u8 v108, v109, v110, v111\n v_lerp_u8 v112, v113, v114, v115\n")
        T(77, "v_cubeid_f32 v100, v101, v102, v103\n v_cubeid_f32 v104, v105, v106, v107\n v_cubeid_f32 v108, v109, v110, v111\n v_cubeid_f32 v112, v113, v114, v115\n")
        T(78, "v_max3_u32 v100, v101, v102, v103\n v_max3_u32 v104, v105, v106, v107\n v_max3_u32 v108, v109, v110, v111\n v_max3_u32 v112, v113, v114, v115\n")
        T(79, "v_dot2c_f32_f16 v100, v101, v102\n v_dot2c_f32_f16 v104, v105, v106\n v_dot2c_f32_f16 v108, v109, v110\n v_dot2c_f32_f16 v112, v113, v114\n")
    }
    float r;
    asm volatile("v_add_f32 %0, v100, v104\n v_add_f32 %0, %0, v108\n v_add_f32 %0, %0, v112" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double g_base = 0.0;

template <int MODE>
void run(const char* name)
{
    static float* out = nullptr;
    if (!out) hipMalloc(&out, 1 << 24);
    const int iters = 4000, ninstr = 32, waves_per_simd = 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * 2, 1024>>>(out, 10);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<256 * 2, 1024>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double cyc = best * 1e6 / ((double)iters * ninstr * waves_per_simd) * 2.4;
    if (MODE == 0) g_base = cyc;
    printf("%-44s %6.2f cycles  x%.2f\n", name, cyc, cyc / g_base);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

#define R(ID, NAME) run<ID>(NAME);

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    R(0, "v_add_f32 (banks 1,2 -> 0)") R(1, "v_add_f32 sources in one bank") R(2, "v_add_f32 dst = src0")
    R(3, "v_fma_f32 3 VGPRs, banks 1,2,3") R(4, "v_fma_f32 3 VGPRs, one bank") R(5, "v_fmac_f32") R(73, "v_fma_f32 v, 0.5, v")
    R(72, "v_max_f32 / v_fma_f32 alternating")
    R(6, "v_min3_f32 3 VGPRs") R(7, "v_min3_f32 v, v(same), v") R(8, "v_med3_f32") R(78, "v_max3_u32")
    R(9, "v_lshl_or_b32 v, 1, v") R(10, "v_lshl_or_b32 v, v, v") R(11, "v_lshl_add_u32 v, 11, v") R(12, "v_add3_u32")
    R(13, "v_or3_b32") R(14, "v_and_or_b32") R(15, "v_bfe_u32 v, v, 1") R(16, "v_bfi_b32") R(17, "v_alignbit_b32")
    R(18, "v_perm_b32") R(19, "v_mad_u32_u24") R(74, "v_mad_i32_i24") R(20, "v_xad_u32") R(21, "v_add_lshl_u32") R(75, "v_sad_u32")
    R(76, "v_lerp_u8") R(77, "v_cubeid_f32")
    R(22, "v_cmp_lt_f32_e32 vcc") R(23, "v_cmp_lt_f32_e64 sgpr") R(24, "v_cmp_lt_u32_e32 vcc") R(70, "v_cmp_class_f32") R(58, "v_cmp_lt_f32_sdwa vcc")
    R(59, "v_cmpx_le_f32_e32")
    R(25, "v_cndmask_b32_e32 v, v, vcc") R(26, "v_cndmask_b32_e64 0, v, sgpr") R(27, "v_addc_co_u32") R(28, "v_add_co_u32")
    R(29, "v_and_b32 sgpr, v") R(30, "v_mul_f32 sgpr, v") R(31, "v_readfirstlane_b32") R(32, "v_readlane_b32") R(33, "v_writelane_b32")
    R(34, "v_mov_b32") R(35, "v_lshlrev_b32 v, v") R(36, "v_xor_b32") R(37, "v_min_f32") R(38, "v_min_u32") R(39, "v_sub_u32")
    R(40, "v_ffbh_u32") R(41, "v_mul_u32_u24") R(42, "v_mul_lo_u32") R(43, "v_mul_hi_u32") R(44, "v_cvt_f32_u32") R(45, "v_rcp_f32")
    R(46, "v_rsq_f32") R(71, "v_ldexp_f32") R(68, "v_cvt_pkrtz_f16_f32") R(69, "v_mbcnt_lo_u32_b32") R(79, "v_dot2c_f32_f16")
    R(47, "v_pk_mul_f32") R(48, "v_pk_add_f32") R(53, "v_pk_add_f32 neg") R(49, "v_pk_fma_f32") R(50, "v_pk_mov_b32") R(51, "v_mov_b64")
    R(52, "v_lshlrev_b64") R(54, "v_pk_add_u16") R(55, "v_pk_lshlrev_b16")
    R(56, "v_add_u32_sdwa") R(57, "v_mov_b32_dpp quad_perm")
    R(60, "mix: cmp / cndmask / sub / xor") R(61, "mix: sub / lshr / lshl_or / xor")
    R(62, "v_and_b32 literal") R(63, "v_mul_f32 0.5, v") R(64, "s_nop 0") R(65, "v_nop") R(66, "v_add_f32_e64 clamp") R(67, "v_add_f32_e64 |a|, -b")
    return 0;
}
