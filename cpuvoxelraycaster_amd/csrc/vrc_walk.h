// vrc_walk.h -- the octree walk (the while loop of LSVO<N>::castRay, lsvo.hpp:72-146) as one
// hand-scheduled gfx950 loop.  Included by vrc_device.h; `ray_step` there is the same loop in C++
// (kept as the readable statement of the algorithm and as the A/B reference, -DVRC_WALK_CPP).
//
// Why assembly: the frame kernels are VALU-issue bound and the descend / advance / pop paths are
// disjoint code that a divergent wave issues every iteration.  hipcc's structurizer merges the paths
// through ~20 register copies per iteration (v_mov phis) on top of ~80 instructions of arithmetic;
// written by hand every path updates the ray state in place under its own exec mask.
//
// Same arithmetic, same order, same rounding as ray_step (no FMA, IEEE mul / add / min):
//   header   c = p * t_coef - t_offset; tc = min3(c); child / span / LOD / leaf tests      (:74-95)
//   descend  push, h = tc, parent += offset + shift, half, child select, t_max = min(t_max, tc)  (:96-111)
//   advance  step mask, p -= step, t_min = tc, child ^= mask                                (:113-122)
//   pop      differing bits -> scale, stack read, p &= ~0 << scale, child bits, h = 0      (:124-145)
// Exits: LOD cut (kind 2), leaf (kind 1), root left (miss), iteration cap (miss).
//
// Register plan: 64-bit pairs are needed for v_pk_mul_f32 / v_pk_add_f32, ds_read_b64 / ds_write_b64
// and the node load, and inline asm cannot name the halves of a 64-bit operand, so the paired values
// live in fixed registers v32..v53; everything else is allocated by the compiler.
#pragma once

namespace vrc {

// The span test of :81 / :89, t_min <= min(t_max, tc) (no NaN in the walk).  Without a LOD test one min and one
// compare (a compare costs 1.7x a min); with it `t_min <= t_max` is needed on its own (:81 guards :82), so two compares.
#define VRC_WALK_SPAN_MIN                                                                       \
    "v_min_f32_e32 %[t2], v43, %[tc]\n"         /* :86 tv_max */                                \
    "v_cmp_le_f32_e32 vcc, v41, %[t2]\n"                                                        \
    "s_and_b64 %[sin], %[schild], vcc\n"
#define VRC_WALK_TMAX_MIN "v_mov_b32_e32 v43, %[t2]\n"
#define VRC_WALK_SPAN_CMP                                                                       \
    "v_cmp_le_f32_e32 vcc, v41, v43\n"          /* :81 t_min <= t_max */                        \
    "s_and_b64 %[stm], %[schild], vcc\n"                                                        \
    "v_cmp_le_f32_e32 vcc, v41, %[tc]\n"        /* with the line above: t_min <= min(t_max, tc), :86,:89 */ \
    "s_and_b64 %[sin], %[stm], vcc\n"
#define VRC_WALK_TMAX_CMP "v_min_f32_e32 v43, v43, %[tc]\n"

// HitPoint::complexity (:73).  Per lane (the per-ray operator returns it, the primary-hit capture records it) or, for the
// frame kernels, which only need the launch's total: the lanes at the loop top counted by the scalar unit (a lane's
// complexity is the number of loop tops it was live at) -- one VALU instruction per iteration less in an issue-bound loop.
#define VRC_WALK_COUNT_LANE "v_add_u32_e32 %[cmpl], 1, %[cmpl]\n"
#define VRC_WALK_COUNT_WAVE                                                                     \
    "s_bcnt1_i32_b64 %[sb], exec\n"                                                             \
    "s_add_u32 %[ssteps], %[ssteps], %[sb]\n"

// LOD_MODE: 0 = no LOD test (coef == bias == 0), 1 = bias == 0 (x + 0 does not change the comparison),
// 2 = general.
#define VRC_WALK_LOD_NONE ""
#define VRC_WALK_LOD_COEF                                                                       \
    "v_mul_f32_e32 %[t2], %[coef], %[tc]\n"                                                     \
    "v_cmp_ge_f32_e32 vcc, %[t2], v44\n"                                                         \
    "s_and_b64 %[st], %[stm], vcc\n"            /* child && t_min <= t_max && tc*coef >= scale_f */ \
    "s_or_b64 %[slod], %[slod], %[st]\n"                                                        \
    "s_andn2_b64 %[sin], %[sin], %[st]\n"                                                       \
    "s_andn2_b64 exec, exec, %[st]\n"
#define VRC_WALK_LOD_FULL                                                                       \
    "v_mul_f32_e32 %[t2], %[coef], %[tc]\n"                                                     \
    "v_add_f32_e32 %[t2], %[t2], %[bias]\n"                                                     \
    "v_cmp_ge_f32_e32 vcc, %[t2], v44\n"                                                         \
    "s_and_b64 %[st], %[stm], vcc\n"                                                            \
    "s_or_b64 %[slod], %[slod], %[st]\n"                                                        \
    "s_andn2_b64 %[sin], %[sin], %[st]\n"                                                       \
    "s_andn2_b64 exec, exec, %[st]\n"

// The x,y pairs of the walk's float arithmetic.  Round 3 (tools/micro/op_issue.hip, op_mix.hip): a v_pk_*_f32 occupies a whole
// issue slot of the SIMD (~4.7 cycles) and nothing co-issues with it, while a plain v_mul_f32 / v_add_f32 / v_sub_f32 co-issues
// beside another wave's "slow class" instruction (compare, min3, lshl_or, bfe ...), of which the walk has plenty: two plain
// operations cost less than the packed one.  -DVRC_WALK_PK restores the packed forms (A/B).
#ifdef VRC_WALK_PK
#define VRC_WALK_C_MUL_XY "v_pk_mul_f32 v[48:49], v[38:39], v[32:33]\n"
#define VRC_WALK_C_SUB_XY "v_pk_add_f32 v[48:49], v[48:49], v[36:37] neg_lo:[0,1] neg_hi:[0,1]\n"
#define VRC_WALK_H_MUL_XY "v_pk_mul_f32 v[50:51], v[32:33], v[44:45] op_sel_hi:[1,0]\n"
#define VRC_WALK_H_ADD_XY "v_pk_add_f32 v[50:51], v[50:51], v[48:49]\n"
#define VRC_WALK_P_SUB_XY "v_pk_add_f32 v[38:39], v[38:39], v[50:51] neg_lo:[0,1] neg_hi:[0,1]\n"
#define VRC_WALK_P_ADD_XY "v_pk_add_f32 v[48:49], v[38:39], v[50:51]\n"
#else
#define VRC_WALK_C_MUL_XY "v_mul_f32_e32 v48, v38, v32\n v_mul_f32_e32 v49, v39, v33\n"
#define VRC_WALK_C_SUB_XY "v_sub_f32_e32 v48, v48, v36\n v_sub_f32_e32 v49, v49, v37\n"
#define VRC_WALK_H_MUL_XY "v_mul_f32_e32 v50, v32, v44\n v_mul_f32_e32 v51, v33, v44\n"
#define VRC_WALK_H_ADD_XY "v_add_f32_e32 v50, v50, v48\n v_add_f32_e32 v51, v51, v49\n"
#define VRC_WALK_P_SUB_XY "v_sub_f32_e32 v38, v38, v50\n v_sub_f32_e32 v39, v39, v51\n"
#define VRC_WALK_P_ADD_XY "v_add_f32_e32 v48, v38, v50\n v_add_f32_e32 v49, v39, v51\n"
#endif

// v32:33 t_coef.xy  v34 t_coef.z  v35 t_offset.z  v36:37 t_offset.xy  v38:39 pos.xy  v40 pos.z
// v41 t_min  v42:43 {parent_id, t_max} (the stack entry)  v44 scale_f (v45 h)  v46:47 node
// v48:49 c.xy  v52 c.z  v50:51 / v53 per-axis step / half-point scratch
#define VRC_WALK_ASM(SPANPART, LODPART, TMAXPART, COUNTPART)                                                                \
    "s_mov_b64 %[sentry], exec\n"                                                               \
    "s_and_b64 exec, exec, %[sact]\n"           /* the lanes with a ray to walk */              \
    "s_mov_b64 %[sleaf], 0\n"                                                                   \
    "s_mov_b64 %[slod], 0\n"                                                                    \
    "s_mov_b32 %[siter], 0\n"                                                                   \
    "s_cbranch_execz Lwalk_done%=\n"                                                            \
    "Lwalk_loop%=:\n"                                                                           \
    "s_cmpk_gt_u32 %[siter], 0xffff\n"          /* safety net, MAX_ITERATIONS */                \
    "s_cbranch_scc1 Lwalk_done%=\n"                                                             \
    "s_add_u32 %[siter], %[siter], 1\n"                                                         \
    COUNTPART                                   /* :73 ++complexity */                          \
    "v_xor_b32_e32 %[cs], %[co], %[mirror]\n"   /* :79 child_shift */                           \
    VRC_WALK_C_MUL_XY                           /* :76 */                                       \
    "v_mul_f32_e32 v52, v40, v34\n"                                                             \
    "s_waitcnt vmcnt(0)\n"                      /* node of the current parent */                \
    "v_lshrrev_b32_e32 %[m], %[cs], v46\n"      /* bit 8: child, bit 16: leaf */                \
    VRC_WALK_C_SUB_XY                                                                           \
    "v_sub_f32_e32 v52, v52, v35\n"                                                             \
    "v_and_b32_e32 %[t], 0x10100, %[m]\n"       /* leaf_mask is a subset of child_mask (k_sanitize_nodes): 0, 0x100 or 0x10100 */ \
    "v_min3_f32 %[tc], v48, v49, v52\n"         /* :77 */                                       \
    "v_cmp_ne_u32_e64 %[schild], 0, %[t]\n"                                                     \
    SPANPART                                                                                    \
    LODPART                                                                                     \
    "v_cmp_lt_u32_e32 vcc, 0x100, %[t]\n"       /* child and leaf */                            \
    "s_and_b64 %[st], %[sin], vcc\n"            /* leaf hit :90-94 */                           \
    "s_or_b64 %[sleaf], %[sleaf], %[st]\n"                                                      \
    "s_andn2_b64 %[sin], %[sin], %[st]\n"       /* lanes that descend */                        \
    "s_andn2_b64 exec, exec, %[st]\n"                                                           \
    "s_cbranch_execz Lwalk_done%=\n"                                                            \
    "s_mov_b64 %[srun], exec\n"                                                                 \
    /* ---- descend :96-111 ---- */                                                             \
    "s_and_b64 exec, %[srun], %[sin]\n"                                                         \
    "s_cbranch_execz Lwalk_adv%=\n"                                                             \
    "v_cmp_lt_f32_e32 vcc, %[tc], v45\n"        /* :97 tc_max < h */                            \
    "v_mul_f32_e32 v44, 0.5, v44\n"             /* :87 half, becomes scale_f :105 */            \
    "s_and_saveexec_b64 %[st], vcc\n"                                                           \
    "v_lshl_add_u32 %[t], %[scale], 11, %[stkb]\n"                                              \
    "ds_write_b64 %[t], v[42:43]\n"             /* :98-99 */                                    \
    "s_mov_b64 exec, %[st]\n"                                                                   \
    "v_add3_u32 v42, v42, v47, %[cs]\n"         /* :103 */                                      \
    "v_lshlrev_b32_e32 %[t], 3, v42\n"                                                          \
    "global_load_dwordx2 v[46:47], %[t], %[nodes]\n"                                            \
    "v_mov_b32_e32 v45, %[tc]\n"                /* :101 */                                      \
    VRC_WALK_H_MUL_XY                           /* :88 half * t_coef + c */                     \
    "v_mul_f32_e32 v53, v34, v44\n"                                                             \
    VRC_WALK_H_ADD_XY                                                                           \
    "v_add_f32_e32 v53, v53, v52\n"                                                             \
    TMAXPART                                    /* :110 t_max = tv_max */                       \
    /* :107-109 without masks (anything touching VCC / an SGPR costs ~1.7x a plain VALU op): th > t_min is the \
       sign of t_min - th (never -0: t_min >= +0; never NaN: both finite), and pos + half sets bit scale-1 of   \
       the mantissa, which is 0 before (pos is a multiple of 2 * half in [1, 2)) */                          \
    "v_sub_f32_e32 v50, v41, v50\n"                                                             \
    "v_sub_f32_e32 v51, v41, v51\n"                                                             \
    "v_sub_f32_e32 v53, v41, v53\n"                                                             \
    "v_add_u32_e32 %[scale], -1, %[scale]\n"    /* :104 */                                      \
    "v_lshrrev_b32_e32 v50, 31, v50\n"                                                          \
    "v_lshrrev_b32_e32 v51, 31, v51\n"                                                          \
    "v_lshrrev_b32_e32 v53, 31, v53\n"                                                          \
    "v_lshl_or_b32 v38, v50, %[scale], v38\n"                                                   \
    "v_lshl_or_b32 v39, v51, %[scale], v39\n"                                                   \
    "v_lshl_or_b32 v40, v53, %[scale], v40\n"                                                   \
    "v_lshl_or_b32 %[co], v51, 1, v50\n"                                                        \
    "v_lshl_or_b32 %[co], v53, 2, %[co]\n"                                                      \
    /* ---- advance :113-122 ---- */                                                            \
    "Lwalk_adv%=:\n"                                                                            \
    "s_andn2_b64 exec, %[srun], %[sin]\n"                                                       \
    "s_cbranch_execz Lwalk_next%=\n"                                                            \
    "v_cmp_le_f32_e64 %[schild], v48, %[tc]\n"      /* :115-118 */                                  \
    "v_cmp_le_f32_e64 %[stm], v49, %[tc]\n"                                                      \
    "v_cmp_le_f32_e32 vcc, v52, %[tc]\n"                                                        \
    "v_mov_b32_e32 v41, %[tc]\n"                /* :120 */                                      \
    "v_cndmask_b32_e64 %[normal], 0, 1, vcc\n"                                                  \
    "v_cndmask_b32_e64 v50, 0, v44, %[schild]\n"                                                    \
    "v_addc_co_u32_e64 %[normal], %[st], %[normal], %[normal], %[stm]\n"                         \
    "v_cndmask_b32_e64 v51, 0, v44, %[stm]\n"                                                    \
    "v_cndmask_b32_e32 v53, 0, v44, vcc\n"                                                      \
    "v_addc_co_u32_e64 %[normal], %[st], %[normal], %[normal], %[schild]\n"   /* :122 */            \
    VRC_WALK_P_SUB_XY                                                                           \
    "v_sub_f32_e32 v40, v40, v53\n"                                                             \
    "v_xor_b32_e32 %[co], %[co], %[normal]\n"   /* :121 */                                      \
    "v_and_b32_e32 %[t], %[co], %[normal]\n"                                                    \
    "v_cmp_ne_u32_e32 vcc, 0, %[t]\n"           /* :124 */                                      \
    "s_and_b64 exec, exec, vcc\n"                                                               \
    "s_cbranch_execz Lwalk_next%=\n"                                                            \
    /* ---- pop :125-145 ---- */                                                                \
    VRC_WALK_P_ADD_XY                           /* pos before the step (exact) */               \
    "v_add_f32_e32 v52, v40, v53\n"                                                             \
    "v_xor_b32_e32 v48, v48, v38\n"                                                             \
    "v_xor_b32_e32 v49, v49, v39\n"                                                             \
    "v_xor_b32_e32 v52, v52, v40\n"                                                             \
    "v_or3_b32 %[t], v48, v49, v52\n"           /* differing_bits :125-131 */                   \
    "v_ffbh_u32_e32 %[t], %[t]\n"                                                               \
    "v_sub_u32_e32 %[scale], 31, %[t]\n"        /* :132; 0 bits -> 32 -> leaves below */        \
    "v_cmp_gt_u32_e32 vcc, 23, %[scale]\n"                                                      \
    "s_andn2_b64 %[st], exec, vcc\n"            /* left the root: miss */                       \
    "s_andn2_b64 %[srun], %[srun], %[st]\n"                                                     \
    "s_and_b64 exec, exec, vcc\n"                                                               \
    "s_cbranch_execz Lwalk_next%=\n"                                                            \
    "v_lshl_add_u32 %[t], %[scale], 11, %[stkb]\n"                                              \
    "ds_read_b64 v[42:43], %[t]\n"              /* :134-136 */                                  \
    "v_lshl_add_u32 v44, %[scale], 23, %[sfc]\n"    /* :133 2^(scale-23) */                     \
    "v_lshlrev_b32_e64 %[t2], %[scale], -1\n"                                                   \
    "v_bfe_u32 v48, v38, %[scale], 1\n"         /* :137-143 */                                  \
    "v_bfe_u32 v49, v39, %[scale], 1\n"                                                         \
    "v_bfe_u32 v52, v40, %[scale], 1\n"                                                         \
    "v_and_b32_e32 v38, v38, %[t2]\n"                                                           \
    "v_and_b32_e32 v39, v39, %[t2]\n"                                                           \
    "v_and_b32_e32 v40, v40, %[t2]\n"                                                           \
    "v_lshl_or_b32 %[co], v49, 1, v48\n"                                                        \
    "v_mov_b32_e32 v45, 0\n"                    /* :144 */                                      \
    "v_lshl_or_b32 %[co], v52, 2, %[co]\n"                                                      \
    "s_waitcnt lgkmcnt(0)\n"                                                                    \
    "v_lshlrev_b32_e32 %[t], 3, v42\n"                                                          \
    "global_load_dwordx2 v[46:47], %[t], %[nodes]\n"                                            \
    "Lwalk_next%=:\n"                                                                           \
    "s_mov_b64 exec, %[srun]\n"                                                                 \
    "s_cbranch_execnz Lwalk_loop%=\n"                                                           \
    "Lwalk_done%=:\n"                                                                           \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                           \
    "s_mov_b64 exec, %[sentry]\n"                                                               \
    "v_cndmask_b32_e64 %[kind], 0, 1, %[sleaf]\n"                                               \
    "v_cndmask_b32_e64 %[kind], %[kind], 2, %[slod]\n"

// Runs the walk for the lanes with `active` set (the others keep their state: kind 0, complexity 0).
// `stk` = this lane's column of the LDS stack, rows of 256 lanes (2 KiB apart).
// COUNT_LANE: r.complexity is maintained (per lane); otherwise it is left alone and the return value is the number of
// loop iterations all the wave's lanes executed in this call together (the same value in every lane that made the call).
template <int LOD_MODE, bool COUNT_LANE>
__device__ __forceinline__ uint32_t ray_walk(Ray& r, bool active, const uint2* __restrict__ nodes, uint2* stk,
                                             int depth, float coef, float bias)
{
    // (no branch around the loop for the lanes without a ray: the loop masks them itself, so that the SGPR results
    // are the same for every lane that made the call)
    const unsigned long long sact = __ballot(active);
    // LDS byte address of slot 0 minus the rows of the scales that have no slot (slot = scale - depth_off - 1)
    const uint32_t stkb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint2*)stk
                          - (uint32_t)(SVO_MAX - depth + 1) * 2048u;
    const uint32_t sfc = 104u << 23;            // (scale - 23 + 127) << 23 = (scale << 23) + (104 << 23)
    uint64_t sentry, sleaf, slod, srun, sin_, schild, stm, st;   // schild / stm double as the x / y step masks of the advance
    uint32_t siter, cs, m, t, t2, kind = 0u, ssteps = 0u, sb;
    float tc;
    float tcx = r.tcx, tcy = r.tcy, tcz = r.tcz, tox = r.tox, toy = r.toy, toz = r.toz;
    float px = r.px, py = r.py, pz = r.pz, t_min = r.t_min, t_max = r.t_max, sf = r.scale_f, h = r.h;
    uint32_t pid = r.parent_id, ndx = r.nd.x, ndy = r.nd.y;
    uint32_t scale = (uint32_t)r.scale, co = r.child_offset, normal = r.normal, cmpl = COUNT_LANE ? r.complexity : 0u;
#define VRC_WALK_OPERANDS(COUNT_OPERANDS)                                                                         \
        : [kind] "+v"(kind), COUNT_OPERANDS, [scale] "+v"(scale), [co] "+v"(co), [normal] "+v"(normal),              \
          "+{v38}"(px), "+{v39}"(py), "+{v40}"(pz), "+{v41}"(t_min), "+{v42}"(pid), "+{v43}"(t_max),                  \
          "+{v44}"(sf), "+{v45}"(h), "+{v46}"(ndx), "+{v47}"(ndy),                                                    \
          [sentry] "=&s"(sentry), [sleaf] "=&s"(sleaf), [slod] "=&s"(slod), [srun] "=&s"(srun), [sin] "=&s"(sin_),    \
          [schild] "=&s"(schild), [stm] "=&s"(stm), [st] "=&s"(st),                                                   \
          [siter] "=&s"(siter), [cs] "=&v"(cs), [m] "=&v"(m), [t] "=&v"(t), [t2] "=&v"(t2), [tc] "=&v"(tc)            \
        : "{v32}"(tcx), "{v33}"(tcy), "{v34}"(tcz), "{v35}"(toz), "{v36}"(tox), "{v37}"(toy),                         \
          [mirror] "v"(r.mirror), [stkb] "v"(stkb), [nodes] "s"(nodes), [sfc] "s"(sfc), [coef] "v"(coef), [bias] "v"(bias), \
          [sact] "s"(sact)                                                                                             \
        : "v48", "v49", "v50", "v51", "v52", "v53", "vcc", "scc", "memory"
#define VRC_WALK_OPS_LANE [cmpl] "+v"(cmpl)
#define VRC_WALK_OPS_WAVE [ssteps] "+s"(ssteps), [sb] "=&s"(sb)
    if (COUNT_LANE) {
        if (LOD_MODE == 0) asm volatile(VRC_WALK_ASM(VRC_WALK_SPAN_MIN, VRC_WALK_LOD_NONE, VRC_WALK_TMAX_MIN, VRC_WALK_COUNT_LANE) VRC_WALK_OPERANDS(VRC_WALK_OPS_LANE));
        else if (LOD_MODE == 1) asm volatile(VRC_WALK_ASM(VRC_WALK_SPAN_CMP, VRC_WALK_LOD_COEF, VRC_WALK_TMAX_CMP, VRC_WALK_COUNT_LANE) VRC_WALK_OPERANDS(VRC_WALK_OPS_LANE));
        else asm volatile(VRC_WALK_ASM(VRC_WALK_SPAN_CMP, VRC_WALK_LOD_FULL, VRC_WALK_TMAX_CMP, VRC_WALK_COUNT_LANE) VRC_WALK_OPERANDS(VRC_WALK_OPS_LANE));
    } else {
        if (LOD_MODE == 0) asm volatile(VRC_WALK_ASM(VRC_WALK_SPAN_MIN, VRC_WALK_LOD_NONE, VRC_WALK_TMAX_MIN, VRC_WALK_COUNT_WAVE) VRC_WALK_OPERANDS(VRC_WALK_OPS_WAVE));
        else if (LOD_MODE == 1) asm volatile(VRC_WALK_ASM(VRC_WALK_SPAN_CMP, VRC_WALK_LOD_COEF, VRC_WALK_TMAX_CMP, VRC_WALK_COUNT_WAVE) VRC_WALK_OPERANDS(VRC_WALK_OPS_WAVE));
        else asm volatile(VRC_WALK_ASM(VRC_WALK_SPAN_CMP, VRC_WALK_LOD_FULL, VRC_WALK_TMAX_CMP, VRC_WALK_COUNT_WAVE) VRC_WALK_OPERANDS(VRC_WALK_OPS_WAVE));
    }
#undef VRC_WALK_OPS_LANE
#undef VRC_WALK_OPS_WAVE
#undef VRC_WALK_OPERANDS
    r.px = px; r.py = py; r.pz = pz; r.t_min = t_min; r.t_max = t_max; r.scale_f = sf; r.h = h;
    r.parent_id = pid; r.nd = make_uint2(ndx, ndy);
    r.scale = (int)scale; r.child_offset = co; r.normal = normal;
    if (COUNT_LANE) r.complexity = cmpl;
    r.kind = kind;                              // 0 for a lane without a ray, as ray_setup left it
    r.shift = co ^ r.mirror;                    // child_shift of the child that was hit (unused on a miss)
    return ssteps;
}

}  // namespace vrc
