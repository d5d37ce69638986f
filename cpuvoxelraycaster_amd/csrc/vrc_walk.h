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
//   descend  push, parent += offset + shift, half, child select                            (:96-111)
//   advance  step mask, p -= step, t_min = tc, child ^= mask                                (:113-122)
//   pop      differing bits -> scale, stack read, p &= ~0 << scale, child bits              (:124-145)
// Exits: LOD cut (kind 2), leaf (kind 1), root left (miss), iteration cap (miss).
//
// What an instruction costs here (round 3, tools/micro/op_issue.hip, gen_op_mix.py; profiles/r03/op_*.txt).  A SIMD issues in
// slots of ~4.4-5.5 cycles, filled from DIFFERENT waves with either
//   two "simple" instructions       v_add/sub/mul/fma_f32, v_mov (class A); v_xor/and/or, v_add/sub_u32, v_lshrrev, v_ashrrev
//                                   (class B); simple ops with an SGPR operand: v_cndmask, v_addc (class S, not two of them),
//   one "slow" instruction + one A  every compare, v_min/max, v_min3, v_lshlrev (!), v_lshl_or / lshl_add / add3 / or3 / and_or /
//                                   bfe, v_ffbh, v_cvt: ~4.4 cycles alone, and nothing but a class-A instruction runs beside it,
//   or one packed / 24-bit-multiply / lane-access instruction (v_pk_*_f32, v_mul_u32_u24, v_readlane): nothing beside it.
// So: no packed math (two plain operations co-issue, a packed one does not: C3 frame 1.239 -> 1.136 ms), as few slow-class
// instructions as the algorithm allows (one slot each), float adds rather than shift-ors where the reference itself adds
// (pos += scale_f, :107-109), class-A instructions up to the number of slow ones are nearly free.  The scalar unit is the
// second resource (one per CU, ~1 instruction per cycle for four SIMDs): compares write EXEC directly (v_cmpx) instead of an
// SGPR pair and an s_and.
//
// Register plan: the node load returns a register pair and inline asm cannot name the halves of a 64-bit operand, so the
// node lives in fixed registers (v46:47); everything else is allocated by the compiler.  The stack entry is the parent index
// alone (ray_step in vrc_device.h says why t_max need not be kept): rows of 256 lanes x 4 bytes.  Seven scratch registers serve all four blocks (the kernels are at
// their register budget): `kind`, the result, is child_shift inside the loop; descend computes its half-points in m, t, t2
// once the header is done with them; the advance keeps its steps in kind, m, t2 for the pop.  Every read-write operand is early-clobber ("+&v"): without it the
// compiler may give an INPUT that happens to hold the same value (coef = 0.5 and the initial scale_f = 0.5) the same register,
// and the loop then changes the input under its own feet.
#pragma once

namespace vrc {

typedef uint32_t walk_rsrc_t __attribute__((ext_vector_type(4)));

// Buffer resource over the node array with stride 8 (one LNode): `buffer_load_dwordx2 ... idxen` then takes the node
// INDEX and the address arithmetic (index * 8) is done by the memory pipeline instead of a slow-class v_lshlrev_b32.
__device__ __forceinline__ walk_rsrc_t walk_node_rsrc(const uint2* __restrict__ nodes)
{
    const uint64_t p = (uint64_t)(uintptr_t)nodes;
    walk_rsrc_t r;
    r.x = (uint32_t)p;
    r.y = ((uint32_t)(p >> 32) & 0xffffu) | (8u << 16);     // base[47:32] | stride 8
    r.z = 0x20000000u;                                      // records (VRC_MAX_NODES = 2^29)
    r.w = 0x00020000u;                                      // untyped 32-bit data
    return r;
}

// HitPoint::complexity (:73).  Per lane (the per-ray operator returns it, the primary-hit capture records it) or, for the
// frame kernels, which only need the launch's total: the lanes at the loop top counted by the scalar unit (a lane's
// complexity is the number of loop tops it was live at) -- one VALU instruction per iteration less in an issue-bound loop.
#define VRC_WALK_COUNT_LANE "v_add_u32_e32 %[cmpl], 1, %[cmpl]\n"
#define VRC_WALK_COUNT_WAVE                                                                     \
    "s_bcnt1_i32_b64 %[sb], exec\n"                                                             \
    "s_add_u32 %[ssteps], %[ssteps], %[sb]\n"

// The tests of :81-89 that leave EXEC = the lanes whose child exists and whose span is not empty (LOD cuts taken out).
// Without a LOD test: t_min <= min(t_max, tc) in one min and one compare (:81 is implied, see ray_step).
#define VRC_WALK_SPAN_NOLOD                                                                     \
    "v_cmpx_ne_u32_e32 vcc, 0, %[t]\n"          /* :81 child exists */                          \
    "v_min_f32_e32 %[t2], 1.0, %[tc]\n"         /* :86 tv_max = min(t_max, tc) = min(1, tc) */  \
    "v_cmpx_le_f32_e32 vcc, %[tmin], %[t2]\n"   /* :81,:89 */
// With it (LOD_MODE 1: bias == 0, x + 0 does not change the comparison; 2: general): :81 guards :82, so t_min <= t_max is
// a compare of its own; with it, t_min <= tc is :89.
#define VRC_WALK_SPAN_LOD(BIASPART)                                                             \
    "v_cmpx_ne_u32_e32 vcc, 0, %[t]\n"          /* :81 child exists */                          \
    "v_cmpx_le_f32_e32 vcc, %[tmin], %[tmax]\n" /* :81 t_min <= t_max (= min(1, exit of the root)) */ \
    "v_mul_f32_e32 %[t2], %[coef], %[tc]\n"                                                     \
    BIASPART                                                                                    \
    "v_cmp_ge_f32_e32 vcc, %[t2], %[sf]\n"      /* :82 tc * coef + bias >= scale_f */           \
    "s_or_b64 %[slod], %[slod], vcc\n"                                                          \
    "s_andn2_b64 %[srun], %[srun], vcc\n"                                                       \
    "s_andn2_b64 exec, exec, vcc\n"                                                             \
    "v_cmpx_le_f32_e32 vcc, %[tmin], %[tc]\n"   /* :89 */
#define VRC_WALK_BIAS_NONE ""
#define VRC_WALK_BIAS_ADD "v_add_f32_e32 %[t2], %[t2], %[bias]\n"

#ifdef VRC_WALK_GLOBAL_LOAD   /* A/B: the byte offset computed by the VALU */
#define VRC_WALK_LOAD_NODE "v_lshlrev_b32_e32 %[m], 3, %[pid]\n global_load_dwordx2 v[46:47], %[m], %[nodes]\n"
#define VRC_WALK_NODES_OPERAND [nodes] "s"(nodes),
#else
#define VRC_WALK_LOAD_NODE "buffer_load_dwordx2 v[46:47], %[pid], %[rsrc], 0 idxen\n"
#define VRC_WALK_NODES_OPERAND [rsrc] "s"(rsrc),
#endif

// Start below the root (ray_walk's BELOW): the lanes whose start scale (in %[t]) is below 22 enter through the second half of the pop block with their start
// scale in %[t] and the (mirrored) ray origin in px, py, pz -- stack row -> parent, scale_f, pos = the origin's bits above the
// scale, child bits, node load: exactly the state a pop to that scale leaves (:133-143).  The other lanes start at the root.
#define VRC_WALK_START_ROOT ""
#define VRC_WALK_START_BELOW                                                                    \
    "v_cmpx_gt_u32_e32 vcc, 22, %[t]\n"                                                         \
    "s_cbranch_execnz Lwalk_start%=\n"                                                          \
    "s_mov_b64 exec, %[srun]\n"

// The rest of the header.  The node is only needed for the child / leaf bits; everything about the exits (c, tc) is not: with
// VRC_WALK_LATE_WAIT the wait for the node load comes after them (more of a pop's load latency hidden; A/B, round 5).
#ifdef VRC_WALK_LATE_WAIT
#define VRC_WALK_HEADER_REST                                                                    \
    "v_sub_f32_e32 %[cx], %[cx], %[tox]\n"                                                      \
    "v_sub_f32_e32 %[cy], %[cy], %[toy]\n"                                                      \
    "v_sub_f32_e32 %[cz], %[cz], %[toz]\n"                                                      \
    "v_min3_f32 %[tc], %[cx], %[cy], %[cz]\n"   /* :77 */                                       \
    "s_waitcnt vmcnt(0)\n"                      /* node of the current parent */                \
    "v_lshrrev_b32_e32 %[m], %[kind], v46\n"      /* bit 8: child, bit 16: leaf */                \
    "v_and_b32_e32 %[t], 0x10100, %[m]\n"       /* leaf_mask is a subset of child_mask (k_sanitize_nodes): 0, 0x100 or 0x10100 */
#else
#define VRC_WALK_HEADER_REST                                                                    \
    "s_waitcnt vmcnt(0)\n"                      /* node of the current parent */                \
    "v_lshrrev_b32_e32 %[m], %[kind], v46\n"      /* bit 8: child, bit 16: leaf */                \
    "v_sub_f32_e32 %[cx], %[cx], %[tox]\n"                                                      \
    "v_sub_f32_e32 %[cy], %[cy], %[toy]\n"                                                      \
    "v_sub_f32_e32 %[cz], %[cz], %[toz]\n"                                                      \
    "v_and_b32_e32 %[t], 0x10100, %[m]\n"       /* leaf_mask is a subset of child_mask (k_sanitize_nodes): 0, 0x100 or 0x10100 */ \
    "v_min3_f32 %[tc], %[cx], %[cy], %[cz]\n"   /* :77 */
#endif

// v46:47 node
#define VRC_WALK_ASM(SPANPART, COUNTPART, STARTPART, ROWSH, ROWDOWN)                                           \
    "s_mov_b64 %[sentry], exec\n"                                                               \
    "s_and_b64 exec, exec, %[sx]\n"             /* the lanes with a ray to walk (%[sx] arrives holding them) */ \
    "s_mov_b64 %[sleaf], 0\n"                                                                   \
    "s_mov_b64 %[slod], 0\n"                                                                    \
    "s_mov_b32 %[siter], 0xfffeffff\n"          /* carries out at the 65537th loop top: MAX_ITERATIONS */ \
    "s_mov_b64 %[srun], exec\n"                                                                 \
    "s_cbranch_execz Lwalk_done%=\n"                                                            \
    STARTPART                                                                                   \
    "Lwalk_loop%=:\n"                                                                           \
    "s_add_u32 %[siter], %[siter], 1\n"                                                         \
    "s_cbranch_scc1 Lwalk_done%=\n"             /* safety net */                                \
    COUNTPART                                   /* :73 ++complexity */                          \
    "v_xor_b32_e32 %[kind], %[co], %[mirror]\n"   /* :79 child_shift */                           \
    "v_mul_f32_e32 %[cx], %[px], %[tcx]\n"      /* :76 */                                       \
    "v_mul_f32_e32 %[cy], %[py], %[tcy]\n"                                                      \
    "v_mul_f32_e32 %[cz], %[pz], %[tcz]\n"                                                      \
    VRC_WALK_HEADER_REST                                                                        \
    SPANPART                                                                                    \
    "v_cmp_lt_u32_e32 vcc, 0x100, %[t]\n"       /* child and leaf, among the lanes inside: leaf hit :90-94 */ \
    "s_or_b64 %[sleaf], %[sleaf], vcc\n"                                                        \
    "s_andn2_b64 %[srun], %[srun], vcc\n"                                                       \
    "s_cbranch_scc0 Lwalk_done%=\n"             /* nobody left */                               \
    "s_andn2_b64 exec, exec, vcc\n"             /* lanes that descend */                        \
    "s_mov_b64 %[sin], exec\n"                                                                  \
    "s_cbranch_scc0 Lwalk_adv%=\n"                                                              \
    /* ---- descend :96-111 ---- */                                                             \
    /* :97-100 the push is unconditional: `tc_max < h` only skips writes of what the slot already holds (the level's    \
       parent does not change while the walk is below it) or of entries that are never read (last child on the ray); h  \
       is not kept at all */                                                                    \
    "ds_write_b32 %[sp], %[pid]\n"                                                              \
    "v_mul_f32_e32 %[sf], 0.5, %[sf]\n"         /* :87 half, becomes scale_f :105 */            \
    "v_add3_u32 %[pid], %[pid], v47, %[kind]\n"   /* :103 */                                      \
    VRC_WALK_LOAD_NODE                                                                          \
    /* h = half * t_coef + c (:88) in the header's scratch registers: m, t, t2 */               \
    "v_mul_f32_e32 %[m], %[tcx], %[sf]\n"                                                       \
    "v_mul_f32_e32 %[t], %[tcy], %[sf]\n"                                                       \
    "v_mul_f32_e32 %[t2], %[tcz], %[sf]\n"                                                      \
    "v_add_f32_e32 %[m], %[m], %[cx]\n"                                                         \
    "v_add_f32_e32 %[t], %[t], %[cy]\n"                                                         \
    "v_add_f32_e32 %[t2], %[t2], %[cz]\n"                                                       \
    /* :107-109 without compares: th > t_min is the sign of t_min - th (never -0: t_min >= +0; never NaN: both finite); \
       as a mask it gives the child bit and selects scale_f for the float add the reference does */                  \
    "v_sub_f32_e32 %[m], %[tmin], %[m]\n"                                                       \
    "v_sub_f32_e32 %[t], %[tmin], %[t]\n"                                                       \
    "v_sub_f32_e32 %[t2], %[tmin], %[t2]\n"                                                     \
    "v_add_u32_e32 %[sp], " ROWDOWN ", %[sp]\n"  /* :104 --scale: the next stack row down */    \
    "v_ashrrev_i32_e32 %[m], 31, %[m]\n"                                                        \
    "v_ashrrev_i32_e32 %[t], 31, %[t]\n"                                                        \
    "v_ashrrev_i32_e32 %[t2], 31, %[t2]\n"                                                      \
    "v_and_b32_e32 %[co], 1, %[m]\n"                                                            \
    "v_and_or_b32 %[co], %[t], 2, %[co]\n"                                                      \
    "v_and_or_b32 %[co], %[t2], 4, %[co]\n"                                                     \
    "v_and_b32_e32 %[m], %[sf], %[m]\n"                                                         \
    "v_and_b32_e32 %[t], %[sf], %[t]\n"                                                         \
    "v_and_b32_e32 %[t2], %[sf], %[t2]\n"                                                       \
    "v_add_f32_e32 %[px], %[px], %[m]\n"        /* exact: pos is a multiple of 2 * half */      \
    "v_add_f32_e32 %[py], %[py], %[t]\n"                                                        \
    "v_add_f32_e32 %[pz], %[pz], %[t2]\n"                                                       \
    /* ---- advance :113-122 ---- */                                                            \
    "Lwalk_adv%=:\n"                                                                            \
    "s_andn2_b64 exec, %[srun], %[sin]\n"                                                       \
    "s_cbranch_scc0 Lwalk_next%=\n"                                                             \
    "v_cmp_le_f32_e64 %[sx], %[cx], %[tc]\n"    /* :115-118 */                                  \
    "v_cmp_le_f32_e64 %[sy], %[cy], %[tc]\n"                                                    \
    "v_cmp_le_f32_e32 vcc, %[cz], %[tc]\n"                                                      \
    "v_mov_b32_e32 %[tmin], %[tc]\n"            /* :120 */                                      \
    "v_cndmask_b32_e64 %[normal], 0, 1, vcc\n"                                                  \
    "v_cndmask_b32_e64 %[kind], 0, %[sf], %[sx]\n"                                                \
    "v_addc_co_u32_e64 %[normal], %[sin], %[normal], %[normal], %[sy]\n"   /* (carry out: into the dead sin) */ \
    "v_cndmask_b32_e64 %[m], 0, %[sf], %[sy]\n"                                                \
    "v_cndmask_b32_e32 %[t2], 0, %[sf], vcc\n"                                                  \
    "v_addc_co_u32_e64 %[normal], %[sin], %[normal], %[normal], %[sx]\n"  /* :122 */            \
    "v_sub_f32_e32 %[px], %[px], %[kind]\n"                                                       \
    "v_sub_f32_e32 %[py], %[py], %[m]\n"                                                       \
    "v_sub_f32_e32 %[pz], %[pz], %[t2]\n"                                                       \
    "v_xor_b32_e32 %[co], %[co], %[normal]\n"   /* :121 */                                      \
    "v_and_b32_e32 %[t], %[co], %[normal]\n"                                                    \
    "v_cmpx_ne_u32_e32 vcc, 0, %[t]\n"          /* :124 */                                      \
    "s_cbranch_execz Lwalk_next%=\n"                                                            \
    /* ---- pop :125-145 ---- */                                                                \
    "v_add_f32_e32 %[cx], %[px], %[kind]\n"       /* pos before the step (exact) */               \
    "v_add_f32_e32 %[cy], %[py], %[m]\n"                                                       \
    "v_add_f32_e32 %[cz], %[pz], %[t2]\n"                                                       \
    "v_xor_b32_e32 %[cx], %[cx], %[px]\n"                                                       \
    "v_xor_b32_e32 %[cy], %[cy], %[py]\n"                                                       \
    "v_xor_b32_e32 %[cz], %[cz], %[pz]\n"                                                       \
    "v_or3_b32 %[t], %[cx], %[cy], %[cz]\n"     /* differing_bits :125-131 */                   \
    "v_ffbh_u32_e32 %[t], %[t]\n"                                                               \
    "v_sub_u32_e32 %[t], 31, %[t]\n"            /* :132 scale; 0 bits -> 32 -> leaves below */  \
    "v_cmp_le_u32_e32 vcc, 23, %[t]\n"          /* left the root: miss */                       \
    "s_andn2_b64 %[srun], %[srun], vcc\n"                                                       \
    "s_andn2_b64 exec, exec, vcc\n"                                                             \
    "s_cbranch_scc0 Lwalk_next%=\n"                                                             \
    "Lwalk_start%=:\n"                                                                          \
    "v_lshl_add_u32 %[sp], %[t], " ROWSH ", %[stkb]\n"                                                 \
    "ds_read_b32 %[pid], %[sp]\n"               /* :134-136 */                                  \
    "v_lshl_add_u32 %[sf], %[t], 23, %[sfc]\n"  /* :133 2^(scale-23) */                         \
    "v_lshlrev_b32_e64 %[t2], %[t], -1\n"       /* (the steps in cs, m, t2 are dead by now) */  \
    "v_bfe_u32 %[cx], %[px], %[t], 1\n"         /* :137-143 */                                  \
    "v_bfe_u32 %[cy], %[py], %[t], 1\n"                                                         \
    "v_bfe_u32 %[cz], %[pz], %[t], 1\n"                                                         \
    "v_and_b32_e32 %[px], %[px], %[t2]\n"                                                       \
    "v_and_b32_e32 %[py], %[py], %[t2]\n"                                                       \
    "v_and_b32_e32 %[pz], %[pz], %[t2]\n"                                                       \
    "v_lshl_or_b32 %[co], %[cy], 1, %[cx]\n"                                                    \
    "v_lshl_or_b32 %[co], %[cz], 2, %[co]\n"                                                    \
    "s_waitcnt lgkmcnt(0)\n"                                                                    \
    VRC_WALK_LOAD_NODE                                                                          \
    "Lwalk_next%=:\n"                                                                           \
    "s_mov_b64 exec, %[srun]\n"                                                                 \
    "s_cbranch_execnz Lwalk_loop%=\n"                                                           \
    "Lwalk_done%=:\n"                                                                           \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                           \
    "s_or_b64 exec, %[sleaf], %[slod]\n"                                                        \
    "ds_write_b32 %[sp], %[pid]\n"              /* the parent of the hit in the row of its scale: with the rows above it, \
                                                   the path from the root (a ray that starts next to the hit reads it) */ \
    "s_mov_b64 exec, %[sentry]\n"                                                               \
    "v_cndmask_b32_e64 %[kind], 0, 1, %[sleaf]\n"                                               \
    "v_cndmask_b32_e64 %[kind], %[kind], 2, %[slod]\n"

// Runs the walk for the lanes with `active` set (the others keep their state: kind 0, complexity 0).
// `stk` = this lane's column of the LDS stack, `depth` rows of 256 lanes, 1 KiB apart, or 2 KiB (TWO_STACKS: the frame
// kernels keep two stacks, their rows interleaved, so that one is the other plus a constant: copy_stack).
// COUNT_LANE: r.complexity is maintained (per lane); otherwise it is left alone and the return value is the number of
// loop iterations all the wave's lanes executed in this call together (the same value in every lane that made the call).
// r.h is not maintained (see the push above); nothing reads it after a walk.
// BELOW: lanes whose r.scale is below 22 (ray_start_below in vrc_device.h set it, and r.px, r.py, r.pz to the ray's origin)
// start at that scale instead of at the root -- see start_scale_next_to for when that is the walk the reference does; the
// iterations such a lane does not execute (22 - scale) are NOT in the return value, the caller books them (COUNT_LANE: the
// caller presets r.complexity with them).
template <int LOD_MODE, bool COUNT_LANE, bool BELOW = false, bool TWO_STACKS = false>
__device__ __forceinline__ uint32_t ray_walk(Ray& r, bool active, const uint2* __restrict__ nodes, stack_t* stk,
                                             int depth, float coef, float bias)
{
    // (no branch around the loop for the lanes without a ray: the loop masks them itself, so that the SGPR results
    // are the same for every lane that made the call)
    const unsigned long long sact = __ballot(active);
    // LDS byte address of slot 0 minus the rows of the scales that have no slot (slot = scale - depth_off)
    const uint32_t stkb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) stack_t*)stk
                          - (uint32_t)(SVO_MAX - depth) * (TWO_STACKS ? 2048u : 1024u);
    const uint32_t sfc = 104u << 23;            // (scale - 23 + 127) << 23 = (scale << 23) + (104 << 23)
    const walk_rsrc_t rsrc = walk_node_rsrc(nodes);
    uint64_t sentry, sleaf, slod, srun, sin_, sy;
    uint64_t sx = sact;                         // the lanes to walk on entry, then the x step mask of the advance (sy: the y mask)
    uint32_t siter, m, t = (uint32_t)r.scale, t2, kind, ssteps = 0u, sb;
    float tc, cx, cy, cz;
    float px = r.px, py = r.py, pz = r.pz, t_min = r.t_min, sf = r.scale_f;
    uint32_t pid = r.parent_id, ndx = r.nd.x, ndy = r.nd.y;
    uint32_t sp = stkb + ((uint32_t)r.scale << (TWO_STACKS ? 11 : 10));   // the level, as the LDS address of its stack row
    uint32_t co = r.child_offset, normal = r.normal, cmpl = COUNT_LANE ? r.complexity : 0u;
#define VRC_WALK_OPERANDS(...)                                                                                    \
        : [kind] "=&v"(kind), __VA_ARGS__, [sp] "+&v"(sp), [co] "+&v"(co), [normal] "+&v"(normal),                     \
          [px] "+&v"(px), [py] "+&v"(py), [pz] "+&v"(pz), [tmin] "+&v"(t_min), [pid] "+&v"(pid),                          \
          [sf] "+&v"(sf), "+{v46}"(ndx), "+{v47}"(ndy),                                                                \
          [sentry] "=&s"(sentry), [sleaf] "=&s"(sleaf), [slod] "=&s"(slod), [srun] "=&s"(srun), [sin] "=&s"(sin_),    \
          [sx] "+&s"(sx), [sy] "=&s"(sy),                                                                             \
          [siter] "=&s"(siter), [m] "=&v"(m), [t] "+&v"(t), [t2] "=&v"(t2), [tc] "=&v"(tc),                           \
          [cx] "=&v"(cx), [cy] "=&v"(cy), [cz] "=&v"(cz)                                                              \
        : [tcx] "v"(r.tcx), [tcy] "v"(r.tcy), [tcz] "v"(r.tcz), [tox] "v"(r.tox), [toy] "v"(r.toy), [toz] "v"(r.toz), \
          [mirror] "v"(r.mirror), [stkb] "v"(stkb), VRC_WALK_NODES_OPERAND [sfc] "s"(sfc), [coef] "v"(coef), [bias] "v"(bias), [tmax] "v"(r.t_max) \
        : "vcc", "scc", "memory"
#define VRC_WALK_OPS_LANE [cmpl] "+&v"(cmpl)
#define VRC_WALK_OPS_WAVE [ssteps] "+&s"(ssteps), [sb] "=&s"(sb)
#define VRC_WALK_RUN(SPAN, COUNT, OPS)                                                                              \
    do {                                                                                                            \
        if (BELOW && TWO_STACKS) asm volatile(VRC_WALK_ASM(SPAN, COUNT, VRC_WALK_START_BELOW, "11", "0xfffff800") VRC_WALK_OPERANDS(OPS)); \
        else if (TWO_STACKS) asm volatile(VRC_WALK_ASM(SPAN, COUNT, VRC_WALK_START_ROOT, "11", "0xfffff800") VRC_WALK_OPERANDS(OPS)); \
        else if (BELOW) asm volatile(VRC_WALK_ASM(SPAN, COUNT, VRC_WALK_START_BELOW, "10", "0xfffffc00") VRC_WALK_OPERANDS(OPS));      \
        else asm volatile(VRC_WALK_ASM(SPAN, COUNT, VRC_WALK_START_ROOT, "10", "0xfffffc00") VRC_WALK_OPERANDS(OPS));                  \
    } while (0)
    if (COUNT_LANE) {
        if (LOD_MODE == 0) VRC_WALK_RUN(VRC_WALK_SPAN_NOLOD, VRC_WALK_COUNT_LANE, VRC_WALK_OPS_LANE);
        else if (LOD_MODE == 1) VRC_WALK_RUN(VRC_WALK_SPAN_LOD(VRC_WALK_BIAS_NONE), VRC_WALK_COUNT_LANE, VRC_WALK_OPS_LANE);
        else VRC_WALK_RUN(VRC_WALK_SPAN_LOD(VRC_WALK_BIAS_ADD), VRC_WALK_COUNT_LANE, VRC_WALK_OPS_LANE);
    } else {
        if (LOD_MODE == 0) VRC_WALK_RUN(VRC_WALK_SPAN_NOLOD, VRC_WALK_COUNT_WAVE, VRC_WALK_OPS_WAVE);
        else if (LOD_MODE == 1) VRC_WALK_RUN(VRC_WALK_SPAN_LOD(VRC_WALK_BIAS_NONE), VRC_WALK_COUNT_WAVE, VRC_WALK_OPS_WAVE);
        else VRC_WALK_RUN(VRC_WALK_SPAN_LOD(VRC_WALK_BIAS_ADD), VRC_WALK_COUNT_WAVE, VRC_WALK_OPS_WAVE);
    }
#undef VRC_WALK_RUN
#undef VRC_WALK_OPS_LANE
#undef VRC_WALK_OPS_WAVE
#undef VRC_WALK_OPERANDS
    r.px = px; r.py = py; r.pz = pz; r.t_min = t_min; r.scale_f = sf;
    r.parent_id = pid; r.nd = make_uint2(ndx, ndy);
    r.scale = (int)((sp - stkb) >> (TWO_STACKS ? 11 : 10)); r.child_offset = co; r.normal = normal;
    if (COUNT_LANE) r.complexity = cmpl;
    r.kind = kind;                              // 0 for a lane without a ray, as ray_setup left it
    r.shift = co ^ r.mirror;                    // child_shift of the child that was hit (unused on a miss)
    return ssteps;
}

}  // namespace vrc
