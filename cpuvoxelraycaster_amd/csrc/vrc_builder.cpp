// vrc_builder.cpp -- host-side LSVO construction: emits the
// exact array compileSVO produces (src/lsvo_utils.cpp:4-49 over the tree that
// SVO::setCell builds, include/svo.hpp:72-114) straight from an occupancy
// description, with no pointer octree in between.
//
// Layout rules reproduced (lsvo_utils.cpp):
//   * data[0] = root; visiting a node appends its 8 child slots at the current
//     end of the array and stores child_offset = first_slot - own_index (:8-10,:25-27);
//   * children are visited x-outer, y-middle, z-inner (:29-31) and written to
//     slot z*4 + y*2 + x (:34); unit voxels set leaf_mask (:40-42), larger
//     children recurse immediately (depth-first pre-order, :37-39);
//   * every slot starts as LNode() = {color 1, masks 0, offset 0} (lsvo_utils.hpp:7-12).
// A child exists iff its cube holds at least one solid voxel (rec_setCell always
// descends to size 1 and never merges, svo.hpp:97-113).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/vrc.h"

namespace {

struct Emitter {
    std::vector<vrc_lnode> data;
    void push8()
    {
        const vrc_lnode blank = {1u, 0u, 0u, 0u, 0u};
        data.insert(data.end(), 8, blank);
    }
};

// Occ: bool exists(x0, y0, z0, size) -- any solid voxel in the cube?
template <class Occ>
void build_rec(Emitter& e, const Occ& occ, uint32_t node_index, uint32_t x0, uint32_t y0, uint32_t z0, uint32_t size)
{
    const uint32_t child_pos = (uint32_t)e.data.size();
    e.data[node_index].child_offset = child_pos - node_index;
    const uint32_t half = size / 2;
    uint8_t present = 0;  // bit (x*4 + y*2 + z) in visiting order
    for (uint32_t x = 0; x < 2; ++x)
        for (uint32_t y = 0; y < 2; ++y)
            for (uint32_t z = 0; z < 2; ++z)
                if (occ(x0 + x * half, y0 + y * half, z0 + z * half, half)) present |= (uint8_t)(1u << (x * 4 + y * 2 + z));
    if (!present) return;  // lsvo_utils.cpp:12-23: an empty node gets no slots
    e.push8();
    for (uint32_t x = 0; x < 2; ++x)
        for (uint32_t y = 0; y < 2; ++y)
            for (uint32_t z = 0; z < 2; ++z) {
                if (!(present & (1u << (x * 4 + y * 2 + z)))) continue;
                const uint32_t sub_index = z * 4 + y * 2 + x;
                e.data[node_index].child_mask |= (uint8_t)(1u << sub_index);
                if (half == 1)
                    e.data[node_index].leaf_mask |= (uint8_t)(1u << sub_index);
                else
                    build_rec(e, occ, child_pos + sub_index, x0 + x * half, y0 + y * half, z0 + z * half, half);
            }
}

int finish(Emitter& e, vrc_lnode** out, uint64_t* n_nodes)
{
    const size_t bytes = e.data.size() * sizeof(vrc_lnode);
    void* p = std::malloc(bytes ? bytes : 1);
    if (!p) return VRC_ERR_OOM;
    std::memcpy(p, e.data.data(), bytes);
    *out = (vrc_lnode*)p;
    *n_nodes = e.data.size();
    return VRC_OK;
}

}  // namespace

extern "C" int vrc_build_terrain_lsvo(const int32_t* height, uint32_t depth, vrc_lnode** out, uint64_t* n_nodes)
{
    if (!height || !out || !n_nodes || depth < 1 || depth > VRC_MAX_DEPTH) return VRC_ERR_INVALID;
    try {
        const uint32_t S = 1u << depth;
        // main.cpp:65-72: column (x,z) is solid for y in [S/2 + 1, S/2 + lim), lim = max(16, min(S, height)).
        // Max-pyramid of lim over square footprints: level l holds (S >> l)^2 entries of 2^l x 2^l columns.
        std::vector<std::vector<int32_t>> pyr(depth + 1);
        pyr[0].resize((size_t)S * S);
        for (uint32_t x = 0; x < S; ++x)
            for (uint32_t z = 0; z < S; ++z) {
                int32_t lim = height[(size_t)x * S + z];
                if (lim > (int32_t)S) lim = (int32_t)S;
                if (lim < 16) lim = 16;
                pyr[0][(size_t)x * S + z] = lim;
            }
        for (uint32_t l = 1; l <= depth; ++l) {
            const uint32_t n = S >> l, m = S >> (l - 1);
            pyr[l].resize((size_t)n * n);
            for (uint32_t x = 0; x < n; ++x)
                for (uint32_t z = 0; z < n; ++z) {
                    const int32_t a = pyr[l - 1][(size_t)(2 * x) * m + 2 * z], b = pyr[l - 1][(size_t)(2 * x) * m + 2 * z + 1];
                    const int32_t c = pyr[l - 1][(size_t)(2 * x + 1) * m + 2 * z], d = pyr[l - 1][(size_t)(2 * x + 1) * m + 2 * z + 1];
                    int32_t mx = a > b ? a : b;
                    if (c > mx) mx = c;
                    if (d > mx) mx = d;
                    pyr[l][(size_t)x * n + z] = mx;
                }
        }
        const int64_t base = (int64_t)(S / 2);
        auto occ = [&](uint32_t x0, uint32_t y0, uint32_t z0, uint32_t size) -> bool {
            uint32_t l = 0;
            while ((1u << l) < size) ++l;
            const uint32_t n = S >> l;
            const int64_t lim = pyr[l][(size_t)(x0 >> l) * n + (z0 >> l)];
            // solid y range of the tallest column: [base + 1, base + lim - 1]
            return (int64_t)y0 <= base + lim - 1 && (int64_t)y0 + size - 1 >= base + 1;
        };
        Emitter e;
        const vrc_lnode blank = {1u, 0u, 0u, 0u, 0u};
        e.data.push_back(blank);
        build_rec(e, occ, 0u, 0u, 0u, 0u, S);
        return finish(e, out, n_nodes);
    } catch (const std::bad_alloc&) {
        return VRC_ERR_OOM;
    }
}

extern "C" int vrc_build_volume_lsvo(const uint8_t* solid, uint32_t depth, vrc_lnode** out, uint64_t* n_nodes)
{
    if (!solid || !out || !n_nodes || depth < 1 || depth > 9) return VRC_ERR_INVALID;  // dense input: 512^3 bytes max
    try {
        const uint32_t S = 1u << depth;
        // OR-pyramid: level l has (S >> l)^3 bytes
        std::vector<std::vector<uint8_t>> pyr(depth + 1);
        pyr[0].assign(solid, solid + (size_t)S * S * S);
        for (auto& v : pyr[0]) v = v ? 1 : 0;
        for (uint32_t l = 1; l <= depth; ++l) {
            const size_t n = S >> l, m = S >> (l - 1);
            pyr[l].resize(n * n * n);
            for (size_t x = 0; x < n; ++x)
                for (size_t y = 0; y < n; ++y)
                    for (size_t z = 0; z < n; ++z) {
                        uint8_t any = 0;
                        for (size_t k = 0; k < 8; ++k)
                            any |= pyr[l - 1][((2 * x + (k >> 2)) * m + (2 * y + ((k >> 1) & 1))) * m + (2 * z + (k & 1))];
                        pyr[l][(x * n + y) * n + z] = any;
                    }
        }
        auto occ = [&](uint32_t x0, uint32_t y0, uint32_t z0, uint32_t size) -> bool {
            uint32_t l = 0;
            while ((1u << l) < size) ++l;
            const size_t n = S >> l;
            return pyr[l][(((size_t)x0 >> l) * n + (y0 >> l)) * n + (z0 >> l)] != 0;
        };
        Emitter e;
        const vrc_lnode blank = {1u, 0u, 0u, 0u, 0u};
        e.data.push_back(blank);
        build_rec(e, occ, 0u, 0u, 0u, 0u, S);
        return finish(e, out, n_nodes);
    } catch (const std::bad_alloc&) {
        return VRC_ERR_OOM;
    }
}

extern "C" void vrc_free_host(void* p) { std::free(p); }
