// vrc_present.hip -- SURVEY 8f row N2: the frame post-process / present chain that follows the hot path
// (src/main.cpp:160-182), kept on the device so that the frame never crosses PCIe between the renderer and
// whatever shows it:
//
//   render_tex   = render_image, then x cache2  (sf::BlendMultiply, Color(255 * (1 - old)))      main.cpp:163-172
//   denoised_tex = denoised_tex x cache1 (BlendMultiply, Color(255 * old)) + render_tex (BlendAdd)  :161-162,175-177
//   window       = denoised_tex scaled by 1 / render_scale, nearest sampling                      :179-182
//   optional per-channel median around the sampled texel: the min/max exchange networks of the reference's (unused)
//   res/median_3.frag (3x3) and res/median.frag (5x5)
//
// Two HBM-bound byte kernels (no MFMA): blend = 8 B read + 4 B written per render pixel, window = 4 B written per
// window pixel (taps come from L1 / L2).  Integer definition of the GL blending and the stated +-1 LSB tolerance
// against a real GL device: oracle/vrc_oracle.c (unorm8_mul); this file equals the oracle bit for bit.
//
// The exchange lists below are those of the two shaders, which carry this notice:
//   Copyright (c) Morgan McGuire and Williams College, 2006.  All rights reserved.
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the
//   following conditions are met: Redistributions of source code must retain the above copyright notice, this list of
//   conditions and the following disclaimer.  Redistributions in binary form must reproduce the above copyright
//   notice, this list of conditions and the following disclaimer in the documentation and/or other materials provided
//   with the distribution.  THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY EXPRESS
//   OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A
//   PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT SHALL THE COPYRIGHT HOLDER OR CONTRIBUTORS BE LIABLE FOR ANY
//   DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES (INCLUDING, BUT NOT LIMITED TO,
//   PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES; LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER
//   CAUSED AND ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT (INCLUDING NEGLIGENCE OR
//   OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
// (Note: the 5x5 list as shipped in the reference is shorter than a complete median-of-25 network; it returns an
// element of rank 9..14 -- the exact median for ~73 % of random windows.  It is restated as shipped.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>

#include "../../include/vrc.h"
#include "vrc_internal.h"

struct vrc_presenter {
    int device = 0;
    uint32_t width = 0, height = 0, out_width = 0, out_height = 0;
    uint32_t* d_denoised = nullptr;   // denoised_tex, RGBA8 width x height, persistent across frames
    uint32_t* d_window = nullptr;     // what window.draw(final_sprite) produces, RGBA8 out_width x out_height
};

namespace {

#define HIP_TRYP(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return vrc::fail(e_ == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// round(a * b / 255): a UNORM8 product stored to a UNORM8 target (255 is odd: no ties)
__device__ __forceinline__ uint32_t unorm8_mul(uint32_t a, uint32_t b) { return (a * b + 127u) / 255u; }

__global__ void __launch_bounds__(256)
k_present_blend(const uint32_t* __restrict__ image, uint32_t* __restrict__ denoised, uint32_t n, uint32_t c1, uint32_t c2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = image[i], old = denoised[i];
    uint32_t out = 0xff000000u;
    for (uint32_t sh = 0; sh < 24u; sh += 8u) {
        const uint32_t r = unorm8_mul((src >> sh) & 0xffu, c2);      // render_tex: sprite, then BlendMultiply with cache2
        const uint32_t d = unorm8_mul((old >> sh) & 0xffu, c1);      // denoised_tex x cache1
        out |= min(255u, d + r) << sh;                               // BlendAdd, clamped by the UNORM8 target
    }
    denoised[i] = out;
}

// s2(a, b) of the shaders: a = min, b = max
__device__ __forceinline__ void s2(uint32_t& a, uint32_t& b)
{
    const uint32_t lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}

// res/median_3.frag:57-82 -- w in fetch order (-1,-1) (0,-1) (+1,-1) (-1,0) (0,0) (+1,0) (-1,+1) (0,+1) (+1,+1)
__device__ __forceinline__ uint32_t median9(const uint32_t w[9])
{
    uint32_t v0 = w[0], v1 = w[1], v2 = w[2], v3 = w[3], v4 = w[4], v5 = w[5];
    s2(v0, v3); s2(v1, v4); s2(v2, v5); s2(v0, v1); s2(v0, v2); s2(v4, v5); s2(v3, v5);   // mnmx6
    v5 = w[6];
    s2(v1, v2); s2(v3, v4); s2(v1, v3); s2(v1, v5); s2(v4, v5); s2(v2, v5);               // mnmx5
    v5 = w[7];
    s2(v2, v3); s2(v4, v5); s2(v2, v4); s2(v3, v5);                                       // mnmx4
    v5 = w[8];
    s2(v4, v5); s2(v3, v5); s2(v3, v4);                                                   // mnmx3
    return v4;
}

// res/median.frag:68-86 -- the exchange list as shipped (94 exchanges), v[(dX + 2) * 5 + (dY + 2)]
#define VRC_M25(X) \
    X(0,1) X(3,4) X(2,4) X(2,3) X(6,7) X(5,7) X(5,6) X(9,7) X(1,7) X(1,4) X(12,13) X(11,13) X(11,12) X(15,16) X(14,16) \
    X(14,15) X(18,19) X(17,19) X(17,18) X(21,22) X(20,22) X(20,21) X(23,24) X(2,5) X(3,6) X(0,6) X(0,3) X(4,7) X(1,7) X(1,4) \
    X(11,14) X(8,14) X(8,11) X(12,15) X(9,15) X(9,12) X(13,16) X(10,16) X(10,13) X(20,23) X(17,23) X(17,20) X(21,24) X(18,24) X(18,21) \
    X(19,22) X(8,17) X(9,18) X(0,18) X(0,9) X(10,19) X(1,19) X(1,10) X(11,20) X(2,20) X(2,11) X(12,21) X(3,21) X(3,12) X(13,22) \
    X(4,22) X(4,13) X(14,23) X(5,23) X(5,14) X(15,24) X(6,24) X(6,15) X(7,16) X(7,19) X(3,11) X(5,17) X(11,17) X(9,17) X(4,10) \
    X(6,12) X(7,14) X(4,6) X(4,7) X(12,14) X(10,14) X(6,7) X(10,12) X(6,10) X(6,17) X(12,17) X(7,17) X(7,10) X(12,18) X(7,12) \
    X(10,18) X(12,20) X(10,20) X(10,12)

__device__ __forceinline__ uint32_t median25(uint32_t v[25])
{
#define VRC_M25_S2(a, b) s2(v[a], v[b]);
    VRC_M25(VRC_M25_S2)
#undef VRC_M25_S2
    return v[12];
}

template <int MEDIAN>
__global__ void __launch_bounds__(256)
k_present_window(const uint32_t* __restrict__ denoised, uint32_t W, uint32_t H, uint32_t* __restrict__ window, uint32_t OW, uint32_t OH)
{
    const uint32_t X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= OW) return;
    // nearest sampling: the texel that contains the pixel centre, (X + 0.5) * W / OW
    const uint32_t sx = (uint32_t)(((uint64_t)(2u * X + 1u) * W) / (2ull * OW));
    const uint32_t sy = (uint32_t)(((uint64_t)(2u * Y + 1u) * H) / (2ull * OH));
    uint32_t out;
    if (MEDIAN == 0) {
        out = denoised[(uint64_t)sy * W + sx];
    } else {
        constexpr int R = MEDIAN / 2, N = MEDIAN ? MEDIAN * MEDIAN : 1;
        uint32_t px[N];
        // taps one texel apart, clamped to the edge; 3x3 in the fetch order of median_3.frag (rows outer), 5x5 stored
        // column-major like median.frag (v[(dX + R) * 5 + (dY + R)])
        for (int a = -R; a <= R; ++a)
            for (int b = -R; b <= R; ++b) {
                const int dx = MEDIAN == 3 ? b : a, dy = MEDIAN == 3 ? a : b;
                const uint32_t tx = (uint32_t)min(max((int)sx + dx, 0), (int)W - 1), ty = (uint32_t)min(max((int)sy + dy, 0), (int)H - 1);
                px[(a + R) * MEDIAN + (b + R)] = denoised[(uint64_t)ty * W + tx];
            }
        out = 0u;
        for (uint32_t sh = 0; sh < 24u; sh += 8u) {
            uint32_t v[N];
            for (int k = 0; k < N; ++k) v[k] = (px[k] >> sh) & 0xffu;
            out |= (MEDIAN == 3 ? median9(v) : median25(v)) << sh;
        }
    }
    window[(uint64_t)Y * OW + X] = out | 0xff000000u;
}

}  // namespace

extern "C" int vrc_presenter_create(int device, uint32_t width, uint32_t height, uint32_t out_width, uint32_t out_height, vrc_presenter** out)
{
    if (!out || !width || !height || !out_width || !out_height || (uint64_t)width * height > 0x7fffffffull ||
        (uint64_t)out_width * out_height > 0x7fffffffull || out_height > 65535u)
        return vrc::fail(VRC_ERR_INVALID, "vrc_presenter_create: bad argument");
    int rc = vrc::require_device(device, nullptr);
    if (rc) return rc;
    vrc_presenter* p = new (std::nothrow) vrc_presenter();
    if (!p) return vrc::fail(VRC_ERR_OOM, "out of host memory");
    p->device = device; p->width = width; p->height = height; p->out_width = out_width; p->out_height = out_height;
    hipError_t e = hipMalloc((void**)&p->d_denoised, (size_t)width * height * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_window, (size_t)out_width * out_height * 4);
    if (e == hipSuccess) e = hipMemset(p->d_denoised, 0, (size_t)width * height * 4);
    if (e == hipSuccess) e = hipMemset(p->d_window, 0, (size_t)out_width * out_height * 4);
    if (e != hipSuccess) {
        (void)hipFree(p->d_denoised); (void)hipFree(p->d_window);
        delete p;
        return vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_presenter_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return VRC_OK;
}

extern "C" int vrc_presenter_destroy(vrc_presenter* p)
{
    if (!p) return VRC_OK;
    (void)hipSetDevice(p->device);
    (void)hipFree(p->d_denoised);
    (void)hipFree(p->d_window);
    delete p;
    return VRC_OK;
}

extern "C" int vrc_present_image(vrc_presenter* p, const void* image_dev, float old_value_conservation, uint32_t median, void* stream)
{
    if (!p || !image_dev) return vrc::fail(VRC_ERR_INVALID, "vrc_present: null argument");
    if (median != 0 && median != 3 && median != 5) return vrc::fail(VRC_ERR_INVALID, "vrc_present: median must be 0, 3 or 5");
    if (!(old_value_conservation >= 0.0f && old_value_conservation <= 1.0f))
        return vrc::fail(VRC_ERR_INVALID, "vrc_present: old_value_conservation must be in [0, 1]");
    HIP_TRYP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    // main.cpp:161-166: sf::Color(255 * c, 255 * c, 255 * c) -- float to Uint8 truncates
    const uint32_t c1 = (uint32_t)(uint8_t)(255 * old_value_conservation);
    const uint32_t c2 = (uint32_t)(uint8_t)(255 * (1.0f - old_value_conservation));
    const uint32_t n = p->width * p->height;
    hipLaunchKernelGGL(k_present_blend, dim3((n + 255u) / 256u), dim3(256), 0, st, (const uint32_t*)image_dev, p->d_denoised, n, c1, c2);
    const dim3 grid((p->out_width + 255u) / 256u, p->out_height);
    if (median == 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_present_window<0>), grid, dim3(256), 0, st, p->d_denoised, p->width, p->height, p->d_window, p->out_width, p->out_height);
    else if (median == 3)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_present_window<3>), grid, dim3(256), 0, st, p->d_denoised, p->width, p->height, p->d_window, p->out_width, p->out_height);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_present_window<5>), grid, dim3(256), 0, st, p->d_denoised, p->width, p->height, p->d_window, p->out_width, p->out_height);
    HIP_TRYP(hipGetLastError());
    return VRC_OK;
}

extern "C" int vrc_present(vrc_presenter* p, vrc_renderer* r, float old_value_conservation, uint32_t median, void* stream)
{
    if (!p || !r) return vrc::fail(VRC_ERR_INVALID, "vrc_present: null argument");
    int device = 0;
    uint32_t w = 0, h = 0;
    vrc::renderer_info(r, &device, &w, &h);
    if (device != p->device || w != p->width || h != p->height)
        return vrc::fail(VRC_ERR_INVALID, "vrc_present: the presenter is %ux%u on device %d, the renderer %ux%u on device %d",
                         p->width, p->height, p->device, w, h, device);
    return vrc_present_image(p, vrc_image_device_ptr(r), old_value_conservation, median, stream);
}

extern "C" void* vrc_presenter_window_ptr(vrc_presenter* p) { return p ? p->d_window : nullptr; }
extern "C" void* vrc_presenter_denoised_ptr(vrc_presenter* p) { return p ? p->d_denoised : nullptr; }

extern "C" int vrc_presenter_clear(vrc_presenter* p, void* stream)
{
    if (!p) return vrc::fail(VRC_ERR_INVALID, "null presenter");
    HIP_TRYP(hipSetDevice(p->device));
    HIP_TRYP(hipMemsetAsync(p->d_denoised, 0, (size_t)p->width * p->height * 4, (hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_presenter_read(vrc_presenter* p, uint8_t* window_rgba_host, uint8_t* denoised_rgba_host, void* stream)
{
    if (!p) return vrc::fail(VRC_ERR_INVALID, "null presenter");
    HIP_TRYP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    if (window_rgba_host) HIP_TRYP(hipMemcpyAsync(window_rgba_host, p->d_window, (size_t)p->out_width * p->out_height * 4, hipMemcpyDeviceToHost, st));
    if (denoised_rgba_host) HIP_TRYP(hipMemcpyAsync(denoised_rgba_host, p->d_denoised, (size_t)p->width * p->height * 4, hipMemcpyDeviceToHost, st));
    HIP_TRYP(hipStreamSynchronize(st));
    return VRC_OK;
}
