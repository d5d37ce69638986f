// Exhaustive check (all 2^31 non-negative float bit patterns): which short sequences give the correctly rounded sqrt(x)
// (what hipcc's IEEE sqrtf gives on gfx950)?  Reports mismatches per exponent field of x for each candidate.
//   A: y = v_rsq_f32(x); g = x * y; h = 0.5 * y; d = fma(-g, g, x); s = fma(d, h, g)
//   B: A with one more refinement of (g, h) before the residual step
//   C: s0 = v_sqrt_f32(x); d = fma(-s0, s0, x); s = fma(d, 0.5 * v_rcp_f32(s0), s0)
// Also checks get_rand's quotient m / 100 (m = 0..99) by q0 = m * (1/100); r = fma(-100, q0, m); q = fma(r, 1/100, q0).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/sqrt_exact.hip -o /tmp/sqrt_exact && /tmp/sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float hw_rsq(float x) { float y; asm volatile("v_rsq_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ float hw_sqrt(float x) { float y; asm volatile("v_sqrt_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ float hw_rcp(float x) { float y; asm volatile("v_rcp_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }

__device__ __forceinline__ float sqrt_a(float x)
{
    const float y = hw_rsq(x), g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_b(float x)
{
    const float y = hw_rsq(x);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_c(float x)
{
    const float s0 = hw_sqrt(x);
    const float d = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(d, 0.5f * hw_rcp(s0), s0);
}

__global__ void k_check(unsigned long long* bad /* [3][256] */)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // 2^23 threads x 256 values
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t bits = (uint32_t)(tid * 256u + k);
        const float x = __uint_as_float(bits);
        const float ref = sqrtf(x);
        const uint32_t rb = __float_as_uint(ref);
        const float c[3] = {sqrt_a(x), sqrt_b(x), sqrt_c(x)};
        for (int v = 0; v < 3; ++v) {
            const bool both_nan = (ref != ref) && (c[v] != c[v]);
            if (__float_as_uint(c[v]) != rb && !both_nan) atomicAdd(&bad[v * 256 + ((bits >> 23) & 0xff)], 1ull);
        }
    }
}

__global__ void k_rand(unsigned int* badq)
{
    const uint32_t m = threadIdx.x;
    if (m >= 100u) return;
    const float a = (float)m;
    const float ref = a / 100.0f;
    const float y = 1.0f / 100.0f;                     // the correctly rounded reciprocal, a compile-time constant
    const float q0 = a * y;
    const float r = __builtin_fmaf(-100.0f, q0, a);
    const float q = __builtin_fmaf(r, y, q0);
    if (__float_as_uint(q) != __float_as_uint(ref)) atomicAdd(badq, 1u);
}

int main()
{
    unsigned long long *d_bad, bad[3 * 256];
    unsigned int *d_q, q = 0;
    hipMalloc(&d_bad, sizeof(bad)); hipMemset(d_bad, 0, sizeof(bad));
    hipMalloc(&d_q, 4); hipMemset(d_q, 0, 4);
    hipLaunchKernelGGL(k_check, dim3(1u << 15), dim3(256), 0, 0, d_bad);
    hipLaunchKernelGGL(k_rand, dim3(1), dim3(128), 0, 0, d_q);
    hipDeviceSynchronize();
    hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    hipMemcpy(&q, d_q, 4, hipMemcpyDeviceToHost);
    const char* names[3] = {"A rsq + residual", "B rsq + refine + residual", "C sqrt + residual"};
    for (int v = 0; v < 3; ++v) {
        unsigned long long total = 0;
        int first = -1, last = -1;
        for (int e = 0; e < 256; ++e) if (bad[v * 256 + e]) { total += bad[v * 256 + e]; if (first < 0) first = e; last = e; }
        printf("%-28s total mismatches %llu; exponent fields with mismatches:", names[v], total);
        for (int e = 0; e < 256; ++e) if (bad[v * 256 + e]) printf(" %d(%llu)", e, bad[v * 256 + e]);
        printf(" [first %d last %d]\n", first, last);
    }
    printf("m / 100 for m = 0..99 by reciprocal + residual: %u mismatches\n", q);
    return 0;
}
