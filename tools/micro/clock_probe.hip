// Shader clock while something else runs on the GPU: one wave samples s_memtime (shader clock cycles) against s_memrealtime
// (100 MHz) over windows of ~20 ms and prints the ratio.  Run it beside bench.py (another process on the same GPU) to see the
// clock the frame kernel actually gets -- bench.py prices cycles at the nominal 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe [windows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k(unsigned long long* out, int windows, unsigned long long window_ticks)
{
    if (threadIdx.x) return;
    for (int w = 0; w < windows; ++w) {
        unsigned long long c0, r0, c1, r1;
        asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
        do {
            __builtin_amdgcn_s_sleep(32);
            asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
        } while (r1 - r0 < window_ticks);
        out[2 * w] = c1 - c0;
        out[2 * w + 1] = r1 - r0;
    }
}

int main(int argc, char** argv)
{
    const int windows = argc > 1 ? atoi(argv[1]) : 50;
    unsigned long long* d;
    (void)hipMalloc(&d, 16 * windows);
    k<<<1, 64>>>(d, windows, 2000000ull);        // 20 ms at 100 MHz
    (void)hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(16 * windows);
    (void)hipMemcpy(h, d, 16 * windows, hipMemcpyDeviceToHost);
    double lo = 1e9, hi = 0, sum = 0;
    for (int w = 0; w < windows; ++w) {
        const double mhz = (double)h[2 * w] / (double)h[2 * w + 1] * 100.0;
        printf("window %2d: %.0f MHz\n", w, mhz);
        lo = mhz < lo ? mhz : lo; hi = mhz > hi ? mhz : hi; sum += mhz;
    }
    printf("shader clock over %d windows of 20 ms: min %.0f  mean %.0f  max %.0f MHz\n", windows, lo, sum / windows, hi);
    return 0;
}
