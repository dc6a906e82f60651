// Does the fp32 MFMA rate depend on the DATA?  Same dependent 32x32x2 chain, 4 waves per SIMD, sustained for ~0.2 s per
// mode: (0) constant operands, zero-ish accumulators; (1) sixteen random operand registers rotating, random signs.
// Reports TFLOP/s of the last launches and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float a[16], b[16];
    for (int u = 0; u < 16; ++u) {
        if (MODE == 0) { a[u] = 0.f; b[u] = 0.f; }
        else {
            const unsigned h = hash(threadIdx.x * 977u + blockIdx.x * 131u + u * 7919u);
            a[u] = __uint_as_float(0x3f000000u | (h & 0x807fffffu));          // +-[0.5, 1), random mantissa
            b[u] = __uint_as_float(0x3f000000u | (hash(h) & 0x807fffffu));
        }
    }
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE> void run(const char* name, float* out, unsigned long long* clk) {
    const int blocks = 1024, iters = 4000;      // 4 waves per SIMD; ~1.7 ms per launch at peak
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);   // warm: ~0.2 s
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 16 * 4096.0 * 20;
    printf("%-28s %7.1f TFLOP/s   shader clock %.0f MHz\n", name, flops / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 100.0);
}
int main() {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    run<0>("zero operands", out, clk);
    run<1>("random operands", out, clk);
    run<0>("zero operands (again)", out, clk);
    run<1>("random operands (again)", out, clk);
    return 0;
}
