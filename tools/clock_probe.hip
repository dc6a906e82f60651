// What clock does the shader engine actually run at under load?  s_memtime counts shader cycles, s_memrealtime a constant
// 100 MHz reference: their ratio over a ~1 ms kernel is the sustained clock.  Two loads: back-to-back fp32 MFMAs only, and
// MFMAs interleaved with streaming 16-byte global loads (the mix of the implicit-GEMM kernels).
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MEM>
__global__ __launch_bounds__(256) void probe(const f32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ t, int iters, int n4) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 0.5f;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    int idx = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MEM) { v += src[idx]; idx += gridDim.x * 256; if (idx >= n4) idx -= n4; }
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    const int blocks = 1024, n4 = 1 << 26;       // 1 GiB of float4
    f32x4* src; float* out; unsigned long long* t;
    (void)hipMalloc(&src, (size_t)n4 * 16); (void)hipMemset(src, 0, (size_t)n4 * 16);
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&t, blocks * 16);
    unsigned long long* h = new unsigned long long[blocks * 2];
    for (int iters : {50, 200, 2000, 20000})
    for (int mem = 0; mem < 2; ++mem) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (mem) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, src, out, t, iters, n4);
            else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, src, out, t, iters, n4);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, t, blocks * 16, hipMemcpyDeviceToHost);
            double cyc = 0, ref = 0;
            for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; ref += h[2 * i + 1]; }
            const double mhz = cyc / ref * 100.0;
            const double tflops = (double)blocks * 4 /*waves*/ * iters * 32 /*mfma*/ * 4096.0 / (ms * 1e-3) / 1e12;
            printf("iters %5d %s rep %d: %.3f ms, shader clock %.0f MHz (s_memtime / s_memrealtime @100 MHz), %.1f TFLOP/s fp32 MFMA\n",
                   iters, mem ? "mfma + streaming loads" : "mfma only             ", rep, ms, mhz, tflops);
        }
    }
    return 0;
}
