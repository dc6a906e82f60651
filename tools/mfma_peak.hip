// Calibration: fp32 MFMA issue rate on this chip, registers only.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.1f us  %7.1f TFLOP/s\n", name, ms * 1e3 / 5, flops / (ms / 5 * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 200;
    for (int blocks : {256, 512, 768, 1024}) {
        double f32 = (double)blocks * 4 * iters * 16 * 4096.0;
        char nm[128];
        snprintf(nm, 128, "32x32x2 1acc  %d WG", blocks); run(nm, [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, f32 * 1);
        snprintf(nm, 128, "32x32x2 2acc  %d WG", blocks); run(nm, [&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, f32 * 2);
        snprintf(nm, 128, "32x32x2 4acc  %d WG", blocks); run(nm, [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, f32 * 4);
        double f16 = (double)blocks * 4 * iters * 16 * 2048.0;
        snprintf(nm, 128, "16x16x4 4acc  %d WG", blocks); run(nm, [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, f16 * 4);
        snprintf(nm, 128, "16x16x4 8acc  %d WG", blocks); run(nm, [&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, f16 * 8);
    }
    return 0;
}
