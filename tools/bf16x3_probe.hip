// Round 5 probe: fp32-accurate GEMM from THREE-way bf16 splits on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32
// MFMA rate on gfx950).  x = hi + mid + lo exactly (each the top 8 significant bits of what is left, taken by truncation -- bit
// masks and exact subtractions), a b ~ the six products down to 2^-16 of the leading one.
//   (1) rate of the bf16 MFMA alone, of 300 v_fma alone, and of both in one instruction stream (do they overlap? fp32 MFMAs do not:
//       profiles/r4_pipe_overlap_probe.txt);
//   (2) accuracy against float64 of the fp32 MFMA chain, the six-product split and the three-product (two-way) split, K = 4608.
//   hipcc --offload-arch=gfx950 -O3 tools/bf16x3_probe.hip -o /tmp/bf16x3_probe && /tmp/bf16x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WHAT, int NM, int NV>
__global__ __launch_bounds__(512) void rate(float* __restrict__ out, unsigned long long* __restrict__ t, int iters) {
    const int tid = threadIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 au = {0x3f803f80u + tid, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, bu = {0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u + tid};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
    float va[8];
    for (int j = 0; j < 8; ++j) va[j] = tid * 0.01f + j;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (WHAT & 4) {
#pragma unroll
            for (int j = 0; j < NV; ++j) va[j & 7] = __builtin_fmaf(va[j & 7], 1.0001f, va[(j + 3) & 7]);
        }
        if (WHAT & 1) {
#pragma unroll
            for (int k = 0; k < NM / 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if constexpr ((WHAT & 1) && (WHAT & 4)) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (NV + NM - 1) / NM, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += va[j];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int WHAT, int NM, int NV>
static void run_rate(const char* what, float* out, unsigned long long* t) {
    const int blocks = 256, iters = 2000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((rate<WHAT, NM, NV>), dim3(blocks), dim3(512), 0, 0, out, t, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-64s %8.3f ms  = %7.1f ns per iteration\n", what, best, best * 1e6 / iters);
}

// ---- accuracy: one wave, C (32 x 32) = A (32 x K) B (K x 32); A row-major [32][K], B as [32 cols][K] (both K-contiguous)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const float hf = __builtin_bit_cast(float, xb & 0xffff0000u);
    const float r = x - hf;
    const unsigned rb = __builtin_bit_cast(unsigned, r);
    const float mf = __builtin_bit_cast(float, rb & 0xffff0000u);
    const float q = r - mf;
    h = xb >> 16; m = rb >> 16; l = __builtin_bit_cast(unsigned, q) >> 16;      // (q has at most 8 significant bits: exact)
}
__device__ __forceinline__ bf16x8 pack8(const unsigned (&v)[8]) {
    u32x4 u = {v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
    return __builtin_bit_cast(bf16x8, u);
}
__global__ __launch_bounds__(64) void accuracy(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C) {
    const int lane = threadIdx.x, l31 = lane & 31, kh = lane >> 5;
    f32x16 c32, c6, c3;
    for (int e = 0; e < 16; ++e) c32[e] = c6[e] = c3[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        // fp32 chain: 8 MFMAs 32x32x2 (k pairs k0 + 2 s + kh)
        for (int s = 0; s < 8; ++s) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k0 + 2 * s + kh], B[l31 * K + k0 + 2 * s + kh], c32, 0, 0, 0);
        unsigned ah[8], am[8], al[8], bh[8], bm[8], bl[8];
        for (int j = 0; j < 8; ++j) {
            split3(A[l31 * K + k0 + 8 * kh + j], ah[j], am[j], al[j]);
            split3(B[l31 * K + k0 + 8 * kh + j], bh[j], bm[j], bl[j]);
        }
        const bf16x8 a1 = pack8(ah), a2 = pack8(am), a3 = pack8(al), b1 = pack8(bh), b2 = pack8(bm), b3 = pack8(bl);
        // smallest terms first
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c6, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kh;
        C[0 * 1024 + row * 32 + l31] = c32[e];
        C[1 * 1024 + row * 32 + l31] = c6[e];
        C[2 * 1024 + row * 32 + l31] = c3[e];
    }
}

int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 16);
    printf("per wave and iteration, 8 waves per CU, 256 workgroups (one per CU)\n");
    run_rate<1, 36, 0>("36 MFMA 32x32x16 bf16 alone", out, t);
    run_rate<1, 72, 0>("72 MFMA 32x32x16 bf16 alone", out, t);
    run_rate<4, 0, 252>("252 v_fma_f32 alone", out, t);
    run_rate<5, 36, 252>("36 bf16 MFMA + 252 v_fma_f32, one stream", out, t);
    run_rate<5, 72, 252>("72 bf16 MFMA + 252 v_fma_f32, one stream", out, t);
    run_rate<5, 36, 144>("36 bf16 MFMA + 144 v_fma_f32, one stream", out, t);
    const int K = 4608;
    std::vector<float> A(32 * K), B(32 * K), C(3 * 1024);
    srand(7);
    auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
    for (auto& x : A) x = (float)nrm();
    for (auto& x : B) x = (float)(nrm() / sqrt((double)K));
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, C.size() * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, e[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)A[i * K + k] * (double)B[j * K + k];
            mx = fmax(mx, fabs(r));
            for (int v = 0; v < 3; ++v) { const double d = fabs(C[v * 1024 + i * 32 + j] - r); e[v] = fmax(e[v], d); sq[v] += d * d; }
        }
    printf("accuracy against float64, K = %d, 32 x 32 outputs, max|ref| %.3f\n", K, mx);
    const char* nm[3] = {"fp32 MFMA chain (32x32x2 f32)", "bf16 three-way split, six products", "bf16 two-way split, three products"};
    for (int v = 0; v < 3; ++v) printf("  %-40s max error %.3e of max|ref|, rms %.3e\n", nm[v], e[v] / mx, sqrt(sq[v] / 1024) / mx);
    return 0;
}
