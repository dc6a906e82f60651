// Lab harness: bisects the per-chunk cost of the implicit-GEMM main loop (generator config: 64 px x 32 n per wave,
// 9 chunks of 32 channels).  Variants selected by template flags.  hipcc --offload-arch=gfx950 -O3 igemm_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LOAD_A, bool USE_B_LDS, bool BARRIER, bool STORE, int NCHUNK>
__global__ __launch_bounds__(256) void lab(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int in_bytes) {
    __shared__ __attribute__((aligned(16))) float Bs[2][32 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 256 + wave * 64;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), (short)0, in_bytes, 0x00020000);
    unsigned boff[2];
    for (int i = 0; i < 2; ++i) boff[i] = (unsigned)(((m0 + i * 32 + l31) * 32 + kh * 16) * 4);
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 an[2][4], ac[2][4], bc[4];
    float bn[4];
    auto load = [&](int chunk) {
        if (LOAD_A) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) an[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, boff[i] + (chunk % 3) * 128, 16 * j, 0));
        }
        if (USE_B_LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int e = tid + i * 256, c = e & 31, n = e >> 5;
                bn[i] = w[n * 288 + c * 9 + chunk];
            }
        }
    };
    auto store_b = [&](int buf) {
        if (USE_B_LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { int e = tid + i * 256, c = e & 31, n = e >> 5; Bs[buf][n * 36 + c] = bn[i]; }
        }
    };
    if (!LOAD_A) for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) an[i][j] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
    if (!USE_B_LDS) for (int q = 0; q < 4; ++q) bc[q] = f32x4{0.5f, 0.25f, 0.125f, 1.f + lane};
    load(0);
    store_b(0);
    __syncthreads();
    int buf = 0;
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) ac[i][j] = an[i][j];
        if (USE_B_LDS) {
            const float* row = &Bs[buf][l31 * 36 + kh * 16];
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[q] = *reinterpret_cast<const f32x4*>(row + 4 * q);
        }
        load(chunk + 1 < NCHUNK ? chunk + 1 : chunk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[i][kk >> 2][kk & 3], bc[kk >> 2][kk & 3], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        store_b(buf ^ 1);
        if (BARRIER) __syncthreads();
        buf ^= 1;
    }
    if (STORE) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int m = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                out[(long long)m * 32 + l31] = acc[i][e];
            }
    } else {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        if (s == 12345.678f) out[tid] = s;
    }
}

template <typename F>
void run(const char* name, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f us\n", name, ms * 1e3 / 20);
}
#define RUN(LA, LB, BAR, ST, NC) run("A=" #LA " Blds=" #LB " bar=" #BAR " store=" #ST " chunks=" #NC, [&] { hipLaunchKernelGGL((lab<LA, LB, BAR, ST, NC>), dim3(512), dim3(256), 0, 0, in, w, out, bytes); })
int main() {
    const int bytes = 32 * 64 * 64 * 32 * 4;
    float *in, *w, *out;
    (void)hipMalloc(&in, bytes); (void)hipMalloc(&out, bytes); (void)hipMalloc(&w, 32 * 288 * 4);
    (void)hipMemset(in, 0, bytes); (void)hipMemset(w, 0, 32 * 288 * 4);
    RUN(true, true, true, true, 9);
    RUN(true, true, true, true, 1);
    RUN(true, true, true, false, 9);
    RUN(false, true, true, true, 9);
    RUN(true, false, false, true, 9);
    RUN(false, false, false, true, 9);
    RUN(false, false, false, false, 9);
    RUN(false, false, false, false, 1);
    RUN(false, false, true, false, 9);
    RUN(false, true, false, false, 9);
    RUN(true, false, false, false, 9);
    RUN(true, true, true, true, 18);
    return 0;
}
