// What does a vector-ALU instruction cost a SIMD that also runs fp32 MFMAs?  (tools/overlap_probe.hip: fp32 MFMAs and vector-ALU
// instructions do not overlap on a SIMD at all -- the fp32 matrix rate IS the vector FMA rate -- so every instruction of a
// transform is paid in full.)  Here: 512-thread workgroups, one per CU, each wave issues N instructions of one kind per iteration
// (independent chains); clocks per instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* __restrict__ out, unsigned long long* __restrict__ t, int iters) {
    const int tid = threadIdx.x;
    float a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    float b = 1.0001f, c = 0.5f;
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            } else if (KIND == 1) {          // packed: two floats per lane and instruction (register pairs a0:a1 ...)
                asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3"
                             : "+v"(*reinterpret_cast<double*>(&a0)), "+v"(*reinterpret_cast<double*>(&a2)) : "v"(*reinterpret_cast<double*>(&a4)), "v"(*reinterpret_cast<double*>(&a6)));
            } else if (KIND == 2) {
                asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_mov_b32_dpp %1, %5 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_mov_b32_dpp %2, %4 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_mov_b32_dpp %3, %5 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a4), "v"(a5));
            } else if (KIND == 3) {
                asm volatile("v_add_f32_dpp %0, %4, %0 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_mul_f32_dpp %1, %5, %1 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_add_f32_dpp %2, %4, %2 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_mul_f32_dpp %3, %5, %3 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
            } else if (KIND == 4) {
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3\n\tv_pk_mul_f32 %0, %0, %2\n\tv_pk_mul_f32 %1, %1, %3"
                             : "+v"(*reinterpret_cast<double*>(&a0)), "+v"(*reinterpret_cast<double*>(&a2)) : "v"(*reinterpret_cast<double*>(&a4)), "v"(*reinterpret_cast<double*>(&a6)));
            } else {
                asm volatile("v_add_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %5\n\tv_mul_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    out[blockIdx.x * 512 + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (tid == 0) t[blockIdx.x] = c1 - c0;
}

template <int KIND>
static void run(const char* what, float* out, unsigned long long* t) {
    const int blocks = 256, iters = 4000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND>), dim3(blocks), dim3(512), 0, 0, out, t, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // 2 waves per SIMD x 64 instructions per iteration
    printf("%-46s %7.3f ms   %5.2f ns per instruction and SIMD (= %4.1f clocks at 2.3 GHz)\n", what, best, best * 1e6 / (4000.0 * 128), best * 1e6 / (4000.0 * 128) * 2.3);
}

int main() {
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 8);
    run<0>("v_fma_f32", out, t);
    run<5>("v_add / v_sub / v_mul _f32", out, t);
    run<1>("v_pk_fma_f32 (two floats per lane)", out, t);
    run<4>("v_pk_add_f32 / v_pk_mul_f32", out, t);
    run<2>("v_mov_b32_dpp quad_perm", out, t);
    run<3>("v_add_f32_dpp / v_mul_f32_dpp quad_perm", out, t);
    return 0;
}
