// How do the pipes of a CU compose?  One 512-thread workgroup per CU (two waves per SIMD, as in the Winograd kernels) runs, per
// iteration and wave, NM fp32 MFMAs (32x32x2, 64 clocks of the matrix pipe each), NL 16-byte buffer loads from an L2-resident
// window (8 full 128-byte lines per instruction, consumed two iterations later), NV plain vector-ALU instructions and ND
// 16-byte LDS loads -- each kind alone, all in one instruction stream (every wave does everything, interleaved), and
// SPECIALISED: the first four waves (one per SIMD) take all the MFMAs, the last four all the loads / ALU / LDS work, so that
// every SIMD has one matrix wave and one other wave.  Times per iteration in shader clocks (s_memtime) and the clock itself.
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o /tmp/overlap_probe && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// -DPROBE_BF16 (round 5): the matrix instruction is v_mfma_f32_32x32x16_bf16 (32 clocks of the matrix pipe each) instead of the fp32 one --
// does the BF16 matrix pipe run beside vector-ALU work of the same SIMD, which the fp32 one does not?
#ifdef PROBE_BF16
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PROBE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, a), 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), __builtin_bit_cast(bf16x8, u32x4{__builtin_bit_cast(unsigned, b), 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), c, 0, 0, 0)
#define PROBE_NAME "32x32x16 bf16"
#else
#define PROBE_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define PROBE_NAME "32x32x2 f32"
#endif

// WHAT bits: 1 MFMA, 2 loads, 4 vector ALU, 8 LDS.  SPEC: 0 = every wave does its share of everything; 1 = waves 0-3 the MFMAs
// (twice as many each), waves 4-7 the rest (twice as much each): the same work per CU and per SIMD.
template <int WHAT, int SPEC, int NM, int NL, int NV, int ND>
__global__ __launch_bounds__(512) void probe(const float* __restrict__ src, unsigned src_bytes, float* __restrict__ out,
                                             unsigned long long* __restrict__ t, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 512) lds[i] = (float)i;
    __syncthreads();
    const bool mat = !SPEC || wave < 4, oth = !SPEC || wave >= 4;
    constexpr int MUL = SPEC ? 2 : 1;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 0.5f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, (int)src_bytes, 0x00020000);
    // a 64 KB window per workgroup (16 MB for the chip: L2-resident, larger than the 32 KB L1); an instruction reads 1 KB =
    // 8 whole lines: lane l the 16 bytes at 16 l of a 1 KB row
    const unsigned win = (blockIdx.x & 255) * 65536u;
    f32x4 d0[NL * MUL > 0 ? NL * MUL : 1], d1[NL * MUL > 0 ? NL * MUL : 1];
    for (int j = 0; j < NL * MUL; ++j) { d0[j] = f32x4{0, 0, 0, 0}; d1[j] = d0[j]; }
    float va[8];
    for (int j = 0; j < 8; ++j) va[j] = tid * 0.01f + j;
    f32x4 ls = {0, 0, 0, 0};
    unsigned row = wave * 8;
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    auto body = [&](f32x4 (&dn)[NL * MUL > 0 ? NL * MUL : 1]) {
        if ((WHAT & 2) && oth) {
#pragma unroll
            for (int j = 0; j < NL * MUL; ++j) {
                asm volatile("" :: "v"(dn[j]));                       // (the data requested two iterations ago is "used" here)
                dn[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, win + ((row + j) & 63u) * 1024u + lane * 16u, 0, 0));
            }
            row += NL * MUL;
        }
        if ((WHAT & 8) && oth) {
#pragma unroll
            for (int j = 0; j < ND * MUL; ++j) ls += *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + j * 256 + wave * 64) & 16383));
        }
        if ((WHAT & 4) && oth) {
#pragma unroll
            for (int j = 0; j < NV * MUL; ++j) va[j & 7] = __builtin_fmaf(va[j & 7], 1.0001f, va[(j + 3) & 7]);
        }
        if ((WHAT & 1) && mat) {
#pragma unroll
            for (int k = 0; k < NM * MUL / 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = PROBE_MFMA(a, b, acc[i]);
        }
        if (!SPEC && (WHAT & 1) && (WHAT & ~1)) {
            // one stream: a matrix instruction, then its share of the rest
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (WHAT & 4) __builtin_amdgcn_sched_group_barrier(0x002, (NV + NM - 1) / NM, 0);
                if ((WHAT & 2) && (i % (NM / (NL ? NL : 1) ? NM / (NL ? NL : 1) : 1)) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((WHAT & 8) && (i % (NM / (ND ? ND : 1) ? NM / (ND ? ND : 1) : 1)) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 2) {
        body(d0);
        body(d1);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = ls[0] + ls[1] + ls[2] + ls[3];
    for (int j = 0; j < NL * MUL; ++j) s += d0[j][0] + d1[j][3];
    for (int j = 0; j < 8; ++j) s += va[j];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int WHAT, int SPEC, int NM, int NL, int NV, int ND>
static void run(const char* what, const float* src, unsigned bytes, float* out, unsigned long long* t, unsigned long long* h) {
    const int blocks = 256, iters = 2000;
    float best = 1e9f; double clk = 0, cyc_it = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<WHAT, SPEC, NM, NL, NV, ND>), dim3(blocks), dim3(512), 0, 0, src, bytes, out, t, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, t, blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, ref = 0;
        for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; ref += h[2 * i + 1]; }
        if (ms < best) { best = ms; clk = cyc / ref * 100.0; cyc_it = cyc / blocks / iters; }
    }
    printf("%-58s %8.3f ms  %7.0f clocks / iteration  %5.0f MHz\n", what, best, cyc_it, clk);
}

int main() {
    const unsigned bytes = 16u << 20;
    float* src; float* out; unsigned long long* t;
    (void)hipMalloc(&src, bytes); (void)hipMemset(src, 0, bytes);
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&t, 256 * 16);
    unsigned long long* h = new unsigned long long[512];
    // per wave and iteration: 24 MFMAs (2 x 24 x 64 = 3072 clocks of a SIMD's matrix pipe), 12 loads, 300 ALU instructions, 24 LDS loads
#ifdef PROBE_BF16
    constexpr int NM = 48, NL = 12, NV = 300, ND = 24;      // (48 x 32 clocks: the same matrix-pipe time as 24 fp32 MFMAs)
#else
    constexpr int NM = 24, NL = 12, NV = 300, ND = 24;
#endif
    printf("per wave and iteration: %d MFMA " PROBE_NAME ", %d buffer_load_dwordx4 (1 KB each, L2-resident), %d v_fma_f32, %d ds_read_b128; 8 waves per CU\n", NM, NL, NV, ND);
    run<1, 0, NM, NL, NV, ND>("MFMAs alone", src, bytes, out, t, h);
    run<2, 0, NM, NL, NV, ND>("loads alone", src, bytes, out, t, h);
    run<4, 0, NM, NL, NV, ND>("vector ALU alone", src, bytes, out, t, h);
    run<8, 0, NM, NL, NV, ND>("LDS loads alone", src, bytes, out, t, h);
    run<3, 0, NM, NL, NV, ND>("MFMAs + loads, one stream per wave", src, bytes, out, t, h);
    run<3, 1, NM, NL, NV, ND>("MFMAs + loads, specialised waves (one of each per SIMD)", src, bytes, out, t, h);
    run<5, 0, NM, NL, NV, ND>("MFMAs + ALU, one stream per wave", src, bytes, out, t, h);
    run<5, 1, NM, NL, NV, ND>("MFMAs + ALU, specialised waves", src, bytes, out, t, h);
    run<9, 0, NM, NL, NV, ND>("MFMAs + LDS, one stream per wave", src, bytes, out, t, h);
    run<9, 1, NM, NL, NV, ND>("MFMAs + LDS, specialised waves", src, bytes, out, t, h);
    run<6, 0, NM, NL, NV, ND>("loads + ALU, one stream per wave", src, bytes, out, t, h);
    run<15, 0, NM, NL, NV, ND>("everything, one stream per wave", src, bytes, out, t, h);
    run<15, 1, NM, NL, NV, ND>("everything, specialised waves", src, bytes, out, t, h);
    // the same with fewer ALU instructions (what fits the gaps of the MFMAs: 6 per MFMA)
    run<5, 0, NM, NL, 144, ND>("MFMAs + 144 ALU, one stream per wave", src, bytes, out, t, h);
    run<5, 1, NM, NL, 144, ND>("MFMAs + 144 ALU, specialised waves", src, bytes, out, t, h);
    return 0;
}
