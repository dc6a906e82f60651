// Column transforms + channel mix of the Res-FFT-Conv block (arch/Ours/networks.py:24-30 and its backward), four waves
// per unit.  Same mathematics, data layouts and slab format as spec_mix_fwd_kernel / spec_mix_bwd_kernel (resfft.hip):
//   forward : S = FFT_h(R) / 8;  Z = W2 S + b2;  T = IFFT_h(relu(Z)) / 8          per (patch, kw column)
//   backward: gZ = [Z > 0] FFT_h(gR) / 8;  gS = W2^T gZ;  dW2 += gZ S^T;  db2 += sum gZ;  gT = IFFT_h(gS) * sc
//
// Why a second form.  A unit is two kw columns of one patch (544 units at 32 patches).  The one-wave kernels run a unit's
// phases one after the other in a single wave -- loads 2.3 us, 64-point transform in registers + LDS + spectrum store
// 5.3, 256 dependent-chain MFMAs 11.9, inverse transform + stores 3.1 (in-kernel stamps) -- on 544 of the
// chip's 1024 SIMDs: 28 / 40 us per launch at 0.14 / 0.19 MFMA utilisation, latency-bound.  Here a unit is a 256-thread
// workgroup:
//   * the columns arrive by LDS-DMA (32 x 1 KB instructions over the four waves instead of 128 dword loads per lane);
//   * a 64-point transform is split over the FOUR lanes of a quad: lane j takes points 4 m + j through a 16-point
//     register FFT (radix-2, immediates as twiddles), multiplies by w64^(j r) and the quad finishes with a 4-point
//     butterfly through DPP quad_perm -- 64 transforms x 4 lanes = all 256 threads busy, ~45 % of the one-lane work each;
//   * the 64 x 64 mix of a column is four 32 x 32 accumulator blocks: 64 MFMAs per wave instead of 256;
//   * the saved pre-activation (only its sign is ever used) leaves as a BIT MASK, 1 KB per unit instead of 32 KB.
// 2 176 waves on 1024 SIMDs, two workgroups per CU (66 KB of LDS each), so a unit's phases overlap its neighbour's.
// Roofline: HBM (forward 17.3 MB in, 34.6 MB + 0.5 MB out per 32 patches; backward 17.3 + 17.3 + 0.5 in, 17.3 + 9.2 out).
#include "common.h"
#include "fft64.h"

namespace {

constexpr int XLD4 = 65;                  // LDS row stride (floats) of the [frequency][64 channel] operand image
constexpr int MIX_SLAB4 = 64 * 64 + 128;  // dW2 partial + two db2 partial rows per unit (resfft.hip MIX_SLAB)
typedef __attribute__((address_space(3))) float lds_float4k;

// 32 KB (two adjacent 16 KB column blocks) or 16 KB global -> LDS by the four waves of the workgroup
__device__ __forceinline__ void dma_columns(const float* gsrc, float* lds_dst, int wave, int lane, bool two) {
    const int n = two ? 32 : 16;
    for (int i = wave; i < n; i += 4)
        __builtin_amdgcn_global_load_lds(gsrc + i * 256 + lane * 4, (lds_float4k*)(lds_dst + i * 256), 16, 0, 0);
}

// 32 rows (stride ld floats in LDS) x 64 floats -> 32 consecutive 256-byte rows of global memory, one wave, 16-byte stores
// (a vector-memory store costs the CU about the same whatever its width: 8 instructions here instead of 32 dword ones)
__device__ __forceinline__ void store_rows32(const float* lds_rows, int ld, float* gdst, int lane) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), q = lane & 15;
        const float* s = lds_rows + row * ld + 4 * q;
        const f32x4 v = {s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(gdst + row * 64 + 4 * q) = v;
    }
}

// Lab (MTD_SPECMIX_STAGGER = d, MTD_SPECMIX_STAGGER_MOD = m): workgroups that share a CU start their compute phases d half-microseconds
// apart (group = (linear workgroup id / 256) % m), after their input request is out -- de-phases the all-resident lock step.
__device__ __forceinline__ void stagger_wait(int lin, int d, int m) {
    if (d > 0) {
        const int n = ((lin >> 8) % m) * d;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
}

// position of element (row kh of a 64-row column, channel o) in the sign masks: word, bit
__device__ __forceinline__ int mask_word(int k2, int kh, int o) {
    const int i = kh >> 5, r32 = kh & 31;
    const int e = (r32 & 3) + 4 * (r32 >> 3);
    return ((k2 * 2 + i) * 2 + (o >> 5)) * 16 + e;
}
__device__ __forceinline__ int mask_bit(int kh, int o) { return (o & 31) + 32 * (((kh & 31) >> 2) & 1); }

// COLS: kw columns per workgroup (2: 256 threads, a pair; 1: 128 threads, one column of a pair -- twice the units in flight)
template <int COLS>
__global__ __launch_bounds__(128 * COLS, 3) void spec_mix_fwd4_kernel(const float* __restrict__ R, const float* __restrict__ w2t,
                                                               const float* __restrict__ b2, float* __restrict__ T,
                                                               float* __restrict__ S_save, unsigned long long* __restrict__ zmask,
                                                               int stg_d, int stg_m) {
    // ONE LDS image, used in turn as the DMA target (rows of 64 floats), the MFMA operand image (rows of XLD4) and the
    // staging area of the result rows (64 again): 33 KB per workgroup, so THREE workgroups per CU = 768 slots for the 544
    // units of a 32-patch launch.  With an input and an operand buffer (66 KB, two per CU, 512 slots) the last 32 units
    // ran as a second round on an empty chip: 26 us per launch for 13 us of work per unit.
    __shared__ __attribute__((aligned(1024))) float Xs[COLS * 64 * XLD4];
    float* const Xin = Xs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int kw0 = COLS == 2 ? 2 * (int)blockIdx.x : (int)blockIdx.x;      // first (only) column of this workgroup
    const int unit = b * 17 + (kw0 >> 1);                         // the pair the sign masks are filed under
    const int kpos = COLS == 2 ? 0 : (kw0 & 1);                   // position of column kwl = 0 inside its pair
    const bool two = COLS == 2 && (kw0 + 1) < NKW;                // workgroup-uniform: does the second column exist
    const long long cb0 = ((long long)(b * NKW + kw0) * 64) * 64;
    {
        const int n = two ? 32 : 16;
        for (int i = wave; i < n; i += 2 * COLS)
            __builtin_amdgcn_global_load_lds(R + cb0 + i * 256 + lane * 4, (lds_float4k*)(Xin + i * 256), 16, 0, 0);
    }
    stagger_wait(blockIdx.y * gridDim.x + blockIdx.x, stg_d, stg_m);
    // transform lanes: f = transform (column kwl, channel c), j = position in the quad
    const int f = tid >> 2, j = tid & 3, kwl = f >> 5, c = f & 31;
    const bool valid = kwl == 0 || two;
    float tc[16], ts[16];
    quad_twiddles(j, tc, ts);
    // mix lanes
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int k2 = wave >> 1, ih = wave & 1;
    const bool mix_valid = k2 == 0 || two;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float re[16], im[16];
    // ---- column FFT, spectrum to LDS (MFMA operand image) and to S_save
    {
        const float* src = Xin + kwl * 4096 + j * 64 + c;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            re[m] = valid ? src[m * 256] : 0.f;
            im[m] = valid ? src[m * 256 + 32] : 0.f;
        }
        __syncthreads();                // every column is in registers: the image may be rewritten in operand layout
        fft64_quad<-1>(re, im, tc, ts, j);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kh = 16 * j + r;
            const float sr = re[r] * 0.125f, si = im[r] * 0.125f;
            Xs[(kwl * 64 + kh) * XLD4 + c] = sr;
            Xs[(kwl * 64 + kh) * XLD4 + 32 + c] = si;
        }
    }
    // the mix weights as MFMA B fragments (one batch of loads, under the barrier; issued before the transform instead they
    // made the launch slower, 23.0 -> 24.4 us)
    float wf0[32], wf1[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
        wf0[kk] = w2t[(2 * kk + kh2) * 64 + l31];
        wf1[kk] = w2t[(2 * kk + kh2) * 64 + 32 + l31];
    }
    __syncthreads();
    // ---- channel mix: wave (k2, ih) owns rows k2*64 + ih*32 .. +31 of the operand image, both output halves
    if (mix_valid) {
        if (S_save) store_rows32(Xs + (k2 * 64 + ih * 32) * XLD4, XLD4, S_save + cb0 + (k2 * 64 + ih * 32) * 64, lane);   // the saved spectrum
        f32x16 acc[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jj][e] = 0.f;
        const float* arow = Xs + (k2 * 64 + ih * 32 + l31) * XLD4 + kh2;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float a0 = arow[2 * kk];
            acc[0] = mfma32(a0, wf0[kk], acc[0]);
            acc[1] = mfma32(a0, wf1[kk], acc[1]);
        }
        // only this wave reads these rows: it may overwrite them once its own reads have been issued (LDS is in order per wave)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int o = jj * 32 + l31;
            const float bo = b2[o];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kh = ih * 32 + mfma32_row(e, lane);
                const float z = acc[jj][e] + bo;
                if (zmask) {
                    const unsigned long long bm = __ballot(z > 0.f);
                    if (lane == 0) zmask[(long long)unit * 128 + (((k2 + kpos) * 2 + ih) * 2 + jj) * 16 + e] = bm;
                }
                Xs[(k2 * 64 + kh) * XLD4 + o] = z > 0.f ? z : 0.f;
            }
        }
    }
    __syncthreads();
    // ---- inverse column FFT
    {
        const float* src = Xs + (kwl * 64 + j) * XLD4 + c;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            re[m] = src[4 * m * XLD4];
            im[m] = src[4 * m * XLD4 + 32];
        }
        __syncthreads();                // every spectrum is in registers
        fft64_quad<+1>(re, im, tc, ts, j);
        // through LDS (rows of 64 floats again) so that the result leaves as whole rows
        float* st = Xin + kwl * 4096 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = 16 * j + r;
            st[h * 64] = re[r] * 0.125f;
            st[h * 64 + 32] = im[r] * 0.125f;
        }
    }
    __syncthreads();
    if (mix_valid) store_rows32(Xin + (k2 * 64 + ih * 32) * 64, 64, T + cb0 + (k2 * 64 + ih * 32) * 64, lane);
}

__global__ __launch_bounds__(256, 3) void spec_mix_bwd4_kernel(const float* __restrict__ gR, const float* __restrict__ w2,
                                                               const float* __restrict__ S_save,
                                                               const unsigned long long* __restrict__ zmask, float* __restrict__ gT,
                                                               float* __restrict__ ws, int stg_d, int stg_m) {
    // One LDS image as in the forward kernel (DMA target -> operand image of gZ -> operand image of gS -> staging of the
    // result rows): three workgroups per CU.  The saved spectrum S, the second operand of the weight-gradient product, is
    // read straight into registers in MFMA fragment order (lane = channel, k-step = frequency pair: one dword per lane and
    // k-step, 128 contiguous bytes per 32 lanes) instead of through a second LDS buffer.
    __shared__ __attribute__((aligned(1024))) float Gs[2 * 64 * XLD4];
    __shared__ unsigned long long Zm[128];
    float* const Xin = Gs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, unit = b * gridDim.x + blockIdx.x;
    const bool two = (2 * blockIdx.x + 1) < NKW;
    const long long cb0 = ((long long)(b * NKW + 2 * blockIdx.x) * 64) * 64;
    float* slab = ws + (long long)unit * MIX_SLAB4;
    dma_columns(gR + cb0, Xin, wave, lane, two);
    if (tid < 128) Zm[tid] = zmask[(long long)unit * 128 + tid];
    stagger_wait(unit, stg_d, stg_m);
    const int f = tid >> 2, j = tid & 3, kwl = f >> 5, c = f & 31;
    const bool valid = kwl == 0 || two;
    const int l31 = lane & 31, kh2 = lane >> 5;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float re[16], im[16];
    // The saved spectrum of the first column as B fragments of the weight-gradient product (S comes from HBM: issued before
    // the transform so that the round trip is under its arithmetic; the mix weights, L2 hits, follow after it)
    const int io = wave >> 1, jk = wave & 1;        // weight gradient: wave (io, jk) owns one 32 x 32 block over both columns
    float sf[32];
    {
        const float* sp = S_save + cb0 + kh2 * 64 + jk * 32 + l31;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) sf[kk] = sp[2 * kk * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- column FFT of the cotangent, times the ReLU mask; bias-gradient partial sums
    {
        float tc[16], ts[16];
        quad_twiddles(j, tc, ts);
        const float* src = Xin + kwl * 4096 + j * 64 + c;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            re[m] = valid ? src[m * 256] : 0.f;
            im[m] = valid ? src[m * 256 + 32] : 0.f;
        }
        __syncthreads();                // every column is in registers: the image may be rewritten in operand layout
        fft64_quad<-1>(re, im, tc, ts, j);
        float dbr = 0.f, dbi = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kh = 16 * j + r;
            float gr = re[r] * 0.125f, gi = im[r] * 0.125f;
            const unsigned long long mr = Zm[mask_word(kwl, kh, c)], mi = Zm[mask_word(kwl, kh, 32 + c)];
            const int bit = mask_bit(kh, c);
            gr = (valid && ((mr >> bit) & 1ull)) ? gr : 0.f;
            gi = (valid && ((mi >> bit) & 1ull)) ? gi : 0.f;
            dbr += gr;
            dbi += gi;
            Gs[(kwl * 64 + kh) * XLD4 + c] = gr;
            Gs[(kwl * 64 + kh) * XLD4 + 32 + c] = gi;
        }
        // quad sum in a fixed order (lanes 0, 1, 2, 3 of the quad)
        const float sr = ((quad_bcast(dbr, 0) + quad_bcast(dbr, 1)) + quad_bcast(dbr, 2)) + quad_bcast(dbr, 3);
        const float si = ((quad_bcast(dbi, 0) + quad_bcast(dbi, 1)) + quad_bcast(dbi, 2)) + quad_bcast(dbi, 3);
        if (j == 0) {
            slab[64 * 64 + kwl * 64 + c] = sr;
            slab[64 * 64 + kwl * 64 + 32 + c] = si;
        }
    }
    const int k2 = wave >> 1, ih = wave & 1;        // data gradient: wave (k2, ih) owns rows k2*64 + ih*32 .. +31, both k halves
    const bool mix_valid = k2 == 0 || two;
    f32x16 accd[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) accd[jj][e] = 0.f;
    {
        float wf0[32], wf1[32];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            wf0[kk] = w2[(2 * kk + kh2) * 64 + l31];
            wf1[kk] = w2[(2 * kk + kh2) * 64 + 32 + l31];
        }
        __syncthreads();                                           // Gs complete
        // ---- data gradient  gS[f][k] = sum_o gZ[f][o] W2[o][k]
        if (mix_valid) {
            const float* arow = Gs + (k2 * 64 + ih * 32 + l31) * XLD4 + kh2;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float a0 = arow[2 * kk];
                accd[0] = mfma32(a0, wf0[kk], accd[0]);
                accd[1] = mfma32(a0, wf1[kk], accd[1]);
            }
        }
    }
    // ---- weight gradient  dW2[o][k] += sum_f gZ[f][o] S[f][k]
    {
        f32x16 accw;
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[e] = 0.f;
        float sg[32];
        if (two) {                                                 // second column's fragments, under the first column's MFMAs
            const float* sp = S_save + cb0 + 4096 + kh2 * 64 + jk * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) sg[kk] = sp[2 * kk * 64];
        }
        {
            const float* ga = Gs + kh2 * XLD4 + io * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) accw = mfma32(ga[2 * kk * XLD4], sf[kk], accw);
        }
        if (two) {
            const float* ga = Gs + (64 + kh2) * XLD4 + io * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) accw = mfma32(ga[2 * kk * XLD4], sg[kk], accw);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[(io * 32 + mfma32_row(e, lane)) * 64 + jk * 32 + l31] = accw[e];
    }
    __syncthreads();                                               // every read of Gs (both MFMA phases) is done
    if (mix_valid) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) Gs[(k2 * 64 + ih * 32 + mfma32_row(e, lane)) * XLD4 + jj * 32 + l31] = accd[jj][e];
    }
    __syncthreads();
    // ---- inverse column FFT of gS
    {
        float tc[16], ts[16];
        quad_twiddles(j, tc, ts);
        const float* src = Gs + (kwl * 64 + j) * XLD4 + c;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            re[m] = src[4 * m * XLD4];
            im[m] = src[4 * m * XLD4 + 32];
        }
        __syncthreads();                // every spectrum is in registers
        fft64_quad<+1>(re, im, tc, ts, j);
        const int kw = 2 * blockIdx.x + kwl;
        const float sc = (kw == 0 || kw == 32) ? 0.125f : 0.0625f;          // rfft2 backward: columns 1..31 halved
        float* st = Xin + kwl * 4096 + c;                                    // rows of 64 floats again
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int h = 16 * j + r;
            st[h * 64] = re[r] * sc;
            st[h * 64 + 32] = im[r] * sc;
        }
    }
    __syncthreads();
    if (mix_valid) store_rows32(Xin + (k2 * 64 + ih * 32) * 64, 64, gT + cb0 + (k2 * 64 + ih * 32) * 64, lane);
}

}  // namespace

// zmask: 128 64-bit words per unit (unit = patch * 17 + kw pair): the signs of the pre-activations, in the lane / register
// order of the forward kernel's accumulator blocks (mask_word / mask_bit above).  S_save and zmask may be NULL (no tape).
extern "C" size_t mtd_spec_mix_zmask_bytes(int B) { return B > 0 ? (size_t)B * 17 * 128 * sizeof(unsigned long long) : 0; }

extern "C" int mtd_spec_mix_fwd4(const float* R, const float* w2t, const float* b2, float* T, float* S_save, void* zmask, int B,
                                 void* stream) {
    if (!R || !w2t || !b2 || !T || B <= 0) return MTD_EINVAL;
    if (!aligned16(R)) return MTD_EALIGN;
    // one column per workgroup (1056 + B units of 128 threads: four per CU in flight) or a pair (544 of 256); MTD_SPECMIX_COLS
    static const int env_cols = [] { const char* e = mtd_lab_env("MTD_SPECMIX_COLS"); return e ? atoi(e) : 1; }();
    static const int stg_d = [] { const char* e = mtd_lab_env("MTD_SPECMIX_STAGGER"); return e ? atoi(e) : 0; }();
    static const int stg_m = [] { const char* e = mtd_lab_env("MTD_SPECMIX_STAGGER_MOD"); return e ? atoi(e) : 2; }();
    if (env_cols == 2)
        hipLaunchKernelGGL((spec_mix_fwd4_kernel<2>), dim3(17, B), dim3(256), 0, (hipStream_t)stream, R, w2t, b2, T, S_save,
                           (unsigned long long*)zmask, stg_d, stg_m);
    else
        hipLaunchKernelGGL((spec_mix_fwd4_kernel<1>), dim3(NKW, B), dim3(128), 0, (hipStream_t)stream, R, w2t, b2, T, S_save,
                           (unsigned long long*)zmask, stg_d, stg_m);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_spec_mix_bwd4(const float* gR, const float* w2, const float* S_save, const void* zmask, float* gT, float* ws, int B,
                                 void* stream) {
    if (!gR || !w2 || !S_save || !zmask || !gT || !ws || B <= 0) return MTD_EINVAL;
    if (!aligned16(gR) || !aligned16(S_save)) return MTD_EALIGN;
    static const int stg_d = [] { const char* e = mtd_lab_env("MTD_SPECMIX_STAGGER_BWD"); return e ? atoi(e) : 0; }();
    static const int stg_m = [] { const char* e = mtd_lab_env("MTD_SPECMIX_STAGGER_MOD"); return e ? atoi(e) : 2; }();
    hipLaunchKernelGGL(spec_mix_bwd4_kernel, dim3(17, B), dim3(256), 0, (hipStream_t)stream, gR, w2, S_save,
                       (const unsigned long long*)zmask, gT, ws, stg_d, stg_m);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
