// Spectral path of the Res-FFT-Conv block (arch/Ours/networks.py:21-30) for 64x64 patches, 32 ch.
//
//   y = irfft2( relu( W2 . [Re;Im] rfft2(x) + b2 ) ),  ortho normalisation both ways.
//
// The 2-D transform is split so that the channel mix -- which needs all 64 [Re;Im] channels of one
// frequency -- sits between the two column transforms of ONE kernel:
//   mtd_rfft_rows     rows  : real FFT along W.        x[B][64][64][32] -> R[B][33 kw][64 h][2][32]
//   mtd_spec_mix_fwd  cols  : FFT along H, 64x64 channel mix on fp32 MFMA (+bias, ReLU), IFFT along H.
//   mtd_irfft_rows    rows  : c2r along W (+ residual / conv-branch adds fused into the store).
// With NHWC data the 32 channels of a pixel are contiguous, so a lane owns one channel and performs a
// whole 64-point transform in registers (radix-2 DIF, fully unrolled, twiddles are immediates): no
// LDS exchange, no shuffles, and every global access is a 128-byte row of channels.  Two real rows
// are packed into one complex transform.  LDS is used only to re-shape spectra into MFMA operands.
// The backward pass reuses the same three kernels' structure:
//   irfft2 backward = w(kw) * rfft2(g)   (w = 1,2,...,2,1)      -> mtd_rfft_rows(col_weight=1)
//   rfft2  backward = c2r with columns 1..31 halved             -> halving folded into mtd_spec_mix_bwd
//
// Rooflines: rows kernels are HBM-bound (read 16 MiB + write 16.5 MiB per 32 patches);
// the column/mix kernel is fp32-MFMA-bound (2*64*64 flops per frequency) with VALU FFTs beside it.
#include "common.h"
#include "fft64.h"

// In-kernel phase stamps for the diagnostic build only (tools/specmix_stamp.hip defines MTD_STAMPS and includes this file).
#ifdef MTD_STAMPS
__device__ unsigned long long* rf_stamp_buf;
#define RF_STAMP(i)                                                                                         \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 8 && blockIdx.y < 8) {                                         \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            unsigned long long t__;                                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                      \
            rf_stamp_buf[(blockIdx.y * 8 + blockIdx.x) * 16 + (i)] = t__;                                   \
            __builtin_amdgcn_sched_barrier(0);                                                              \
        }                                                                                                   \
    } while (0)
#else
#define RF_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int XLD = 65;   // LDS row stride (floats) of the [frequency][64 channel] operand image

// ---------------------------------------------------------------------------------------------
// rows forward: two image rows (h, h+1) of one channel per thread (fft64.h rfft_rows_body)
__global__ __launch_bounds__(64) void rfft_rows_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ R,
                                                       int npairs, int col_weight) {
    rfft_rows_body(x, x_ld, R, npairs, col_weight, blockIdx.x * 2 + (threadIdx.x >> 5), threadIdx.x & 31);
}

// rows backward (c2r): two rows per thread, fused epilogue (compile-time variants so that the optional
// operand loads are branch-free and issued in batches of 8 pixels ahead of the stores)
template <bool A1, bool A2, bool MK>
__global__ __launch_bounds__(64) void irfft_rows_kernel(const float* __restrict__ T, float* __restrict__ out, int out_ld,
                                                        const float* __restrict__ add1, int add1_ld,
                                                        const float* __restrict__ add2, int add2_ld,
                                                        const float* __restrict__ mask, int mask_ld, int npairs) {
    const int c = threadIdx.x & 31;
    const int pair = blockIdx.x * 2 + (threadIdx.x >> 5);
    if (pair >= npairs) return;
    const int b = pair >> 5, h = (pair & 31) * 2;
    float re[64], im[64];
    const float* t0 = T + ((long long)(b * NKW) * 64 + h) * 64 + c;
#pragma unroll
    for (int kw = 0; kw <= 32; ++kw) {
        const float* t = t0 + (long long)kw * 64 * 64;
        const float ar = t[0], ai = t[32], br = t[64], bi = t[64 + 32];
        if (kw == 0 || kw == 32) {
            re[kw] = ar;       // imaginary parts of columns 0 and W/2 are ignored by c2r
            im[kw] = br;
        } else {
            re[kw] = ar - bi;
            im[kw] = ai + br;
            re[64 - kw] = ar + bi;
            im[64 - kw] = br - ai;
        }
    }
    fft64<+1>(re, im);
    const long long p0 = (long long)(b * 64 + h) * 64;
#pragma unroll
    for (int w0 = 0; w0 < 64; w0 += 8) {
        float x1a[8], x1b[8], x2a[8], x2b[8], mka[8], mkb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long long pa = p0 + w0 + j, pb = pa + 64;
            if (A1) { x1a[j] = add1[pa * add1_ld + c]; x1b[j] = add1[pb * add1_ld + c]; }
            if (A2) { x2a[j] = add2[pa * add2_ld + c]; x2b[j] = add2[pb * add2_ld + c]; }
            if (MK) { mka[j] = mask[pa * mask_ld + c]; mkb[j] = mask[pb * mask_ld + c]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int w = w0 + j;
            float va = re[brev6(w)] * 0.125f, vb = im[brev6(w)] * 0.125f;
            const long long pa = p0 + w, pb = pa + 64;
            if (A1) { va += x1a[j]; vb += x1b[j]; }
            if (A2) { va += x2a[j]; vb += x2b[j]; }
            if (MK) { va = mka[j] > 0.f ? va : 0.f; vb = mkb[j] > 0.f ? vb : 0.f; }
            out[pa * out_ld + c] = va;
            out[pb * out_ld + c] = vb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// columns + channel mix, forward.  One wave per workgroup, two kw columns per wave.
__global__ __launch_bounds__(64) void spec_mix_fwd_kernel(const float* __restrict__ R, const float* __restrict__ w2t,
                                                          const float* __restrict__ b2, float* __restrict__ T,
                                                          float* __restrict__ S_save, float* __restrict__ Z_save) {
    __shared__ float Xs[2 * 64 * XLD];
    const int lane = threadIdx.x, kwl = lane >> 5, c = lane & 31, l31 = lane & 31, kh2 = lane >> 5;
    const int b = blockIdx.y;
    const int kw = 2 * blockIdx.x + kwl;
    const bool valid = kw < NKW;
    RF_STAMP(0);
    float re[64], im[64];
    const long long colbase = ((long long)(b * NKW + (valid ? kw : 0)) * 64) * 64;
    {
        const float* src = R + colbase + c;
#pragma unroll
        for (int h = 0; h < 64; ++h) {
            re[h] = valid ? src[h * 64] : 0.f;
            im[h] = valid ? src[h * 64 + 32] : 0.f;
        }
    }
    RF_STAMP(1);
    fft64<-1>(re, im);
    RF_STAMP(2);
#pragma unroll
    for (int kh = 0; kh < 64; ++kh) {
        const float sr = re[brev6(kh)] * 0.125f, si = im[brev6(kh)] * 0.125f;
        Xs[(kwl * 64 + kh) * XLD + c] = sr;
        Xs[(kwl * 64 + kh) * XLD + 32 + c] = si;
        if (S_save && valid) {
            S_save[colbase + kh * 64 + c] = sr;
            S_save[colbase + kh * 64 + 32 + c] = si;
        }
    }
    __syncthreads();
    RF_STAMP(3);
    // the mix weights as MFMA B fragments: 64 registers, one batch of loads (re / im are dead from here to the inverse FFT)
    float wf0[32], wf1[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
        wf0[kk] = w2t[(2 * kk + kh2) * 64 + l31];
        wf1[kk] = w2t[(2 * kk + kh2) * 64 + 32 + l31];
    }
    RF_STAMP(4);
#pragma unroll 1
    for (int k2 = 0; k2 < 2; ++k2) {
        const int kw2 = 2 * blockIdx.x + k2;
        if (kw2 >= NKW) break;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 2 * kk + kh2;
            const float a0 = Xs[(k2 * 64 + l31) * XLD + k];
            const float a1 = Xs[(k2 * 64 + 32 + l31) * XLD + k];
            const float b0 = wf0[kk];
            const float b1 = wf1[kk];
            acc[0][0] = mfma32(a0, b0, acc[0][0]);
            acc[0][1] = mfma32(a0, b1, acc[0][1]);
            acc[1][0] = mfma32(a1, b0, acc[1][0]);
            acc[1][1] = mfma32(a1, b1, acc[1][1]);
        }
        __syncthreads();   // all operand reads of this column's rows are done before they are overwritten
        const long long cb2 = ((long long)(b * NKW + kw2) * 64) * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = j * 32 + l31;
            const float bo = b2[o];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kh = i * 32 + mfma32_row(e, lane);
                    const float z = acc[i][j][e] + bo;
                    if (Z_save) Z_save[cb2 + kh * 64 + o] = z;
                    Xs[(k2 * 64 + kh) * XLD + o] = z > 0.f ? z : 0.f;
                }
        }
    }
    __syncthreads();
    RF_STAMP(5);
#pragma unroll
    for (int kh = 0; kh < 64; ++kh) {
        re[kh] = Xs[(kwl * 64 + kh) * XLD + c];
        im[kh] = Xs[(kwl * 64 + kh) * XLD + 32 + c];
    }
    RF_STAMP(6);
    fft64<+1>(re, im);
    RF_STAMP(7);
    if (valid) {
        float* dst = T + colbase + c;
#pragma unroll
        for (int h = 0; h < 64; ++h) {
            dst[h * 64] = re[brev6(h)] * 0.125f;
            dst[h * 64 + 32] = im[brev6(h)] * 0.125f;
        }
    }
    RF_STAMP(8);
}

constexpr int MIX_SLAB = 64 * 64 + 128;   // dW2 partial + two db2 partial rows per workgroup

// columns + channel mix, backward.  One wave per workgroup: every global read that is not the column data itself is
// taken off the critical path -- the two saved pre-activation columns (only their signs are needed) and the saved
// spectrum of the column being reduced stream into LDS with global_load_lds while the column FFT runs in registers,
// and the 64x64 mix weights sit in 64 registers as ready MFMA fragments.
typedef __attribute__((address_space(3))) float lds_float;

__device__ __forceinline__ void glds_tile16k(const float* gsrc, float* lds_dst, int lane) {
    // 16 KiB contiguous global -> 16 KiB contiguous LDS: 16 wave instructions of 64 lanes x 16 bytes
#pragma unroll
    for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds(gsrc + i * 256 + lane * 4, (lds_float*)(lds_dst + i * 256), 16, 0, 0);
}

__global__ __launch_bounds__(64) void spec_mix_bwd_kernel(const float* __restrict__ gR, const float* __restrict__ w2,
                                                          const float* __restrict__ S_save, const float* __restrict__ Z_save,
                                                          float* __restrict__ gT, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float Gs[2 * 64 * XLD];
    __shared__ __attribute__((aligned(16))) float Ss[64 * 64];
    const int lane = threadIdx.x, kwl = lane >> 5, c = lane & 31, l31 = lane & 31, kh2 = lane >> 5;
    const int b = blockIdx.y;
    const int kw = 2 * blockIdx.x + kwl;
    const bool valid = kw < NKW;
    const bool two = (2 * blockIdx.x + 1) < NKW;            // workgroup-uniform: does the second column exist
    float* slab = ws + ((long long)b * gridDim.x + blockIdx.x) * MIX_SLAB;
    float re[64], im[64];
    const long long colbase = ((long long)(b * NKW + (valid ? kw : 0)) * 64) * 64;
    const long long cb0 = ((long long)(b * NKW + 2 * blockIdx.x) * 64) * 64;
    // async: Z of column 0 -> Gs[0 ..), Z of column 1 -> Gs[64*XLD ..) (the bases of the two G images written later, so
    // that the in-place sweep below never overwrites a Z row it still has to read), S of column 0 -> Ss
    glds_tile16k(Z_save + cb0, Gs, lane);
    if (two) glds_tile16k(Z_save + cb0 + 4096, Gs + 64 * XLD, lane);
    glds_tile16k(S_save + cb0, Ss, lane);
    {
        const float* src = gR + colbase + c;
#pragma unroll
        for (int h = 0; h < 64; ++h) {
            re[h] = valid ? src[h * 64] : 0.f;
            im[h] = valid ? src[h * 64 + 32] : 0.f;
        }
    }
    fft64<-1>(re, im);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float dbr = 0.f, dbi = 0.f;
    const int zbase = kwl * 64 * XLD;
    // descending kh: G row kh (stride 65) lands on or above Z row kh (stride 64) and never on a lower one
#pragma unroll
    for (int kh = 63; kh >= 0; --kh) {
        float gr = re[brev6(kh)] * 0.125f, gi = im[brev6(kh)] * 0.125f;
        const float zr = valid ? Gs[zbase + kh * 64 + c] : 0.f;
        const float zi = valid ? Gs[zbase + kh * 64 + 32 + c] : 0.f;
        __builtin_amdgcn_wave_barrier();
        gr = zr > 0.f ? gr : 0.f;
        gi = zi > 0.f ? gi : 0.f;
        dbr += gr;
        dbi += gi;
        Gs[(kwl * 64 + kh) * XLD + c] = gr;
        Gs[(kwl * 64 + kh) * XLD + 32 + c] = gi;
        __builtin_amdgcn_wave_barrier();
    }
    slab[64 * 64 + kwl * 64 + c] = dbr;
    slab[64 * 64 + kwl * 64 + 32 + c] = dbi;
    // mix weights as MFMA B fragments for the data gradient (re / im are dead until the inverse FFT)
    float wf0[32], wf1[32];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
        wf0[kk] = w2[(2 * kk + kh2) * 64 + l31];
        wf1[kk] = w2[(2 * kk + kh2) * 64 + 32 + l31];
    }
    __syncthreads();
    f32x16 accw[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) accw[i][j][e] = 0.f;
#pragma unroll 1
    for (int k2 = 0; k2 < 2; ++k2) {
        const int kw2 = 2 * blockIdx.x + k2;
        if (kw2 >= NKW) break;
        f32x16 accd[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) accd[i][j][e] = 0.f;
        // data gradient  gS[f][k] = sum_o gZ[f][o] W2[o][k]
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int o = 2 * kk + kh2;
            const float a0 = Gs[(k2 * 64 + l31) * XLD + o];
            const float a1 = Gs[(k2 * 64 + 32 + l31) * XLD + o];
            accd[0][0] = mfma32(a0, wf0[kk], accd[0][0]);
            accd[0][1] = mfma32(a0, wf1[kk], accd[0][1]);
            accd[1][0] = mfma32(a1, wf0[kk], accd[1][0]);
            accd[1][1] = mfma32(a1, wf1[kk], accd[1][1]);
        }
        if (k2 == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // S of the second column has landed in Ss
        // weight gradient  dW2[o][k] += sum_f gZ[f][o] S[f][k]
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const int f = 2 * kk + kh2;
            const float a0 = Gs[(k2 * 64 + f) * XLD + l31];
            const float a1 = Gs[(k2 * 64 + f) * XLD + 32 + l31];
            const float b0 = Ss[f * 64 + l31];
            const float b1 = Ss[f * 64 + 32 + l31];
            accw[0][0] = mfma32(a0, b0, accw[0][0]);
            accw[0][1] = mfma32(a0, b1, accw[0][1]);
            accw[1][0] = mfma32(a1, b0, accw[1][0]);
            accw[1][1] = mfma32(a1, b1, accw[1][1]);
        }
        __syncthreads();
        if (k2 == 0 && two) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // every read of Ss has returned
            glds_tile16k(S_save + cb0 + 4096, Ss, lane);                     // lands under the next column's data gradient
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kh = i * 32 + mfma32_row(e, lane);
                    Gs[(k2 * 64 + kh) * XLD + j * 32 + l31] = accd[i][j][e];
                }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[(i * 32 + mfma32_row(e, lane)) * 64 + j * 32 + l31] = accw[i][j][e];
#pragma unroll
    for (int kh = 0; kh < 64; ++kh) {
        re[kh] = Gs[(kwl * 64 + kh) * XLD + c];
        im[kh] = Gs[(kwl * 64 + kh) * XLD + 32 + c];
    }
    fft64<+1>(re, im);
    if (valid) {
        const float sc = (kw == 0 || kw == 32) ? 0.125f : 0.0625f;   // rfft2 backward: columns 1..31 halved
        float* dst = gT + colbase + c;
#pragma unroll
        for (int h = 0; h < 64; ++h) {
            dst[h * 64] = re[brev6(h)] * sc;
            dst[h * 64 + 32] = im[brev6(h)] * sc;
        }
    }
}

// stage 1: out[g][idx] = sum over slabs of group g ; stage 2 (final) folds the two bias rows
__global__ __launch_bounds__(256) void mix_slab_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int nslab, int gs) {
    const int grp = blockIdx.y;
    const int s0 = grp * gs, s1 = min(nslab, s0 + gs);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < MIX_SLAB; idx += gridDim.x * 256) {
        float s = 0.f;
        for (int k = s0; k < s1; ++k) s += in[(long long)k * MIX_SLAB + idx];
        out[(long long)grp * MIX_SLAB + idx] = s;
    }
}
__global__ __launch_bounds__(256) void mix_finish_kernel(const float* __restrict__ in, int nslab, float* __restrict__ dw2,
                                                         float* __restrict__ db2, int accumulate) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 64 * 64 + 64; idx += gridDim.x * 256) {
        float s = 0.f;
        if (idx < 4096) {
            for (int k = 0; k < nslab; ++k) s += in[(long long)k * MIX_SLAB + idx];
            dw2[idx] = accumulate ? dw2[idx] + s : s;
        } else {
            const int o = idx - 4096;
            for (int k = 0; k < nslab; ++k) s += in[(long long)k * MIX_SLAB + 4096 + o] + in[(long long)k * MIX_SLAB + 4096 + 64 + o];
            db2[o] = accumulate ? db2[o] + s : s;
        }
    }
}

// both reduce stages and the accumulate in one launch: workgroups 0..63 own 16 float4 of dW2 each, workgroup 64 the bias
// gradient (the sum of the slab's two db2 rows); 1024 threads = 16 columns x 64 slab runs (common.h block_slab_sum)
__device__ __forceinline__ void mix_reduce_block(const float* __restrict__ in, int nslab, float* __restrict__ dw2,
                                                 float* __restrict__ db2, int accumulate, int blk, f32x4* red) {
    const int tx = threadIdx.x & 15;
    const bool lead = (threadIdx.x >> 4) == 0;
    if (blk < 64) {
        const int i4 = blk * 16 + tx;
        const f32x4 s = block_slab_sum<8>(in, MIX_SLAB, nslab, i4, true, red);
        if (lead) {
            f32x4* dst = reinterpret_cast<f32x4*>(dw2) + i4;
            *dst = accumulate ? (*dst + s) : s;
        }
    } else {
        const f32x4 s1 = block_slab_sum<8>(in + 4096, MIX_SLAB, nslab, tx, true, red);
        const f32x4 s2 = block_slab_sum<8>(in + 4096 + 64, MIX_SLAB, nslab, tx, true, red);
        if (lead) {
            const f32x4 s = s1 + s2;
#pragma unroll
            for (int j = 0; j < 4; ++j) db2[4 * tx + j] = accumulate ? (db2[4 * tx + j] + s[j]) : s[j];
        }
    }
}

__global__ __launch_bounds__(1024) void mix_reduce_finish_kernel(const float* __restrict__ in, int nslab, float* __restrict__ dw2,
                                                                 float* __restrict__ db2, int accumulate) {
    __shared__ f32x4 red[16 * 8 * 9];
    mix_reduce_block(in, nslab, dw2, db2, accumulate, blockIdx.x, red);
}

// the slab sets of many Res-FFT blocks in one launch: blockIdx.y = block of the network
__global__ __launch_bounds__(1024) void mix_reduce_multi_kernel(const mtd_mix_reduce_desc* __restrict__ table) {
    __shared__ f32x4 red[16 * 8 * 9];
    const mtd_mix_reduce_desc& d = table[blockIdx.y];
    mix_reduce_block(d.ws, d.nslab, d.dw2, d.db2, d.accumulate, blockIdx.x, red);
}

__global__ __launch_bounds__(256) void transpose64_kernel(const float* __restrict__ src, float* __restrict__ dst) {
    __shared__ float t[64 * 65];
    for (int i = threadIdx.x; i < 4096; i += 256) t[(i >> 6) * 65 + (i & 63)] = src[i];
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) dst[i] = t[(i & 63) * 65 + (i >> 6)];
}

// all blocks' mix weights in one launch: ptrs = [src_0, dst_0, src_1, dst_1, ...]
__global__ __launch_bounds__(256) void transpose64_multi_kernel(const float* const* __restrict__ ptrs) {
    __shared__ float t[64 * 65];
    const float* src = ptrs[2 * blockIdx.x];
    float* dst = const_cast<float*>(ptrs[2 * blockIdx.x + 1]);
    for (int i = threadIdx.x; i < 4096; i += 256) t[(i >> 6) * 65 + (i & 63)] = src[i];
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) dst[i] = t[(i & 63) * 65 + (i >> 6)];
}

constexpr int MIX_GS = 32;

}  // namespace

extern "C" int mtd_rfft_rows(const float* x, int x_ld, float* R, int B, int col_weight, void* stream) {
    if (!x || !R || B <= 0 || x_ld < 32) return MTD_EINVAL;
    const int npairs = B * 32;
    // one wave (two row pairs x 32 channels) per workgroup: 512 workgroups at B = 32 so that every CU is fed
    hipLaunchKernelGGL(rfft_rows_kernel, dim3((npairs + 1) / 2), dim3(64), 0, (hipStream_t)stream, x, x_ld, R, npairs, col_weight);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_irfft_rows(const float* T, float* out, int out_ld, const float* add1, int add1_ld, const float* add2,
                              int add2_ld, const float* mask, int mask_ld, int B, void* stream) {
    if (!T || !out || B <= 0 || out_ld < 32) return MTD_EINVAL;
    if ((add1 && add1_ld < 32) || (add2 && add2_ld < 32) || (mask && mask_ld < 32)) return MTD_EINVAL;
    const int npairs = B * 32;
    const dim3 grid((npairs + 1) / 2), blk(64);
    hipStream_t s = (hipStream_t)stream;
#define MTD_IRFFT(A, B2, M) hipLaunchKernelGGL((irfft_rows_kernel<A, B2, M>), grid, blk, 0, s, T, out, out_ld, add1, add1_ld, add2, add2_ld, mask, mask_ld, npairs)
    const int variant = (add1 ? 1 : 0) | (add2 ? 2 : 0) | (mask ? 4 : 0);
    switch (variant) {
        case 0: MTD_IRFFT(false, false, false); break;
        case 1: MTD_IRFFT(true, false, false); break;
        case 2: MTD_IRFFT(false, true, false); break;
        case 3: MTD_IRFFT(true, true, false); break;
        case 4: MTD_IRFFT(false, false, true); break;
        case 5: MTD_IRFFT(true, false, true); break;
        case 6: MTD_IRFFT(false, true, true); break;
        default: MTD_IRFFT(true, true, true); break;
    }
#undef MTD_IRFFT
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_spec_mix_fwd(const float* R, const float* w2t, const float* b2, float* T, float* S_save, float* Z_save, int B,
                                void* stream) {
    if (!R || !w2t || !b2 || !T || B <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(spec_mix_fwd_kernel, dim3(17, B), dim3(64), 0, (hipStream_t)stream, R, w2t, b2, T, S_save, Z_save);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

static size_t mix_ws_floats(int B) {
    long long ns = (long long)B * 17;
    long long total = ns * MIX_SLAB;
    while (ns > MIX_GS) {
        ns = (ns + MIX_GS - 1) / MIX_GS;
        total += ns * MIX_SLAB;
    }
    return (size_t)total;
}

extern "C" size_t mtd_spec_mix_bwd_ws_bytes(int B) { return B > 0 ? mix_ws_floats(B) * sizeof(float) : 0; }

extern "C" int mtd_spec_mix_bwd(const float* gR, const float* w2, const float* S_save, const float* Z_save, float* gT, float* ws,
                                int B, void* stream) {
    if (!gR || !w2 || !S_save || !Z_save || !gT || !ws || B <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(spec_mix_bwd_kernel, dim3(17, B), dim3(64), 0, (hipStream_t)stream, gR, w2, S_save, Z_save, gT, ws);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_spec_mix_wgrad_reduce(const float* ws, int B, float* dw2, float* db2, int accumulate, void* stream) {
    if (!ws || !dw2 || !db2 || B <= 0) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int ns = B * 17;
    const float* cur = ws;
    static const int env_fused = [] { const char* e = mtd_lab_env("MTD_MIX_FUSED_REDUCE"); return e ? atoi(e) : 1; }();
    if (env_fused && ns <= 4096 && aligned16(ws) && aligned16(dw2)) {
        hipLaunchKernelGGL(mix_reduce_finish_kernel, dim3(65), dim3(1024), 0, s, cur, ns, dw2, db2, accumulate);
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    float* next = const_cast<float*>(ws) + (long long)ns * MIX_SLAB;
    while (ns > MIX_GS) {
        int ng = (ns + MIX_GS - 1) / MIX_GS;
        hipLaunchKernelGGL(mix_slab_sum_kernel, dim3((MIX_SLAB + 255) / 256, ng), dim3(256), 0, s, cur, next, ns, MIX_GS);
        MTD_LAUNCH_CHECK();
        cur = next;
        next += (long long)ng * MIX_SLAB;
        ns = ng;
    }
    hipLaunchKernelGGL(mix_finish_kernel, dim3((4096 + 64 + 255) / 256), dim3(256), 0, s, cur, ns, dw2, db2, accumulate);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_spec_mix_wgrad_reduce_multi(const mtd_mix_reduce_desc* table_dev, const mtd_mix_reduce_desc* table_host, int count,
                                               void* stream) {
    if (!table_dev || !table_host || count <= 0 || count > 65535) return MTD_EINVAL;
    for (int i = 0; i < count; ++i) {
        const mtd_mix_reduce_desc& d = table_host[i];
        if (!d.ws || !d.dw2 || !d.db2 || d.nslab <= 0 || d.nslab > 4096) return MTD_EINVAL;
        if (!aligned16(d.ws) || !aligned16(d.dw2)) return MTD_EALIGN;
    }
    hipLaunchKernelGGL(mix_reduce_multi_kernel, dim3(65, count), dim3(1024), 0, (hipStream_t)stream, table_dev);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_transpose64_multi(const float* const* ptrs_dev, int n, void* stream) {
    if (!ptrs_dev || n <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(transpose64_multi_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ptrs_dev);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_transpose64(const float* src, float* dst, void* stream) {
    if (!src || !dst) return MTD_EINVAL;
    hipLaunchKernelGGL(transpose64_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, dst);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
