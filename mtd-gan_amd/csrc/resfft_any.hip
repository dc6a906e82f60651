// Res-FFT-Conv spectral path for square maps of any power-of-two size 128 .. 512 (inference on whole slices:
// reference engine.py:89,129 runs the generator on 512 x 512 images, where rfft2 is a 512-point transform).
// The 64 x 64 training path keeps its register-resident kernels (resfft.hip); here one workgroup owns one image line
// (a row, or a frequency column) for all 32 channels and transforms it in LDS: [S points][32 channels] re + im = S * 256
// bytes (128 KB at S = 512, 160 KB per CU), radix-2, one barrier per stage, lane = channel so every LDS access is
// stride-1 across lanes.  Forward transforms are decimation-in-frequency (natural in, bit-reversed out), inverse ones
// decimation-in-time (bit-reversed in, natural out): the 1x1 spectral conv between them is per frequency, so the
// column kernel never reorders anything.  Spectra: [B][kw 0..S/2][h 0..S-1][Re 32 | Im 32], ortho scaling 1/sqrt(S)
// per dimension.  HBM-bound in principle (two passes over 3 x the activation size per block); these kernels are the
// simple, exact version -- the work of a slice is dominated by the 3x3 convolutions.
#include "common.h"

namespace {

__device__ __forceinline__ int brev_n(int k, int logS) { return (int)(__brev((unsigned)k) >> (32 - logS)); }

// twiddle table exp(-i * pi * k / (S/2)), k = 0 .. S/2-1, written once per workgroup (cos | sin)
__device__ __forceinline__ void fill_twiddles(float* tw, int S) {
    const int half = S >> 1;
    const float inv = 1.f / (float)half;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float sn, cs;
        sincospif((float)k * inv, &sn, &cs);
        tw[k] = cs;
        tw[half + k] = sn;
    }
}

// in-place FFT of re/im[S][32]; every thread of the workgroup takes part.  SIGN -1: forward, +1: inverse.
// Two radix-2 stages are fused per pass over LDS (four points in registers: half the LDS traffic and half the barriers
// of a stage-by-stage radix-2); an odd log2(S) leaves one single stage.
template <int SIGN>
__device__ __forceinline__ void twiddle(const float* tw, int hS, int idx, float& cs, float& sn) {
    cs = tw[idx];
    sn = (SIGN < 0) ? -tw[hS + idx] : tw[hS + idx];
}

template <int SIGN, bool DIF>
__device__ __forceinline__ void lds_fft(float* re, float* im, const float* tw, int S, int logS) {
    const int hS = S >> 1;
    int st = 0;
    // single radix-2 stage when log2(S) is odd: the first stage for DIF (half = S/2), the first for DIT (half = 1)
    if (logS & 1) {
        const int lh = DIF ? (logS - 1) : 0;
        const int half = 1 << lh, tshift = logS - 1 - lh;
        for (int e = threadIdx.x; e < hS * 32; e += blockDim.x) {
            const int c = e & 31, pidx = e >> 5;
            const int grp = pidx >> lh, j = pidx & (half - 1);
            const int i0 = (((grp << 1) << lh) + j) * 32 + c, i1 = i0 + half * 32;
            float cs, sn;
            twiddle<SIGN>(tw, hS, j << tshift, cs, sn);
            const float ur = re[i0], ui = im[i0], vr = re[i1], vi = im[i1];
            if (DIF) {
                const float dr = ur - vr, di = ui - vi;
                re[i0] = ur + vr; im[i0] = ui + vi;
                re[i1] = dr * cs - di * sn; im[i1] = dr * sn + di * cs;
            } else {
                const float wr = vr * cs - vi * sn, wi = vr * sn + vi * cs;
                re[i0] = ur + wr; im[i0] = ui + wi;
                re[i1] = ur - wr; im[i1] = ui - wi;
            }
        }
        __syncthreads();
        st = 1;
    }
    const int nq = (S >> 2) * 32;                     // 4-point groups per fused pass (x channel)
    for (; st < logS; st += 2) {
        if (DIF) {
            // stages with half = H and H/2;  points a, b = a + H/2, c = a + H, d = a + 3H/2 of a block of 2H
            const int lH = logS - 1 - st;             // log2(H)
            const int H = 1 << lH, Q = H >> 1;
            const int ts1 = logS - 1 - lH, ts2 = ts1 + 1;
            for (int e = threadIdx.x; e < nq; e += blockDim.x) {
                const int c = e & 31, q = e >> 5;
                const int blk = q >> (lH - 1), j = q & (Q - 1);
                const int ia = ((blk << (lH + 1)) + j) * 32 + c, ib = ia + Q * 32, ic = ia + H * 32, id = ic + Q * 32;
                float c1, s1, c2, s2, c3, s3;
                twiddle<SIGN>(tw, hS, j << ts1, c1, s1);
                twiddle<SIGN>(tw, hS, (j + Q) << ts1, c2, s2);
                twiddle<SIGN>(tw, hS, j << ts2, c3, s3);
                const float ar = re[ia], ai = im[ia], br = re[ib], bi = im[ib], cr = re[ic], ci = im[ic], dr = re[id], di = im[id];
                const float a1r = ar + cr, a1i = ai + ci, tr = ar - cr, ti = ai - ci;
                const float c1r = tr * c1 - ti * s1, c1i = tr * s1 + ti * c1;
                const float b1r = br + dr, b1i = bi + di, ur = br - dr, ui = bi - di;
                const float d1r = ur * c2 - ui * s2, d1i = ur * s2 + ui * c2;
                re[ia] = a1r + b1r; im[ia] = a1i + b1i;
                { const float xr = a1r - b1r, xi = a1i - b1i; re[ib] = xr * c3 - xi * s3; im[ib] = xr * s3 + xi * c3; }
                re[ic] = c1r + d1r; im[ic] = c1i + d1i;
                { const float xr = c1r - d1r, xi = c1i - d1i; re[id] = xr * c3 - xi * s3; im[id] = xr * s3 + xi * c3; }
            }
        } else {
            // stages with half = h and 2h;  points a, b = a + h, c = a + 2h, d = a + 3h of a block of 4h
            const int lh = st;
            const int h = 1 << lh;
            const int ts1 = logS - 1 - lh, ts2 = ts1 - 1;
            for (int e = threadIdx.x; e < nq; e += blockDim.x) {
                const int c = e & 31, q = e >> 5;
                const int blk = q >> lh, j = q & (h - 1);
                const int ia = ((blk << (lh + 2)) + j) * 32 + c, ib = ia + h * 32, ic = ib + h * 32, id = ic + h * 32;
                float c1, s1, c2, s2, c3, s3;
                twiddle<SIGN>(tw, hS, j << ts1, c1, s1);
                twiddle<SIGN>(tw, hS, j << ts2, c2, s2);
                twiddle<SIGN>(tw, hS, (j + h) << ts2, c3, s3);
                const float ar = re[ia], ai = im[ia], br = re[ib], bi = im[ib], cr = re[ic], ci = im[ic], dr = re[id], di = im[id];
                const float bwr = br * c1 - bi * s1, bwi = br * s1 + bi * c1;
                const float dwr = dr * c1 - di * s1, dwi = dr * s1 + di * c1;
                const float a1r = ar + bwr, a1i = ai + bwi, b1r = ar - bwr, b1i = ai - bwi;
                const float c1r = cr + dwr, c1i = ci + dwi, d1r = cr - dwr, d1i = ci - dwi;
                const float cwr = c1r * c2 - c1i * s2, cwi = c1r * s2 + c1i * c2;
                const float ewr = d1r * c3 - d1i * s3, ewi = d1r * s3 + d1i * c3;
                re[ia] = a1r + cwr; im[ia] = a1i + cwi;
                re[ic] = a1r - cwr; im[ic] = a1i - cwi;
                re[ib] = b1r + ewr; im[ib] = b1i + ewi;
                re[id] = b1r - ewr; im[id] = b1i - ewi;
            }
        }
        __syncthreads();
    }
}

// rows forward: one workgroup per PAIR of image rows (b, h), (b, h + 1): the two real rows are the real and imaginary
// part of one complex transform Z; X_h[k] = (Z[k] + conj Z[-k]) / 2, X_{h+1}[k] = (Z[k] - conj Z[-k]) / 2i
__global__ __launch_bounds__(1024) void rfft_rows_any_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ R, int S, int logS) {
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + S * 32;
    float* tw = lds + S * 64;
    fill_twiddles(tw, S);
    const int hp = S >> 1;
    const int b = blockIdx.x / hp, h = (blockIdx.x % hp) * 2;
    const int nkw = S / 2 + 1;
    const float* src = x + ((long long)(b * S + h) * S) * x_ld;
    for (int e = threadIdx.x; e < S * 32; e += blockDim.x) {
        const int c = e & 31, w = e >> 5;
        re[e] = src[(long long)w * x_ld + c];
        im[e] = src[(long long)(S + w) * x_ld + c];
    }
    __syncthreads();
    lds_fft<-1, true>(re, im, tw, S, logS);
    const float sc = 0.5f * rsqrtf((float)S);
    for (int e = threadIdx.x; e < nkw * 32; e += blockDim.x) {
        const int c = e & 31, kw = e >> 5;
        const int pk = brev_n(kw, logS) * 32 + c, pm = brev_n((S - kw) & (S - 1), logS) * 32 + c;
        const float zkr = re[pk], zki = im[pk], zmr = re[pm], zmi = im[pm];
        float* o = R + (((long long)(b * nkw + kw) * S + h) * 64) + c;
        o[0] = (zkr + zmr) * sc;
        o[32] = (zki - zmi) * sc;
        o[64] = (zki + zmi) * sc;
        o[64 + 32] = (zmr - zkr) * sc;
    }
}

// columns + channel mix + columns back: one workgroup per (b, kw)
__global__ __launch_bounds__(1024) void spec_mix_any_kernel(const float* __restrict__ R, const float* __restrict__ w2t,
                                                           const float* __restrict__ b2, float* __restrict__ T, int S, int logS) {
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + S * 32;
    float* tw = lds + S * 64;
    fill_twiddles(tw, S);
    const long long colbase = (long long)blockIdx.x * S * 64;
    for (int e = threadIdx.x; e < S * 32; e += blockDim.x) {
        const int c = e & 31, h = e >> 5;
        re[e] = R[colbase + (long long)h * 64 + c];
        im[e] = R[colbase + (long long)h * 64 + 32 + c];
    }
    __syncthreads();
    lds_fft<-1, true>(re, im, tw, S, logS);
    // channel mix at every frequency (order of the frequencies is irrelevant): lane = output channel o, its 64 weights in
    // registers; a wave owns whole rows, reads all 64 inputs of a row (broadcast reads) before it writes the 64 outputs
    {
        const int o = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
        float wreg[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) wreg[k] = w2t[k * 64 + o];
        const float bo = b2[o];
        const float sc = rsqrtf((float)S);
        for (int n = wv; n < S; n += nwv) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) acc = fmaf(re[n * 32 + k], wreg[k], acc);
#pragma unroll
            for (int k = 0; k < 32; ++k) acc = fmaf(im[n * 32 + k], wreg[32 + k], acc);
            const float z = fmaxf(acc * sc + bo, 0.f);
            __builtin_amdgcn_wave_barrier();
            if (o < 32) re[n * 32 + o] = z;
            else im[n * 32 + o - 32] = z;
        }
    }
    __syncthreads();
    lds_fft<+1, false>(re, im, tw, S, logS);
    const float sc = rsqrtf((float)S);
    for (int e = threadIdx.x; e < S * 32; e += blockDim.x) {
        const int c = e & 31, h = e >> 5;
        T[colbase + (long long)h * 64 + c] = re[e] * sc;
        T[colbase + (long long)h * 64 + 32 + c] = im[e] * sc;
    }
}

// rows back (c2r): one workgroup per pair of image rows; out = y + add1 + add2.  With A = X_h, B = X_{h+1} (Hermitian,
// the imaginary parts of columns 0 and S/2 ignored as torch's c2r does) the complex spectrum Z = A + iB transforms
// back to row h in the real part and row h + 1 in the imaginary part.
__global__ __launch_bounds__(1024) void irfft_rows_any_kernel(const float* __restrict__ T, float* __restrict__ out, int out_ld,
                                                              const float* __restrict__ add1, int add1_ld,
                                                              const float* __restrict__ add2, int add2_ld, int S, int logS) {
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + S * 32;
    float* tw = lds + S * 64;
    fill_twiddles(tw, S);
    const int hp = S >> 1;
    const int b = blockIdx.x / hp, h = (blockIdx.x % hp) * 2;
    const int nkw = S / 2 + 1;
    for (int e = threadIdx.x; e < nkw * 32; e += blockDim.x) {
        const int c = e & 31, kw = e >> 5;
        const float* t = T + (((long long)(b * nkw + kw) * S + h) * 64) + c;
        const bool edge = (kw == 0 || kw == S / 2);
        const float ar = t[0], ai = edge ? 0.f : t[32];
        const float br = t[64], bi = edge ? 0.f : t[64 + 32];
        const int p0 = brev_n(kw, logS) * 32 + c;
        re[p0] = ar - bi;
        im[p0] = ai + br;
        if (!edge) {
            const int p1 = brev_n(S - kw, logS) * 32 + c;       // Z[S-k] = conj(A[k]) + i conj(B[k])
            re[p1] = ar + bi;
            im[p1] = br - ai;
        }
    }
    __syncthreads();
    lds_fft<+1, false>(re, im, tw, S, logS);
    const float sc = rsqrtf((float)S);
    const long long rowpix = (long long)(b * S + h) * S;
    for (int e = threadIdx.x; e < S * 32; e += blockDim.x) {
        const int c = e & 31, w = e >> 5;
        float v0 = re[e] * sc, v1 = im[e] * sc;
        if (add1) { v0 += add1[(rowpix + w) * add1_ld + c]; v1 += add1[(rowpix + S + w) * add1_ld + c]; }
        if (add2) { v0 += add2[(rowpix + w) * add2_ld + c]; v1 += add2[(rowpix + S + w) * add2_ld + c]; }
        out[(rowpix + w) * out_ld + c] = v0;
        out[(rowpix + S + w) * out_ld + c] = v1;
    }
}

int log2_exact(int S) {
    int l = 0;
    while ((1 << l) < S) ++l;
    return ((1 << l) == S) ? l : -1;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return MTD_OK;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? MTD_OK : (int)e;
}

}  // namespace

extern "C" int mtd_rfft_rows_any(const float* x, int x_ld, float* R, int B, int S, void* stream) {
    const int logS = log2_exact(S);
    if (!x || !R || B <= 0 || logS < 6 || S > 512 || x_ld < 32) return MTD_EINVAL;
    const size_t lds = (size_t)S * 256 + (size_t)S * 4;
    int rc = set_lds(rfft_rows_any_kernel, lds);
    if (rc != MTD_OK) return rc;
    // (launch profiler, kernel class 2 = the HBM-bound spectral kernels: bytes = every operand element once, no flops)
    const int prof = mtd_prof_begin(2, 0, 1, (long long)B * S * S, 32, 32, 0, (hipStream_t)stream, 4.0 * B * S * 32.0 * (S + 2.0 * (S / 2 + 1)));
    MTD_LAUNCH(rfft_rows_any_kernel, dim3(B * S / 2), dim3(1024), lds, (hipStream_t)stream, x, x_ld, R, S, logS);
    mtd_prof_end(prof, (hipStream_t)stream);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_spec_mix_any(const float* R, const float* w2t, const float* b2, float* T, int B, int S, void* stream) {
    const int logS = log2_exact(S);
    if (!R || !w2t || !b2 || !T || B <= 0 || logS < 6 || S > 512) return MTD_EINVAL;
    const size_t lds = (size_t)S * 256 + (size_t)S * 4;
    int rc = set_lds(spec_mix_any_kernel, lds);
    if (rc != MTD_OK) return rc;
    const int prof = mtd_prof_begin(2, 1, 1, (long long)B * S * (S / 2 + 1), 64, 64, 0, (hipStream_t)stream, 2.0 * 4.0 * B * S * 64.0 * (S / 2 + 1));
    MTD_LAUNCH(spec_mix_any_kernel, dim3(B * (S / 2 + 1)), dim3(1024), lds, (hipStream_t)stream, R, w2t, b2, T, S, logS);
    mtd_prof_end(prof, (hipStream_t)stream);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_irfft_rows_any(const float* T, float* out, int out_ld, const float* add1, int add1_ld, const float* add2,
                                  int add2_ld, int B, int S, void* stream) {
    const int logS = log2_exact(S);
    if (!T || !out || B <= 0 || logS < 6 || S > 512 || out_ld < 32) return MTD_EINVAL;
    if ((add1 && add1_ld < 32) || (add2 && add2_ld < 32)) return MTD_EINVAL;
    const size_t lds = (size_t)S * 256 + (size_t)S * 4;
    int rc = set_lds(irfft_rows_any_kernel, lds);
    if (rc != MTD_OK) return rc;
    const int prof = mtd_prof_begin(2, 2, 1, (long long)B * S * S, 32, 32, 0, (hipStream_t)stream,
                                    4.0 * B * S * 32.0 * (2.0 * (S / 2 + 1) + S * (1.0 + (add1 ? 1 : 0) + (add2 ? 1 : 0))));
    MTD_LAUNCH(irfft_rows_any_kernel, dim3(B * S / 2), dim3(1024), lds, (hipStream_t)stream, T, out, out_ld, add1, add1_ld, add2,
               add2_ld, S, logS);
    mtd_prof_end(prof, (hipStream_t)stream);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
