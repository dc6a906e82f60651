// 64-point register FFT and the forward row transform of the Res-FFT-Conv spectral path, shared by resfft.hip and by
// conv_wgrad.hip (whose row-window weight-gradient launch can carry the row transform of the same cotangent along).
#pragma once
#include "common.h"

namespace {

__device__ __constant__ const float COS64[32] = {
    1.f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f,
    0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f,
    0.f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f,
    -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f};
__device__ __constant__ const float SIN64[32] = {
    0.f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f,
    0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f,
    1.f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f,
    0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f};

__host__ __device__ constexpr int brev6(int k) {
    return ((k & 1) << 5) | ((k & 2) << 3) | ((k & 4) << 1) | ((k & 8) >> 1) | ((k & 16) >> 3) | ((k & 32) >> 5);
}

// In-place 64-point complex DFT, X[k] = sum_n x[n] e^{SIGN * 2 pi i k n / 64}, unnormalised.
// Radix-2 decimation in frequency: the result for frequency k is left at index brev6(k).
template <int SIGN>
__device__ __forceinline__ void fft64(float (&re)[64], float (&im)[64]) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int half = 32 >> s;
        const int tstep = 1 << s;
#pragma unroll
        for (int blk = 0; blk < 64; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int i0 = blk + j, i1 = i0 + half;
                const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                re[i0] = ar + br;
                im[i0] = ai + bi;
                const float dr = ar - br, di = ai - bi;
                const int tw = j * tstep;
                if (tw == 0) {
                    re[i1] = dr;
                    im[i1] = di;
                } else if (tw == 16) {
                    if (SIGN < 0) { re[i1] = di; im[i1] = -dr; }
                    else { re[i1] = -di; im[i1] = dr; }
                } else {
                    const float c = COS64[tw];
                    const float sn = (SIGN < 0) ? -SIN64[tw] : SIN64[tw];
                    re[i1] = dr * c - di * sn;
                    im[i1] = dr * sn + di * c;
                }
            }
        }
    }
}

// ---- 64-point transform split over the four lanes of a quad (resfft4.hip, and the spectral epilogue of the halo-tile conv)
__host__ __device__ constexpr int brev4(int k) { return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3); }

// In-place 16-point complex DFT, X[k] = sum_n x[n] e^{SIGN 2 pi i k n / 16}, unnormalised; radix-2 decimation in
// frequency: the result for frequency k is left at index brev4(k).  Twiddles w16^t = w64^(4 t).
template <int SIGN>
__device__ __forceinline__ void fft16(float (&re)[16], float (&im)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int half = 8 >> s;
        const int tstep = 1 << s;
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int i0 = blk + j, i1 = i0 + half;
                const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
                re[i0] = ar + br;
                im[i0] = ai + bi;
                const float dr = ar - br, di = ai - bi;
                const int tw = j * tstep;                   // in units of 2 pi / 16
                if (tw == 0) {
                    re[i1] = dr;
                    im[i1] = di;
                } else if (tw == 4) {
                    if (SIGN < 0) { re[i1] = di; im[i1] = -dr; }
                    else { re[i1] = -di; im[i1] = dr; }
                } else {
                    const float c = COS64[4 * tw];
                    const float sn = (SIGN < 0) ? -SIN64[4 * tw] : SIN64[4 * tw];
                    re[i1] = dr * c - di * sn;
                    im[i1] = dr * sn + di * c;
                }
            }
        }
    }
}

__device__ __forceinline__ float quad_bcast(float v, int k) {
    // value of lane k of this lane's quad (DPP quad_perm broadcast)
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (k) {
        case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, true); break;
        case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xf, 0xf, true); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xf, 0xf, true); break;
        default: r = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xf, 0xf, true); break;
    }
    return __builtin_bit_cast(float, r);
}

// 64-point transform over a quad.  On entry lane j (= lane & 3) holds x[4 m + j] in (re, im)[m]; on exit it holds
// X[16 j + r] in (re, im)[r], X[k] = sum_n x[n] e^{SIGN 2 pi i k n / 64} (unnormalised).  tc / ts: cos, sin of
// 2 pi j r / 64 for this lane's j (quad_twiddles).
template <int SIGN>
__device__ __forceinline__ void fft64_quad(float (&re)[16], float (&im)[16], const float (&tc)[16], const float (&ts)[16], int j) {
    fft16<SIGN>(re, im);
    float gr[16], gi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {                          // G_j[r] = w64^(SIGN j r) F_j[r]
        const float fr = re[brev4(r)], fi = im[brev4(r)];
        if (SIGN < 0) { gr[r] = fr * tc[r] + fi * ts[r]; gi[r] = fi * tc[r] - fr * ts[r]; }
        else { gr[r] = fr * tc[r] - fi * ts[r]; gi[r] = fi * tc[r] + fr * ts[r]; }
    }
    const bool odd = (j & 1) != 0;
    const float sgn = (j & 2) ? -1.f : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {                          // X[16 q + r] = sum_j (SIGN i)^(j q) G_j[r], q = this lane
        const float g0r = quad_bcast(gr[r], 0), g1r = quad_bcast(gr[r], 1), g2r = quad_bcast(gr[r], 2), g3r = quad_bcast(gr[r], 3);
        const float g0i = quad_bcast(gi[r], 0), g1i = quad_bcast(gi[r], 1), g2i = quad_bcast(gi[r], 2), g3i = quad_bcast(gi[r], 3);
        const float ar = g0r + g2r, ai = g0i + g2i, br = g0r - g2r, bi = g0i - g2i;
        const float cr = g1r + g3r, ci = g1i + g3i, dr = g1r - g3r, di = g1i - g3i;
        // q even: A +- C;  q odd: B +- (SIGN i) D,  (SIGN i) D = SIGN * (-D_im, D_re)
        const float base_r = odd ? br : ar, base_i = odd ? bi : ai;
        const float add_r = odd ? (SIGN < 0 ? di : -di) : cr;
        const float add_i = odd ? (SIGN < 0 ? -dr : dr) : ci;
        re[r] = fmaf(sgn, add_r, base_r);
        im[r] = fmaf(sgn, add_i, base_i);
    }
}

__device__ __forceinline__ void quad_twiddles(int j, float (&tc)[16], float (&ts)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int idx = j * r;                              // 0 .. 45, angle 2 pi idx / 64
        const float s = (idx & 32) ? -1.f : 1.f;           // cos / sin (theta + pi) = -cos / -sin (theta)
        tc[r] = s * COS64[idx & 31];
        ts[r] = s * SIN64[idx & 31];
    }
}

// the same twiddles without a table read: j is 0..3, so every value is one of four immediates (three selects per value)
__device__ __forceinline__ void quad_twiddles_sel(int j, float (&tc)[16], float (&ts)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float c1 = COS64[r], s1 = SIN64[r];
        const float c2 = ((2 * r) & 32) ? -COS64[(2 * r) & 31] : COS64[(2 * r) & 31];
        const float s2 = ((2 * r) & 32) ? -SIN64[(2 * r) & 31] : SIN64[(2 * r) & 31];
        const float c3 = ((3 * r) & 32) ? -COS64[(3 * r) & 31] : COS64[(3 * r) & 31];
        const float s3 = ((3 * r) & 32) ? -SIN64[(3 * r) & 31] : SIN64[(3 * r) & 31];
        tc[r] = j == 0 ? 1.f : (j == 1 ? c1 : (j == 2 ? c2 : c3));
        ts[r] = j == 0 ? 0.f : (j == 1 ? s1 : (j == 2 ? s2 : s3));
    }
}

constexpr int NKW = 33;

// rows forward: two image rows (h, h+1) of one channel per lane; `pair` = index of this half-wave's row pair
__device__ __forceinline__ void rfft_rows_body(const float* __restrict__ x, int x_ld, float* __restrict__ R, int npairs, int col_weight,
                                               int pair, int c) {
    if (pair >= npairs) return;
    const int b = pair >> 5, h = (pair & 31) * 2;
    float re[64], im[64];
    const float* r0 = x + ((long long)(b * 64 + h) * 64) * x_ld + c;
    const float* r1 = r0 + (long long)64 * x_ld;
#pragma unroll
    for (int w = 0; w < 64; ++w) {
        re[w] = r0[(long long)w * x_ld];
        im[w] = r1[(long long)w * x_ld];
    }
    fft64<-1>(re, im);
    float* o0 = R + ((long long)(b * NKW) * 64 + h) * 64 + c;
#pragma unroll
    for (int kw = 0; kw <= 32; ++kw) {
        const int km = (64 - kw) & 63;
        const float zkr = re[brev6(kw)], zki = im[brev6(kw)];
        const float zmr = re[brev6(km)], zmi = im[brev6(km)];
        float sc = 0.125f * 0.5f;
        if (col_weight && kw != 0 && kw != 32) sc *= 2.f;
        const float ar = (zkr + zmr) * sc, ai = (zki - zmi) * sc;      // row h   : (Z[k] + conj Z[-k]) / 2
        const float br = (zki + zmi) * sc, bi = (zmr - zkr) * sc;      // row h+1 : (Z[k] - conj Z[-k]) / 2i
        float* o = o0 + (long long)kw * 64 * 64;
        o[0] = ar;
        o[32] = ai;
        o[64] = br;
        o[64 + 32] = bi;
    }
}

}  // namespace
