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
