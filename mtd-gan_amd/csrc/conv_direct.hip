// Vector-ALU convolution kernels for the degenerate channel counts of MTD-GAN (1 -> C, C -> 1,
// 1 -> 1, Linear 512 -> 1).  These layers carry < 0.5 % of the step's FLOPs and are bound by HBM/L2
// traffic of the wide side, so they stay off the matrix cores.  Same argument contract and epilogue
// as mtd_conv_igemm.  The weight-gradient variant handles min(N, C) == 1.
#include "common.h"

namespace {

__device__ __forceinline__ void decompose(const mtd_geom& g, int m, int& b, int& oy, int& ox) {
    ox = m % g.OW;
    int t = m / g.OW;
    oy = t % g.OH;
    b = t / g.OH;
}

// one thread per (pixel, n)
__global__ __launch_bounds__(256) void direct_fwd_kernel(const mtd_conv_args a, long long total, int identity) {
    const mtd_geom& g = a.g;
    const float sc = a.scale ? *a.scale : 1.f;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int n = (int)(idx % a.N);
        const int m = (int)(idx / a.N);
        int b, oy, ox;
        decompose(g, m, b, oy, ox);
        float acc = 0.f;
        for (int ty = 0; ty < g.TH; ++ty) {
            const int iy = oy * g.in_sy + g.off_y + ty * g.tap_dy;
            if ((unsigned)iy >= (unsigned)g.IH) continue;
            for (int tx = 0; tx < g.TW; ++tx) {
                const int ix = ox * g.in_sx + g.off_x + tx * g.tap_dx;
                if ((unsigned)ix >= (unsigned)g.IW) continue;
                const int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
                const float* src = a.in + (((long long)b * g.IH + iy) * g.IW + ix) * a.in_ld;
                const float* wp = a.w + (long long)n * a.w_sn + (long long)kidx * a.w_st;
                if ((a.C & 3) == 0) {
                    for (int c = 0; c < a.C; c += 4) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
                        acc = fmaf(v[0], wp[(long long)c * a.w_sc], acc);
                        acc = fmaf(v[1], wp[(long long)(c + 1) * a.w_sc], acc);
                        acc = fmaf(v[2], wp[(long long)(c + 2) * a.w_sc], acc);
                        acc = fmaf(v[3], wp[(long long)(c + 3) * a.w_sc], acc);
                    }
                } else {
                    for (int c = 0; c < a.C; ++c) acc = fmaf(src[c], wp[(long long)c * a.w_sc], acc);
                }
            }
        }
        long long pix = m;
        if (!identity) pix = ((long long)b * g.OHF + (oy * g.out_sy + g.out_oy)) * g.OWF + (ox * g.out_sx + g.out_ox);
        float v = acc * sc + (a.bias ? a.bias[n] : 0.f);
        if (a.add1) v += a.add1[pix * a.add1_ld + n];
        if (a.add2) v += a.add2[pix * a.add2_ld + n];
        v = apply_act(v, a.act);
        if (a.mask) v *= (a.mask[pix * a.mask_ld + n] > 0.f) ? 1.f : a.mask_slope;
        a.out[pix * a.out_ld + n] = v;
    }
}

// ---- weight gradient with min(N, C) == 1 ------------------------------------------------------
// V = max(N, C) channels on the "wide" side.  Workgroup = 256 threads = VL channel lanes x PL pixel
// lanes; thread accumulates up to 16 taps for channels ch, ch+VL, ... (<= 8 per thread).
struct DWParams {
    mtd_wgrad_args a;
    int M, T, V, VL, PL, px_per_block;
    long long slab_stride;
};

template <int CH>   // channels per thread
__global__ __launch_bounds__(256) void direct_wgrad_kernel(const DWParams p) {
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int vl = tid % p.VL, pl = tid / p.VL;
    const bool n_is_one = (a.N == 1);
    float acc[CH][16];
    float bacc[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        bacc[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[k][t] = 0.f;
    }
    const int mb = blockIdx.x * p.px_per_block;
    const int me = min(p.M, mb + p.px_per_block);
    for (int m = mb + pl; m < me; m += p.PL) {
        int b, oy, ox;
        decompose(g, m, b, oy, ox);
        float pvv[CH];
        if (n_is_one) {
            float s = a.p[(long long)m * a.p_ld];
#pragma unroll
            for (int k = 0; k < CH; ++k) { pvv[k] = s; }
            if (vl == 0) bacc[0] += s;
        } else {
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                int ch = vl + k * p.VL;
                pvv[k] = ch < p.V ? a.p[(long long)m * a.p_ld + ch] : 0.f;
                bacc[k] += pvv[k];
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < p.T) {
                int ty = t / g.TW, tx = t % g.TW;
                int iy = oy * g.in_sy + g.off_y + ty * g.tap_dy;
                int ix = ox * g.in_sx + g.off_x + tx * g.tap_dx;
                if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) {
                    const float* src = a.q + (((long long)b * g.IH + iy) * g.IW + ix) * a.q_ld;
                    if (n_is_one) {
#pragma unroll
                        for (int k = 0; k < CH; ++k) {
                            int ch = vl + k * p.VL;
                            if (ch < p.V) acc[k][t] = fmaf(pvv[k], src[ch], acc[k][t]);
                        }
                    } else {
                        float s = src[0];
#pragma unroll
                        for (int k = 0; k < CH; ++k) acc[k][t] = fmaf(pvv[k], s, acc[k][t]);
                    }
                }
            }
        }
    }
    // reduce over the PL pixel lanes through LDS; pixel lane 0 writes the slab
    float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
    const long long nw = (long long)p.T * a.N * a.C;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int ch = vl + k * p.VL;
#pragma unroll
        for (int t = 0; t <= 16; ++t) {
            if (t < p.T || t == 16) {
                float v = (t == 16) ? bacc[k] : acc[k][t];
                __syncthreads();
                red[tid] = v;
                __syncthreads();
                if (pl == 0 && ch < p.V) {
                    float s = 0.f;
                    for (int j = 0; j < p.PL; ++j) s += red[j * p.VL + vl];
                    if (t < 16) {
                        // slab layout [tap][n][c]; one of n, c is 0
                        slab[(long long)t * p.V + ch] = s;
                    } else if (a.db) {
                        if (n_is_one) { if (ch == 0) slab[nw] = s; }
                        else slab[nw + ch] = s;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int mtd_conv_direct(const mtd_conv_args* a, void* stream) {
    if (!a || !a->in || !a->w || !a->out) return MTD_EINVAL;
    const mtd_geom& g = a->g;
    if (a->C <= 0 || a->N <= 0 || g.B <= 0 || g.TH <= 0 || g.TW <= 0) return MTD_EINVAL;
    if (a->in_ld < a->C || a->out_ld < a->N) return MTD_EINVAL;
    if ((a->C & 3) == 0 && ((a->in_ld & 3) || !aligned16(a->in))) return MTD_EALIGN;
    if ((g.OH - 1) * g.out_sy + g.out_oy >= g.OHF || (g.OW - 1) * g.out_sx + g.out_ox >= g.OWF) return MTD_EINVAL;
    long long total = geom_pixels(g) * a->N;
    int identity = (g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(direct_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a, total, identity);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// called from mtd_conv_wgrad when min(N,C)==1; returns the number of slabs written (or <0)
int mtd_direct_wgrad_launch(const mtd_wgrad_args* a, int* nslab_out, long long slab_stride, void* stream) {
    DWParams p;
    p.a = *a;
    p.M = (int)geom_pixels(a->g);
    p.T = a->g.TH * a->g.TW;
    p.V = a->N > a->C ? a->N : a->C;
    if (p.T > 16 || p.V > 2048) return MTD_EINVAL;
    int VL = 1;
    while (VL < p.V && VL < 256) VL <<= 1;
    p.VL = VL;
    p.PL = 256 / VL;
    int CH = (p.V + VL - 1) / VL;
    // pixels per block: aim for ~1024 blocks, at least PL*8 pixels each
    long long ppb = (p.M + 1023) / 1024;
    if (ppb < (long long)p.PL * 8) ppb = (long long)p.PL * 8;
    p.px_per_block = (int)ppb;
    int nblk = (int)((p.M + ppb - 1) / ppb);
    p.slab_stride = slab_stride;
    *nslab_out = nblk;
    hipStream_t s = (hipStream_t)stream;
    if (CH <= 1) hipLaunchKernelGGL((direct_wgrad_kernel<1>), dim3(nblk), dim3(256), 0, s, p);
    else if (CH <= 2) hipLaunchKernelGGL((direct_wgrad_kernel<2>), dim3(nblk), dim3(256), 0, s, p);
    else if (CH <= 8) hipLaunchKernelGGL((direct_wgrad_kernel<8>), dim3(nblk), dim3(256), 0, s, p);
    else return MTD_EINVAL;
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

int mtd_direct_wgrad_nslab(const mtd_wgrad_args* a) {
    long long M = geom_pixels(a->g);
    int V = a->N > a->C ? a->N : a->C;
    int VL = 1;
    while (VL < V && VL < 256) VL <<= 1;
    int PL = 256 / VL;
    long long ppb = (M + 1023) / 1024;
    if (ppb < (long long)PL * 8) ppb = (long long)PL * 8;
    return (int)((M + ppb - 1) / ppb);
}
