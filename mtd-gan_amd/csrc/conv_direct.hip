// Vector-ALU convolution kernels for the degenerate channel counts of MTD-GAN (1 -> C, C -> 1,
// 1 -> 1, Linear 512 -> 1).  These layers carry < 0.5 % of the step's FLOPs and are bound by HBM/L2
// traffic of the wide side, so they stay off the matrix cores.  Same argument contract and epilogue
// as mtd_conv_igemm.  The weight-gradient variant handles min(N, C) == 1.
//
// Roofline: HBM.  Algorithmic bytes per launch = 4 * (wide tensor elements + narrow tensor elements): the wide
// tensor (M x V floats) must cross HBM once.  The fast paths are organised around that:
//   fwd_c1   (1 -> N):  a thread owns 4 output channels with their taps in registers; one 16-byte store per 9 loads.
//   fwd_n1   (C -> 1):  C/4 lanes share a pixel (16-byte loads of the NHWC row), taps re-read through L1/L2 with
//                       an XCD-contiguous block order, DPP/shuffle reduction, one store per pixel.
//   wgrad_wide (N==1 or C==1): loops over the pixels of the WIDE tensor so each of its rows is loaded exactly once
//                       (16 bytes per lane); the narrow tensor is gathered at the tap offsets (L1-resident scalars).
#include "common.h"

namespace {

__device__ __forceinline__ void decompose(const mtd_geom& g, int m, int& b, int& oy, int& ox) {
    pix_decompose(m, g.OW, g.OH, b, oy, ox);
}

// one thread per (pixel, n)
__global__ __launch_bounds__(256) void direct_fwd_kernel(const mtd_conv_args a, long long total, int identity) {
    const mtd_geom& g = a.g;
    const ScalePair sp = load_scale(a);
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
        float v = acc * pick_scale(sp, m) + (a.bias ? a.bias[n] : 0.f);
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


// ---- fast forward paths ------------------------------------------------------------------------
struct FwdParams {
    mtd_conv_args a;
    int M, T, identity, vec_store;
    int G;              // lanes (fwd_n1) or threads (fwd_c1) that share one pixel
    int ppb;            // pixels per workgroup
    int tap_dy[16], tap_dx[16], tap_kidx[16];
};

__device__ __forceinline__ void store_epilogue4(const mtd_conv_args& a, const float acc[4], float sc, const float bias[4],
                                                long long pix, int n, int vec_store) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = acc[j] * sc + bias[j];
        if (a.add1) x += a.add1[pix * a.add1_ld + n + j];
        if (a.add2) x += a.add2[pix * a.add2_ld + n + j];
        x = apply_act(x, a.act);
        if (a.mask) x *= (a.mask[pix * a.mask_ld + n + j] > 0.f) ? 1.f : a.mask_slope;
        v[j] = x;
    }
    float* o = a.out + pix * a.out_ld + n;
    if (vec_store) {
        f32x4 q = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(o) = q;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j];
    }
}

// C == 1: thread = (pixel lane, 4 output channels)
__global__ __launch_bounds__(256) void fwd_c1_kernel(const FwdParams p) {
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int NQ = p.G, PL = 256 / NQ;
    const int nq = threadIdx.x % NQ, pl = threadIdx.x / NQ;
    const int n = nq * 4;
    float w[4][16];
    float bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias[j] = a.bias ? a.bias[n + j] : 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) w[j][t] = (t < p.T) ? a.w[(long long)(n + j) * a.w_sn + (long long)p.tap_kidx[t] * a.w_st] : 0.f;
    }
    const ScalePair sp = load_scale(a);
    const int mb = blockIdx.x * p.ppb;
    const int me = min(p.M, mb + p.ppb);
    for (int m = mb + pl; m < me; m += PL) {
        int b, oy, ox;
        decompose(g, m, b, oy, ox);
        const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < p.T) {
                const int iy = py + p.tap_dy[t], ix = px + p.tap_dx[t];
                if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) {
                    const float v = a.in[(((long long)b * g.IH + iy) * g.IW + ix) * a.in_ld];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(v, w[j][t], acc[j]);
                }
            }
        }
        long long pix = m;
        if (!p.identity) pix = ((long long)b * g.OHF + (oy * g.out_sy + g.out_oy)) * g.OWF + (ox * g.out_sx + g.out_ox);
        store_epilogue4(a, acc, pick_scale(sp, m), bias, pix, n, p.vec_store);
    }
}

// C == 1 on whole image rows, every tap within one pixel of the output position (3x3 "same" layers: the discriminator's
// conv11, the generator's encoder.0, the data gradient of decoder.0).  fwd_c1_kernel above pays, per pixel and thread,
// an index decomposition (two integer divisions), nine bounds tests and nine 64-bit global addresses for 36 FMAs and one
// 16-byte store: ~200 instructions per store, VALU-bound at 0.6-1.1 TB/s of output.  Here a workgroup owns R rows of
// one image: the R + 2 input rows (zero halo) are staged in LDS once, a pixel costs nine LDS reads at immediate
// offsets from three row pointers, the taps' weights sit in registers by NEIGHBOURHOOD position (zero where the
// geometry has no tap), and the store addresses advance by constants.  Taps are summed in neighbourhood order (the
// launch's tap order for a forward conv, its reverse for a data gradient).
// PLAIN: no add / mask operands and 16-byte stores (the forward layers): the epilogue is scale, bias, ACT and one store.
// With the general epilogue in the loop (uniform branches per value for operands that are not there) an iteration
// was ~190 instructions and a dozen taken branches for 36 FMAs -- the kernel was bound by that, not by its stores.
template <bool PLAIN, int ACT>
__global__ __launch_bounds__(256) void fwd_c1_tile_kernel(const FwdParams p, int R) {
    extern __shared__ float c1tile[];
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int W = g.OW, TW2 = W + 2;
    const int row0 = blockIdx.x * R;                       // first image row (over all images) of this workgroup
    const int b = row0 / g.OH, y0 = row0 - b * g.OH;
    for (int i = threadIdx.x; i < (R + 2) * TW2; i += 256) {
        const int ry = i / TW2, rx = i - ry * TW2;
        const int iy = y0 - 1 + ry, ix = rx - 1;
        float v = 0.f;
        if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) v = a.in[(((long long)b * g.IH + iy) * g.IW + ix) * a.in_ld];
        c1tile[i] = v;
    }
    const int NQ = p.G, PL = 256 / NQ;
    const int nq = threadIdx.x % NQ, pl = threadIdx.x / NQ;
    const int n = nq * 4;
    float w9[4][9], bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias[j] = a.bias ? a.bias[n + j] : 0.f;
#pragma unroll
        for (int q = 0; q < 9; ++q) w9[j][q] = 0.f;
    }
    {   // all 36 weight loads in flight at once (a loop over the launch's taps paid one memory round trip per tap, ~6 us
        // of a 13 us launch), then sorted into neighbourhood positions
        float wt[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wt[t][j] = (t < p.T) ? a.w[(long long)(n + j) * a.w_sn + (long long)p.tap_kidx[t] * a.w_st] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int q = (t < p.T) ? (g.off_y + p.tap_dy[t] + 1) * 3 + (g.off_x + p.tap_dx[t] + 1) : -1;     // neighbourhood position of tap t
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qq = 0; qq < 9; ++qq)
                    if (qq == q) w9[j][qq] = wt[t][j];
        }
    }
    const ScalePair sp = load_scale(a);
    __syncthreads();
    for (int ry = 0; ry < R; ++ry) {
        const float* r0 = c1tile + ry * TW2 + pl;          // rows ry-1, ry, ry+1 of the tile, column rx - 1
        const int m0 = (row0 + ry) * W;
        const float sc = pick_scale(sp, m0);                // a paired pass switches scale at an image boundary
        for (int rx = pl; rx < W; rx += PL, r0 += PL) {
            const float v00 = r0[0], v01 = r0[1], v02 = r0[2];
            const float v10 = r0[TW2], v11 = r0[TW2 + 1], v12 = r0[TW2 + 2];
            const float v20 = r0[2 * TW2], v21 = r0[2 * TW2 + 1], v22 = r0[2 * TW2 + 2];
            float acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
                s = fmaf(v00, w9[j][0], s); s = fmaf(v01, w9[j][1], s); s = fmaf(v02, w9[j][2], s);
                s = fmaf(v10, w9[j][3], s); s = fmaf(v11, w9[j][4], s); s = fmaf(v12, w9[j][5], s);
                s = fmaf(v20, w9[j][6], s); s = fmaf(v21, w9[j][7], s); s = fmaf(v22, w9[j][8], s);
                acc[j] = s;
            }
            if constexpr (PLAIN) {
                f32x4 q;
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = apply_act(acc[j] * sc + bias[j], ACT);
                *reinterpret_cast<f32x4*>(a.out + ((long long)m0 + rx) * a.out_ld + n) = q;
            } else {
                store_epilogue4(a, acc, sc, bias, (long long)m0 + rx, n, p.vec_store);
            }
        }
    }
}

// N == 1: G = C/4 lanes per pixel, 64/G pixels per wave iteration
__global__ __launch_bounds__(256) void fwd_n1_kernel(const FwdParams p, int nblk) {
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int CL = p.G, PPW = 64 / CL;
    const int cl = lane % CL, pg = lane / CL;
    f32x4 w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[t][j] = (t < p.T) ? a.w[(long long)(4 * cl + j) * a.w_sc + (long long)p.tap_kidx[t] * a.w_st] : 0.f;
    }
    const ScalePair sp = load_scale(a);
    const float bias = a.bias ? a.bias[0] : 0.f;
    const int blk = xcd_contiguous_block(blockIdx.x, nblk);
    const int mb = blk * p.ppb;
    const int me = min(p.M, mb + p.ppb);
    for (int base = mb + wave * PPW; base < me; base += 4 * PPW) {
        const int m = base + pg;
        const bool live = m < me;
        int b = 0, oy = 0, ox = 0;
        if (live) decompose(g, m, b, oy, ox);
        const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < p.T) {
                const int iy = py + p.tap_dy[t], ix = px + p.tap_dx[t];
                if (live & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(a.in + (((long long)b * g.IH + iy) * g.IW + ix) * a.in_ld + 4 * cl);
                    acc = fmaf(v[0], w[t][0], acc);
                    acc = fmaf(v[1], w[t][1], acc);
                    acc = fmaf(v[2], w[t][2], acc);
                    acc = fmaf(v[3], w[t][3], acc);
                }
            }
        }
        for (int off = CL >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (live && cl == 0) {
            long long pix = m;
            if (!p.identity) pix = ((long long)b * g.OHF + (oy * g.out_sy + g.out_oy)) * g.OWF + (ox * g.out_sx + g.out_ox);
            float v = acc * pick_scale(sp, m) + bias;
            if (a.add1) v += a.add1[pix * a.add1_ld];
            if (a.add2) v += a.add2[pix * a.add2_ld];
            v = apply_act(v, a.act);
            if (a.mask) v *= (a.mask[pix * a.mask_ld] > 0.f) ? 1.f : a.mask_slope;
            a.out[pix * a.out_ld] = v;
        }
    }
}

// N == 1 on whole image rows, 3x3 "same" geometry, C a multiple of 32 (the pixel-level heads s_/r_dconv61: 128 -> 1, the
// generator's decoder.0: 32 -> 1, the data gradient of conv11: 64 -> 1).  fwd_n1_kernel above re-reads every input
// pixel once per tap through L1 / L2 (nine 16-byte loads per lane and pixel) and reduces over the channel lanes with
// shuffles: 134 MB of input took 99 us (1.35 TB/s).  Here every input pixel is loaded ONCE: its nine tap products
//     plane_t[q] = sum_c in[q, c] * w[t, c]
// are one small GEMM -- 32 pixels x C channels times C x 9 (padded to 32) taps -- on the matrix cores (A fragments straight
// from global memory, 16-byte loads, as in the implicit GEMM; the 2 M C 32 flops are free beside the loads), the planes of
// the workgroup's R + 2 input rows go to LDS, and an output pixel is the sum of nine plane values at its tap offsets, in
// the launch's tap order.  Input traffic (R + 2) / R of the tensor, nothing else.
struct N1PlaneParams {
    mtd_conv_args a;
    int M, T, R;
    unsigned in_bytes;
    int tap_kidx[9], ddy[9], ddx[9];
};

template <int NCH>      // 32-channel chunks
__global__ __launch_bounds__(256) void fwd_n1_planes_kernel(const N1PlaneParams p) {
    extern __shared__ float planes[];                       // [9][(R + 2) * 66], zero columns 0 and 65
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    constexpr int W = 64, PW = W + 2;
    const int R = p.R, PR = R + 2, PSZ = PR * PW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int row0 = blockIdx.x * R;
    const int b = row0 / g.OH, y0 = row0 - b * g.OH;
    for (int i = tid; i < 9 * PSZ; i += 256) planes[i] = 0.f;
    // weights as MFMA B fragments: lane (tap = l31, kh) holds w[tap][32 ch + 16 kh + kk]
    float wf[NCH][16];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            wf[ch][kk] = (l31 < p.T) ? a.w[(long long)(32 * ch + 16 * kh + kk) * a.w_sc + (long long)p.tap_kidx[l31] * a.w_st] : 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    __syncthreads();
    // ---- phase 1: tap planes of input rows y0 - 1 .. y0 + R, one 32-pixel block (half a row) per wave and iteration
    const int nblocks = PR * 2;
    // Steps s = 0, 1, ... of this wave: (block wave + 4 * (s / NCH), chunk s % NCH).  TWO steps' loads are in flight while a
    // third is multiplied (registers an[0], an[1] alternate: the step loop is unrolled by two): with one step ahead a
    // chunk took a memory round trip (~1.5 us) for 0.43 us of MFMAs and the launch ran at 2.3-2.9 TB/s.
    f32x4 an[2][4];
    auto load = [&](int slot, int s) {
        const int blk = wave + 4 * (s / NCH), ch = s % NCH;
        const int pr = blk >> 1, x0 = (blk & 1) * 32;
        const int iy = y0 - 1 + pr;
        const bool ok = (blk < nblocks) & ((unsigned)iy < (unsigned)g.IH);
        const unsigned voff = ok ? (unsigned)((((((long long)b * g.IH + iy) * g.IW + x0 + l31) * a.in_ld) + 32 * ch + 16 * kh) * 4) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 4; ++j) an[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 16 * j, 0));
    };
    const int nsteps = ((nblocks - wave + 3) / 4) * NCH;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto step = [&](int slot, int s) {
        const int blk = wave + 4 * (s / NCH), ch = s % NCH;
        f32x4 ac[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ac[j] = an[slot][j];
        load(slot, s + 2);                                          // past the last step: out of range, zeros
#pragma unroll
        for (int c2 = 0; c2 < NCH; ++c2) {
            if (c2 == ch) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc = mfma32(ac[kk >> 2][kk & 3], wf[c2][kk], acc);
            }
        }
        if (ch == NCH - 1 && s < nsteps) {
            if (l31 < 9) {
                const int pr = blk >> 1, x0 = (blk & 1) * 32;
                float* dst = planes + l31 * PSZ + pr * PW + 1 + x0;
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[mfma32_row(e, lane)] = acc[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        }
    };
    load(0, 0);
    load(1, 1);
    for (int s = 0; s < nsteps; s += 2) {       // (an odd count runs one step on zeros: nothing is stored for it)
        step(0, s);
        step(1, s + 1);
    }
    __syncthreads();
    // ---- phase 2: nine plane values per output pixel, launch tap order, then the usual epilogue
    const ScalePair sp = load_scale(a);
    const float bias = a.bias ? a.bias[0] : 0.f;
    for (int i = tid; i < R * W; i += 256) {
        const int ry = i >> 6, rx = i & 63;
        const float* pc = planes + (ry + 1) * PW + rx + 1;
        float s = 0.f;
        for (int t = 0; t < p.T; ++t) s += pc[t * PSZ + p.ddy[t] * PW + p.ddx[t]];
        const int m = (row0 + ry) * W + rx;
        const long long pix = m;
        float v = s * pick_scale(sp, m) + bias;
        if (a.add1) v += a.add1[pix * a.add1_ld];
        if (a.add2) v += a.add2[pix * a.add2_ld];
        v = apply_act(v, a.act);
        if (a.mask) v *= (a.mask[pix * a.mask_ld] > 0.f) ? 1.f : a.mask_slope;
        a.out[pix * a.out_ld] = v;
    }
}

// ---- fast weight gradient: min(N, C) == 1, loop over the pixels of the wide tensor ---------------
struct WideParams {
    mtd_wgrad_args a;
    int Mw;             // pixels of the wide tensor's grid
    int T, V, CL, ppb, n_is_one;
    long long slab_stride;
    int tap_dy[16], tap_dx[16];
    int tile_rows;              // > 0: the workgroup's pixels are tile_rows whole image rows and every tap lies within one
    int ddy[16], ddx[16];       //      pixel of the wide tensor's pixel: narrow-tensor taps come from an LDS tile
};

template <int VEC>
__global__ __launch_bounds__(256) void wgrad_wide_kernel(const WideParams p) {
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    extern __shared__ float red[];                  // [4 waves][T * V + V], then the narrow-tensor tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int CL = p.CL, PPW = 64 / CL;
    const int cl = lane % CL, pg = lane / CL;
    const bool n1 = p.n_is_one != 0;
    const float* wide = n1 ? a.q : a.p;
    const int wide_ld = n1 ? a.q_ld : a.p_ld;
    const float* nar = n1 ? a.p : a.q;
    const int nar_ld = n1 ? a.p_ld : a.q_ld;
    const int GH = n1 ? g.IH : g.OH, GW = n1 ? g.IW : g.OW;      // grid of the wide tensor
    const int NH = n1 ? g.OH : g.IH, NW = n1 ? g.OW : g.IW;      // grid of the narrow tensor
    float acc[VEC][16];
    float bacc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        bacc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[j][t] = 0.f;
    }
    const int mb = blockIdx.x * p.ppb;
    const int me = min(p.Mw, mb + p.ppb);
    if (p.tile_rows > 0) {
        // Whole image rows per workgroup, 3x3-neighbourhood taps: the narrow tensor's rows y0-1 .. y0+R (zero halo) are staged
        // in LDS once, so a pixel costs one 16-byte load of the wide tensor and T broadcast LDS reads instead of T global
        // loads with their bounds tests and 64-bit address arithmetic (the loop below: ~200 instructions per pixel group,
        // VALU-bound at 56 us per launch).  Same pixel order per lane as that loop => the same sums.
        float* tile = red + 4 * (p.T * p.V + p.V);
        const int R = p.tile_rows, TW2 = GW + 2;
        const int row0 = mb / GW;                       // first image row (over all images) of this workgroup
        const int b = row0 / GH, y0 = row0 - b * GH;
        for (int i = threadIdx.x; i < (R + 2) * TW2; i += 256) {
            const int ry = i / TW2, rx = i - ry * TW2;
            const int ny = y0 - 1 + ry, nx = rx - 1;
            float v = 0.f;
            if (((unsigned)ny < (unsigned)NH) & ((unsigned)nx < (unsigned)NW))
                v = nar[(((long long)b * NH + ny) * NW + nx) * nar_ld];
            tile[i] = v;
        }
        __syncthreads();
        const float* wrow = wide + (long long)mb * wide_ld + VEC * cl;
        if (p.T == 9 && 4 * PPW <= GW) {
            // The nine tap offsets inside the tile are launch constants and a lane's pixels advance by a fixed stride: no
            // division, no per-tap branch or address arithmetic in the loop (the general form below is ~150 instructions
            // per 16-byte load of the wide tensor -- 28 us for 33 MB; this one ~55).  Same pixel and tap order per lane.
            int offs[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) offs[t] = p.ddy[t] * TW2 + p.ddx[t];
            const int step = 4 * PPW;
            int i = wave * PPW + pg;
            int ry = i / GW, rx = i - ry * GW;
            constexpr int U = 4;                    // wide-tensor loads in flight per lane (one at a time: 1.4 TB/s, latency-bound)
            const int n = me - mb;
            while (i < n) {
                f32x4 q4[U];
                float w1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int iu = i + u * step;
                    const long long off = (long long)(iu < n ? iu : i) * wide_ld;
                    if (VEC == 4) q4[u] = *reinterpret_cast<const f32x4*>(wrow + off);
                    else w1[u] = wrow[off];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (i < n) {
                        float wv[VEC];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) wv[j] = (VEC == 4) ? q4[u][j] : w1[u];
                        const float* tc = tile + (ry + 1) * TW2 + rx + 1;
                        if (!n1) {
#pragma unroll
                            for (int j = 0; j < VEC; ++j) bacc[j] += wv[j];
                        } else if (cl == 0) {
                            bacc[0] += tc[0];
                        }
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const float s = tc[offs[t]];
#pragma unroll
                            for (int j = 0; j < VEC; ++j) acc[j][t] = fmaf(s, wv[j], acc[j][t]);
                        }
                    }
                    i += step;
                    rx += step;
                    if (rx >= GW) { rx -= GW; ++ry; }
                }
            }
        } else
        for (int i = wave * PPW + pg; i < me - mb; i += 4 * PPW) {
            const int ry = i / GW, rx = i - ry * GW;
            float wv[VEC];
            if (VEC == 4) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(wrow + (long long)i * wide_ld);
#pragma unroll
                for (int j = 0; j < VEC; ++j) wv[j] = q4[j];
            } else {
                wv[0] = wrow[(long long)i * wide_ld];
            }
            const float* tc = tile + (ry + 1) * TW2 + rx + 1;
            if (!n1) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) bacc[j] += wv[j];
            } else if (cl == 0) {
                bacc[0] += tc[0];
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t < p.T) {
                    const float s = tc[p.ddy[t] * TW2 + p.ddx[t]];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j][t] = fmaf(s, wv[j], acc[j][t]);
                }
            }
        }
    } else
    for (int m = mb + wave * PPW + pg; m < me; m += 4 * PPW) {
        const int x = m % GW;
        const int t2 = m / GW;
        const int y = t2 % GH;
        const int b = t2 / GH;
        float wv[VEC];
        if (VEC == 4) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(wide + (long long)m * wide_ld + 4 * cl);
#pragma unroll
            for (int j = 0; j < VEC; ++j) wv[j] = q4[j];
        } else {
            wv[0] = wide[(long long)m * wide_ld + cl];
        }
        if (!n1) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) bacc[j] += wv[j];
        } else if (cl == 0) {
            bacc[0] += nar[(long long)m * nar_ld];             // same-size grids (checked on the host)
        }
        const int by = n1 ? (y - g.off_y) : (y * g.in_sy + g.off_y);
        const int bx = n1 ? (x - g.off_x) : (x * g.in_sx + g.off_x);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t < p.T) {
                const int ny = n1 ? by - p.tap_dy[t] : by + p.tap_dy[t];
                const int nx = n1 ? bx - p.tap_dx[t] : bx + p.tap_dx[t];
                if (((unsigned)ny < (unsigned)NH) & ((unsigned)nx < (unsigned)NW)) {
                    const float s = nar[(((long long)b * NH + ny) * NW + nx) * nar_ld];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j][t] = fmaf(s, wv[j], acc[j][t]);
                }
            }
        }
    }
    // pixel groups inside the wave, then the four waves through LDS (fixed order => deterministic)
    for (int off = CL; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            bacc[j] += __shfl_xor(bacc[j], off, 64);
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < p.T) acc[j][t] += __shfl_xor(acc[j][t], off, 64);
        }
    }
    const int per = p.T * p.V + p.V;
    if (pg == 0) {
        float* r = red + wave * per;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int ch = VEC * cl + j;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < p.T) r[t * p.V + ch] = acc[j][t];
            r[p.T * p.V + ch] = bacc[j];
        }
    }
    __syncthreads();
    float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
    const long long nw = (long long)p.T * p.V;
    for (int i = threadIdx.x; i < per; i += 256) {
        const float s = (red[i] + red[per + i]) + (red[2 * per + i] + red[3 * per + i]);
        if (i < nw) slab[i] = s;
        else if (a.db) {
            const int ch = i - (int)nw;
            if (n1) { if (ch == 0) slab[nw] = s; }
            else slab[nw + ch] = s;
        }
    }
}

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// fast weight-gradient path applicable?  fills the launch shape
bool wide_plan(const mtd_wgrad_args& a, WideParams& p) {
    const mtd_geom& g = a.g;
    p.a = a;
    p.T = g.TH * g.TW;
    p.n_is_one = (a.N == 1);
    p.V = a.N > a.C ? a.N : a.C;
    if (p.T > 16) return false;
    if (p.n_is_one) {
        if (g.in_sy != 1 || g.in_sx != 1 || g.OH != g.IH || g.OW != g.IW) return false;
        p.Mw = g.B * g.IH * g.IW;
    } else {
        p.Mw = g.B * g.OH * g.OW;
    }
    if (p.V == 1) p.CL = 1;
    else {
        if (p.V % 4) return false;
        p.CL = p.V / 4;
        if (!is_pow2(p.CL) || p.CL > 64) return false;
        const int ld = p.n_is_one ? a.q_ld : a.p_ld;
        const float* wide = p.n_is_one ? a.q : a.p;
        if ((ld % 4) || !aligned16(wide)) return false;
    }
    if ((long long)(p.T * p.V + p.V) * 4 * 4 > 48 * 1024) return false;
    // ~512 workgroups (tools/wide_probe.py, MTD_WIDE_WGS = 256 / 512 / 1024 / 2048: 25.4 / 22.9 / 23.1 / 31.5 us for conv11 at 32
    // images, main kernel + slab sum): with four wide loads in flight per lane a workgroup of four image rows amortises
    // its tile load and its cross-wave sum; the 2048 one-row workgroups of the first version were all prologue and epilogue
    static const int env_wgs = [] { const char* e = mtd_lab_env("MTD_WIDE_WGS"); return e ? atoi(e) : 512; }();
    long long ppb = (p.Mw + env_wgs - 1) / env_wgs;
    const long long min_ppb = 4ll * (64 / p.CL) * 4;
    if (ppb < min_ppb) ppb = min_ppb;
    p.ppb = (int)ppb;
    bool near = true;
    for (int t = 0; t < p.T; ++t) {
        p.tap_dy[t] = (t / g.TW) * g.tap_dy;
        p.tap_dx[t] = (t % g.TW) * g.tap_dx;
        // displacement of tap t's narrow-tensor pixel from the wide tensor's pixel (see the kernel's address formulas)
        p.ddy[t] = p.n_is_one ? -(g.off_y + p.tap_dy[t]) : (g.off_y + p.tap_dy[t]);
        p.ddx[t] = p.n_is_one ? -(g.off_x + p.tap_dx[t]) : (g.off_x + p.tap_dx[t]);
        near = near && p.ddy[t] >= -1 && p.ddy[t] <= 1 && p.ddx[t] >= -1 && p.ddx[t] <= 1;
    }
    // LDS-tile path: same-size stride-1 grids, the workgroup's pixel range = whole rows of one image
    const int GW = p.n_is_one ? g.IW : g.OW, GH = p.n_is_one ? g.IH : g.OH;
    p.tile_rows = 0;
    if (near && g.in_sy == 1 && g.in_sx == 1 && g.OH == g.IH && g.OW == g.IW && GW > 0 && p.ppb % GW == 0 && GH % (p.ppb / GW) == 0 &&
        (long long)(p.ppb / GW + 2) * (GW + 2) * 4 <= 16 * 1024)
        p.tile_rows = p.ppb / GW;
    return true;
}

}  // namespace

extern "C" int mtd_conv_direct(const mtd_conv_args* a, void* stream) {
    if (!a || !a->in || !a->w || !a->out || a->out2 || a->act == MTD_ACT_RELU_ADD) return MTD_EINVAL;
    const mtd_geom& g = a->g;
    if (a->C <= 0 || a->N <= 0 || g.B <= 0 || g.TH <= 0 || g.TW <= 0) return MTD_EINVAL;
    if (a->in_ld < a->C || a->out_ld < a->N) return MTD_EINVAL;
    if ((a->C & 3) == 0 && ((a->in_ld & 3) || !aligned16(a->in))) return MTD_EALIGN;
    if ((g.OH - 1) * g.out_sy + g.out_oy >= g.OHF || (g.OW - 1) * g.out_sx + g.out_ox >= g.OWF) return MTD_EINVAL;
    long long total = geom_pixels(g) * a->N;
    int identity = (g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW);
    const long long Mpix = geom_pixels(g);
    const int T = g.TH * g.TW;
    const bool fast_c1 = (a->C == 1 && (a->N % 4) == 0 && is_pow2(a->N / 4) && a->N / 4 <= 256 && T <= 16 && Mpix < (1ll << 31));
    const bool fast_n1 = (a->N == 1 && (a->C % 4) == 0 && is_pow2(a->C / 4) && a->C / 4 <= 64 && T <= 16 && Mpix < (1ll << 31) &&
                          (a->in_ld % 4) == 0 && aligned16(a->in));
    if (fast_c1 || fast_n1) {
        FwdParams p;
        p.a = *a;
        p.M = (int)Mpix;
        p.T = T;
        p.identity = identity;
        p.vec_store = ((a->out_ld % 4) == 0 && aligned16(a->out)) ? 1 : 0;
        for (int t = 0; t < T; ++t) {
            const int ty = t / g.TW, tx = t % g.TW;
            p.tap_dy[t] = ty * g.tap_dy;
            p.tap_dx[t] = tx * g.tap_dx;
            p.tap_kidx[t] = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
        }
        if (fast_c1) {
            p.G = a->N / 4;
            const int PL = 256 / p.G;
            long long ppb = (Mpix + 2047) / 2048;
            if (ppb < PL) ppb = PL;
            p.ppb = (int)ppb;
            const int nblk = (int)((Mpix + ppb - 1) / ppb);
            // whole-row tiles with the input rows in LDS where the geometry allows (see fwd_c1_tile_kernel)
            bool near = g.in_sy == 1 && g.in_sx == 1 && g.OH == g.IH && g.OW == g.IW && identity && T <= 9;
            bool seen[9] = {false, false, false, false, false, false, false, false, false};
            for (int t = 0; t < T && near; ++t) {
                const int dy = g.off_y + p.tap_dy[t], dx = g.off_x + p.tap_dx[t];
                near = dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1 && !seen[(dy + 1) * 3 + dx + 1];
                if (near) seen[(dy + 1) * 3 + dx + 1] = true;
            }
            static const int env_tile = [] { const char* e = mtd_lab_env("MTD_C1_TILE"); return e ? atoi(e) : 1; }();
            int R = 0;
            if (near && env_tile && g.OW % PL == 0 && (a->scale2 == nullptr || a->scale_split % (g.OH * g.OW) == 0)) {
                // ~512 workgroups of 4 or 8 image rows: with one row each (2048 workgroups of four pixels per thread) the
                // launch was ramp and tail (19 us for 16.8 MB of output)
                static const int env_wgs = [] { const char* e = mtd_lab_env("MTD_C1_WGS"); return e ? atoi(e) : 512; }();
                R = (int)(((Mpix + env_wgs - 1) / env_wgs) / g.OW);
                if (R < 1) R = 1;
                while (R > 1 && g.OH % R) --R;                                 // whole tiles per image
                if (R < 1 || (long long)(R + 2) * (g.OW + 2) * 4 > 48 * 1024) R = 0;
            }
            if (R > 0) {
                const size_t lds = (size_t)(R + 2) * (g.OW + 2) * sizeof(float);
                const dim3 grid((unsigned)((long long)g.B * g.OH / R));
                const bool plain = !a->add1 && !a->add2 && !a->mask && p.vec_store;
                if (plain && a->act == MTD_ACT_LRELU) hipLaunchKernelGGL((fwd_c1_tile_kernel<true, MTD_ACT_LRELU>), grid, dim3(256), lds, (hipStream_t)stream, p, R);
                else if (plain && a->act == MTD_ACT_RELU) hipLaunchKernelGGL((fwd_c1_tile_kernel<true, MTD_ACT_RELU>), grid, dim3(256), lds, (hipStream_t)stream, p, R);
                else if (plain && a->act == MTD_ACT_NONE) hipLaunchKernelGGL((fwd_c1_tile_kernel<true, MTD_ACT_NONE>), grid, dim3(256), lds, (hipStream_t)stream, p, R);
                else hipLaunchKernelGGL((fwd_c1_tile_kernel<false, 0>), grid, dim3(256), lds, (hipStream_t)stream, p, R);
            } else
            hipLaunchKernelGGL(fwd_c1_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
        } else if ([&] {
                       // tap planes on the matrix cores (fwd_n1_planes_kernel): 3x3 "same" geometry on 64-pixel rows, C = 32 / 64 / 128
                       static const int env_planes = [] { const char* e = mtd_lab_env("MTD_N1_PLANES"); return e ? atoi(e) : 1; }();
                       if (!env_planes || !(a->C == 32 || a->C == 64 || a->C == 128) || T > 9 || !identity) return false;
                       if (g.in_sy != 1 || g.in_sx != 1 || g.OH != g.IH || g.OW != g.IW || g.OW != 64 || (g.OH % 8)) return false;
                       if (a->w_sc <= 0 || (a->scale2 && a->scale_split % (g.OH * g.OW))) return false;
                       bool seen[9] = {false, false, false, false, false, false, false, false, false};
                       for (int t = 0; t < T; ++t) {
                           const int dy = g.off_y + p.tap_dy[t], dx = g.off_x + p.tap_dx[t];
                           if (dy < -1 || dy > 1 || dx < -1 || dx > 1 || seen[(dy + 1) * 3 + dx + 1]) return false;
                           seen[(dy + 1) * 3 + dx + 1] = true;
                       }
                       const long long bytes = (((long long)g.B * g.IH * g.IW - 1) * a->in_ld + a->C) * 4;
                       return bytes < (1ll << 31);
                   }()) {
            N1PlaneParams q;
            q.a = *a;
            q.M = (int)Mpix;
            q.T = T;
            static const int env_r = [] { const char* e = mtd_lab_env("MTD_N1_R"); return e ? atoi(e) : 8; }();
            q.R = env_r;
            q.in_bytes = (unsigned)((((long long)g.B * g.IH * g.IW - 1) * a->in_ld + a->C) * 4);
            for (int t = 0; t < T; ++t) {
                q.tap_kidx[t] = p.tap_kidx[t];
                q.ddy[t] = g.off_y + p.tap_dy[t];
                q.ddx[t] = g.off_x + p.tap_dx[t];
            }
            const size_t lds = (size_t)9 * (q.R + 2) * 66 * sizeof(float);
            const dim3 grid((unsigned)((long long)g.B * g.OH / q.R));
            if (a->C == 32) hipLaunchKernelGGL((fwd_n1_planes_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, q);
            else if (a->C == 64) hipLaunchKernelGGL((fwd_n1_planes_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((fwd_n1_planes_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, q);
        } else {
            p.G = a->C / 4;
            const int PPW = 64 / p.G;
            long long ppb = (Mpix + 2047) / 2048;
            if (ppb < 4 * PPW) ppb = 4 * PPW;
            p.ppb = (int)ppb;
            const int nblk = (int)((Mpix + ppb - 1) / ppb);
            hipLaunchKernelGGL(fwd_n1_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, nblk);
        }
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(direct_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a, total, identity);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// called from mtd_conv_wgrad when min(N,C)==1; returns the number of slabs written (or <0)
int mtd_direct_wgrad_launch(const mtd_wgrad_args* a, int* nslab_out, long long slab_stride, void* stream) {
    {
        WideParams wp;
        if (wide_plan(*a, wp)) {
            wp.slab_stride = slab_stride;
            const int nblk = (wp.Mw + wp.ppb - 1) / wp.ppb;
            *nslab_out = nblk;
            const int tgw = wp.n_is_one ? a->g.IW : a->g.OW;
            const size_t lds = ((size_t)4 * (wp.T * wp.V + wp.V) + (wp.tile_rows > 0 ? (size_t)(wp.tile_rows + 2) * (tgw + 2) : 0)) * sizeof(float);
            if (wp.V == 1) hipLaunchKernelGGL((wgrad_wide_kernel<1>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, wp);
            else hipLaunchKernelGGL((wgrad_wide_kernel<4>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, wp);
            MTD_LAUNCH_CHECK();
            return MTD_OK;
        }
    }
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
    {
        WideParams wp;
        if (wide_plan(*a, wp)) return (wp.Mw + wp.ppb - 1) / wp.ppb;
    }
    long long M = geom_pixels(a->g);
    int V = a->N > a->C ? a->N : a->C;
    int VL = 1;
    while (VL < V && VL < 256) VL <<= 1;
    int PL = 256 / VL;
    long long ppb = (M + 1023) / 1024;
    if (ppb < (long long)PL * 8) ppb = (long long)PL * 8;
    return (int)((M + ppb - 1) / ppb);
}
