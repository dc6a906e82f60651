// Weight-gradient of a convolution on fp32 MFMA for gfx950.
//
//   dW(n, c, tap) (+)= sum_pix P[pix, n] * Q[gather(pix, tap), c]        db[n] (+)= sum_pix P[pix, n]
//
// GEMM view: output (N x C) per tap, reduction over M = B*OH*OW pixels.  With NHWC tensors both MFMA
// operands are read in their natural order (lane = channel, k = pixel), so LDS tiles are straight
// copies of 32-pixel row groups: P tile [32 px][32*WN], Q tile [32 px][32*WC] gathered per tap with
// zero fill.  A wave owns a (32*WN x 32*WC) output block for up to TG taps (accumulators stay in
// registers across its whole pixel range) and the four waves of a workgroup split the pixel range;
// they are summed through LDS at the end, so one slab per workgroup goes to the workspace.  Slabs are
// then summed in a fixed order by reduce kernels (deterministic) which also scatter into the
// OIHW / IOHW strided view of the parameter gradient.  db falls out of the A-operand registers.
//
// Roofline: fp32 MFMA.  Algorithmic flops = 2*M*N*C*taps.
#include "common.h"
#include "fft64.h"

namespace {

struct WgradParams {
    mtd_wgrad_args a;
    int M;          // pixels
    int ppw;        // pixels per wave (multiple of 32)
    int nCt;        // number of c tiles
    int T;          // taps
    int nslab;
    long long slab_stride;   // floats per slab = T*N*C + N
    unsigned p_bytes, q_bytes;          // extents for the buffer-load range checks
    int tap_dy[16], tap_dx[16], tap_delta[16];   // per-tap input displacement (pixels / bytes)
    // pair form (mtd_conv_wgrad_pair): the launch covers TWO problems of the shape a / M describe -- workgroups with
    // blockIdx.x >= pair_ns work on the second one, whose operands start pair_p_off / pair_q_off floats after a.p / a.q.
    // Slabs stay indexed by blockIdx.x: 0 .. pair_ns - 1 sum to the first gradient, the rest to the second.  0: one problem.
    int pair_ns = 0;
    long long pair_p_off = 0, pair_q_off = 0;
};

// mtd_wgrad_args.half_scale: the factor of the 32-pixel chunk that starts at pixel m (1 without it).  m_first is a multiple of 32 and
// every chunk starts at a multiple of 32, so a chunk lies in one half; the two scales are read once per kernel.
struct HalfScale { float s1, s2; int m_first; };
__device__ __forceinline__ HalfScale half_scale_load(const mtd_wgrad_args& a) {
    HalfScale h{1.f, 1.f, 0x7fffffff};
    if (a.half_scale) { h.s1 = *a.half_scale; h.s2 = *a.half_scale2; h.m_first = a.m_first; }
    return h;
}
__device__ __forceinline__ float half_scale_of(const HalfScale& h, int m) { return m < h.m_first ? h.s1 : h.s2; }

struct PairSel { int bx; const float* P; const float* Q; };
__device__ __forceinline__ PairSel pair_select(const WgradParams& p) {
    PairSel r{(int)blockIdx.x, p.a.p, p.a.q};
    if (p.pair_ns > 0 && r.bx >= p.pair_ns) {
        r.bx -= p.pair_ns;
        r.P += p.pair_p_off;
        r.Q += p.pair_q_off;
    }
    return r;
}

template <int WN, int WC, int TG>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
    constexpr int PLD = 32 * WN, QLD = 32 * WC;
    constexpr int UP = 4 * WN, UQ = 4 * WC;   // float4 units staged per lane
    constexpr int WAVE_FLOATS = 32 * (PLD + QLD);
    __shared__ __attribute__((aligned(16))) float Ls[4 * WAVE_FLOATS];

    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int ntile = blockIdx.y / p.nCt, ctile = blockIdx.y % p.nCt;
    const int n0 = ntile * 32 * WN, c0 = ctile * 32 * WC;
    const int tap0 = blockIdx.z * TG;
    const int ntap = min(TG, p.T - tap0);
    float* Ps = Ls + wave * WAVE_FLOATS;
    float* Qs = Ps + 32 * PLD;
    const PairSel ps = pair_select(p);
    const int mwave0 = (ps.bx * 4 + wave) * p.ppw;
    const HalfScale hs = half_scale_load(a);
    const bool do_bias = (a.db != nullptr) && ctile == 0 && blockIdx.z == 0;
    // Both operands are read with buffer loads: 32-bit byte offsets, an out-of-range offset returns 0 (zero
    // padding and the pixel tail without branches, so the compiler's vmcnt bookkeeping stays exact).
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.P), (short)0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.Q), (short)0, (int)p.q_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    f32x16 acc[TG][WN][WC];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int c = 0; c < WC; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][j][c][e] = 0.f;
    float bsum[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) bsum[j] = 0.f;

    f32x4 pv[UP], qv[UQ];
    unsigned qoff[UQ], qok[UQ];
    // per chunk: byte offsets of the rows this lane stages (P: identity pixel mapping; Q: tap (0,0) position)
    auto chunk_setup = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < UQ; ++i) {
            const int u = lane + 64 * i;
            const int row = u / (8 * WC), quad = u % (8 * WC);
            const int m = mbase + row;
            qoff[i] = 0;
            qok[i] = 0;
            if (m < p.M) {
                int b, oy, ox;
                pix_decompose(m, g.OW, g.OH, b, oy, ox);
                const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
                qoff[i] = (unsigned)(((((long long)b * g.IH + py) * g.IW + px) * a.q_ld + c0 + 4 * quad) * 4);
                for (int t = 0; t < ntap; ++t) {
                    const int iy = py + p.tap_dy[tap0 + t], ix = px + p.tap_dx[tap0 + t];
                    if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) qok[i] |= 1u << t;
                }
            }
        }
    };
    auto loadp = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < UP; ++i) {
            const int u = lane + 64 * i;
            const int row = u / (8 * WN), quad = u % (8 * WN);
            const int m = mbase + row;
            const unsigned off = (m < p.M) ? (unsigned)((((long long)m * a.p_ld) + n0 + 4 * quad) * 4) : OOB;
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, off, 0, 0));
        }
    };
    auto loadq = [&](int t) {
        const unsigned delta = (unsigned)p.tap_delta[tap0 + t];
#pragma unroll
        for (int i = 0; i < UQ; ++i) {
            const unsigned off = ((qok[i] >> t) & 1u) ? (qoff[i] + delta) : OOB;
            qv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, off, 0, 0));
        }
    };

    // No workgroup barrier in the main loop: each wave stages and reads ONLY its own LDS region, and the LDS
    // executes one wave's DS instructions in issue order (a later ds_read sees an earlier ds_write; a later
    // ds_write cannot overtake an earlier ds_read).  wave_barrier() only pins the compiler's ordering.
    chunk_setup(mwave0);
    loadp(mwave0);
    loadq(0);
    for (int mc = 0; mc < p.ppw; mc += 32) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < UP; ++i) *reinterpret_cast<f32x4*>(Ps + 4 * (lane + 64 * i)) = pv[i];   // row*(32*WN) + 4*quad == 4*u
        const bool more_chunks = (mc + 32 < p.ppw);
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            if (t < ntap) {
#pragma unroll
                for (int i = 0; i < UQ; ++i) *reinterpret_cast<f32x4*>(Qs + 4 * (lane + 64 * i)) = qv[i];
                __builtin_amdgcn_wave_barrier();
                // prefetch: next tap of this chunk, or P + first tap of the next chunk (always issued; masked OOB at the end)
                if (t + 1 < ntap) {
                    loadq(t + 1);
                } else {
                    const int mnext = more_chunks ? (mwave0 + mc + 32) : p.M;     // p.M => every offset out of range
                    chunk_setup(mnext);
                    loadp(mnext);
                    loadq(0);
                }
                // operands for the 16 k-steps of this tap, read in one batch, then 16 back-to-back MFMA groups
                float af[16][WN], bf[16][WC];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int k = 2 * kk + kh;
#pragma unroll
                    for (int j = 0; j < WN; ++j) af[kk][j] = Ps[k * PLD + j * 32 + l31];
#pragma unroll
                    for (int c = 0; c < WC; ++c) bf[kk][c] = Qs[k * QLD + c * 32 + l31];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0) {
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                        for (int j = 0; j < WN; ++j) bsum[j] += af[kk][j];
                }
                if (a.half_scale) {                 // (uniform; the bias sum above took the unscaled cotangent)
                    const float hsc = half_scale_of(hs, mwave0 + mc);
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                        for (int j = 0; j < WN; ++j) af[kk][j] *= hsc;
                }
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int c = 0; c < WC; ++c) acc[t][j][c] = mfma32(af[kk][j], bf[kk][c], acc[t][j][c]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();   // Q tile is rewritten for the next tap after these reads (same wave: in order)
            }
        }
    }

    // ---- sum the four waves through LDS (tile by tile, waves 0..3 in that order); every wave finishes a quarter of the
    //      tile (4 values per lane) and writes it into the workgroup's slab
    float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
    float* red = Ls;   // 4 * 1024 floats
    const bool direct = (p.nslab == 1);
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        if (t < ntap) {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int c = 0; c < WC; ++c) {
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[wave * 1024 + e * 64 + lane] = acc[t][j][c][e];
                    __syncthreads();
                    const int tap = tap0 + t;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = wave * 4 + i;
                        v[i] = red[e * 64 + lane] + red[1024 + e * 64 + lane] + red[2048 + e * 64 + lane] + red[3072 + e * 64 + lane];
                    }
                    const int kidx = (g.ky0 + (tap / g.TW) * g.ky_step) * g.KW + (g.kx0 + (tap % g.TW) * g.kx_step);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = wave * 4 + i;
                        const int n = n0 + j * 32 + mfma32_row(e, lane);
                        const int cc = c0 + c * 32 + l31;
                        if (direct) {       // a single pixel split: straight into the strided parameter-gradient view
                            float* dst = a.dw + (long long)n * a.w_sn + (long long)cc * a.w_sc + kidx;
                            *dst = (a.accumulate & 1) ? (*dst + v[i]) : v[i];
                        } else {
                            slab[((long long)tap * a.N + n) * a.C + cc] = v[i];
                        }
                    }
                }
        }
    }
    if (do_bias) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WN; ++j) red[(wave * WN + j) * 64 + lane] = bsum[j];
        __syncthreads();
        if (wave == 0 && lane < 32) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                float s = 0.f;
                for (int w = 0; w < 4; ++w) s += red[(w * WN + j) * 64 + lane] + red[(w * WN + j) * 64 + lane + 32];
                if (direct) {
                    float* dst = a.db + n0 + j * 32 + lane;
                    *dst = (a.accumulate & 2) ? (*dst + s) : s;
                } else {
                    slab[(long long)p.T * a.N * a.C + n0 + j * 32 + lane] = s;
                }
            }
        }
    }
}



// Epilogue of the register-operand kernels: sum the four waves of the workgroup through LDS (fixed order), then wave 0
// writes the tile -- into the workgroup's slab, or, when the launch has a single pixel split, straight into the strided
// parameter-gradient view (no reduce kernels at all).
template <int T, int NW>
__device__ __forceinline__ void reg_kernel_epilogue(const WgradParams& p, f32x16 (&acc)[T], float bsum, float* Ls, int n0, int c0,
                                                    bool do_bias, int bx) {
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31;
    const bool direct = (p.nslab == 1);
    constexpr int EPW = 16 / NW;        // values per lane that each wave finishes (sum over waves 0, 1, ... in that order)
    float* slab = a.ws + (long long)bx * p.slab_stride;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) Ls[wave * 1024 + e * 64 + lane] = acc[t][e];
        __syncthreads();
        const int ty = t / g.TW, tx = t % g.TW;
        const int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
        float v[EPW];
#pragma unroll
        for (int i = 0; i < EPW; ++i) {
            const int e = wave * EPW + i;
            v[i] = Ls[e * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) v[i] += Ls[w * 1024 + e * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < EPW; ++i) {
            const int e = wave * EPW + i;
            const int n = n0 + mfma32_row(e, lane);
            const int cc = c0 + l31;
            if (direct) {
                float* dst = a.dw + (long long)n * a.w_sn + (long long)cc * a.w_sc + kidx;
                *dst = (a.accumulate & 1) ? (*dst + v[i]) : v[i];
            } else {
                slab[((long long)t * a.N + n) * a.C + cc] = v[i];
            }
        }
    }
    if (do_bias) {
        float* red = Ls + NW * 1024;
        __syncthreads();
        red[wave * 64 + lane] = bsum;
        __syncthreads();
        if (wave == 0 && lane < 32) {
            float s2 = 0.f;
            for (int wv = 0; wv < NW; ++wv) s2 += red[wv * 64 + lane] + red[wv * 64 + lane + 32];
            if (direct) {
                float* dst = a.db + n0 + lane;
                *dst = (a.accumulate & 2) ? (*dst + s2) : s2;
            } else {
                slab[(long long)p.T * a.N * a.C + n0 + lane] = s2;
            }
        }
    }
}

// ---- row-window variant --------------------------------------------------------------------------
// Stride-1 convolutions whose output rows are a multiple of 16 pixels wide (all 64/32/16-pixel feature maps: two
// thirds of the step's weight-gradient flops).  No LDS in the main loop: with k-step kk of a 32-pixel chunk defined
// as pixels (kk, 16 + kk), lane (l31, kh) needs P[pixel kh*16 + kk][n0 + l31] and Q[that pixel + tap][c0 + l31] --
// one dword per k-step, 128 contiguous bytes per 32 lanes -- so both MFMA operands are loaded straight into
// registers (buffer loads, out-of-image reads return 0).  A lane's 16 pixels lie in one image row, so the TW taps of
// a filter row read the same 16 + TW - 1 input pixels shifted by one: each row of the window is loaded once and
// serves TW taps (54 instead of 144 Q loads per chunk for 3x3).  The next window row (or the next chunk's P and
// first row) is in flight under the current row's 16 * TW MFMAs.
template <int TH, int TW, int DX, int NW>
__device__ __forceinline__ void wgrad_row_body(const WgradParams& p, const int bx, const int by, float* Ls) {
    constexpr int T = TH * TW, WIN = 16 + TW - 1;
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int ntile = by / p.nCt, ctile = by % p.nCt;
    const int n0 = ntile * 32, c0 = ctile * 32;
    const int mwave0 = (bx * NW + wave) * p.ppw;
    const bool do_bias = (a.db != nullptr) && ctile == 0;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p), (short)0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q), (short)0, (int)p.q_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int smin = (DX > 0) ? 0 : -(TW - 1);          // smallest horizontal tap displacement
    const int pstep = a.p_ld * 4, qstep = a.q_ld * 4;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    float af[16], an[16], w0[WIN], w1[WIN];
    unsigned pbase = OOB, rowoff[TH];
    int xbase = 0;
    auto setup = [&](int mc) {
        const int m = mwave0 + mc + kh * 16;
        pbase = OOB;
        xbase = 0;
#pragma unroll
        for (int ty = 0; ty < TH; ++ty) rowoff[ty] = OOB;
        if (mc < p.ppw && m < p.M) {
            int b, oy, ox;
            pix_decompose(m, g.OW, g.OH, b, oy, ox);
            pbase = (unsigned)(((long long)m * a.p_ld + n0 + l31) * 4);
            xbase = ox + g.off_x + smin;
#pragma unroll
            for (int ty = 0; ty < TH; ++ty) {
                const int iy = oy + g.off_y + ty * g.tap_dy;
                if ((unsigned)iy < (unsigned)g.IH)
                    rowoff[ty] = (unsigned)(((((long long)b * g.IH + iy) * g.IW + xbase) * a.q_ld + c0 + l31) * 4);
            }
        }
    };
    auto load_p = [&](float* dst) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            dst[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, pbase, kk * pstep, 0));
    };
    auto load_row = [&](int ty, float* dst) {
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
            // the whole offset goes through the VGPR: the range check looks at it alone, and rowoff may be "negative"
            // (x = -1 of the first row) until j is added
            const unsigned off = (((unsigned)(xbase + j) < (unsigned)g.IW) & (rowoff[ty] != OOB)) ? rowoff[ty] + (unsigned)(j * qstep) : OOB;
            dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, off, 0, 0));
        }
    };
    auto mfma_row = [&](int ty, const float* wv) {
#pragma unroll
        for (int tx = 0; tx < TW; ++tx) {
            const int sh = (DX > 0) ? tx : (TW - 1 - tx);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc[ty * TW + tx] = mfma32(af[kk], wv[kk + sh], acc[ty * TW + tx]);
        }
    };

    setup(0);
    load_p(af);
    load_row(0, w0);
    for (int mc = 0; mc < p.ppw; mc += 32) {
#pragma unroll
        for (int ty = 0; ty < TH; ++ty) {
            float* cur = (ty & 1) ? w1 : w0;
            float* nxt = (ty & 1) ? w0 : w1;
            __builtin_amdgcn_sched_barrier(0);
            if (ty + 1 < TH) {
                load_row(ty + 1, nxt);
            } else {
                setup(mc + 32);                 // past the wave's range: every offset out of range, loads return 0
                load_p(an);
                load_row(0, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ty == 0) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) bsum += af[kk];
            }
            mfma_row(ty, cur);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) af[kk] = an[kk];
        if (TH & 1) {                           // an odd number of rows leaves the next chunk's first row in w1
#pragma unroll
            for (int j = 0; j < WIN; ++j) w0[j] = w1[j];
        }
    }

    reg_kernel_epilogue<T, NW>(p, acc, bsum, Ls, n0, c0, do_bias, bx);
}

template <int TH, int TW, int DX, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4) ? 2 : 1) void wgrad_row_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) float Ls[NW * 1024 + NW * 64];
    wgrad_row_body<TH, TW, DX, NW>(p, blockIdx.x, blockIdx.y, Ls);
}

// The row-window weight gradient of a generator block and the row transform of the same cotangent (the first kernel of the
// block's spectral backward chain) in ONE launch: workgroups 0 .. nwg-1 are the weight gradient's (MFMA-bound, one wave per
// SIMD), the rest take the 512 row-pair waves of mtd_rfft_rows four at a time (HBM-bound) -- 21 launches and their ramps
// less per generator backward pass, and the transform runs in the issue slots the MFMA waves leave.
struct RowsArgs { const float* x; int x_ld; float* R; int npairs; int col_weight; };
template <int DX>
__global__ __launch_bounds__(256, 2) void wgrad_row_rfft_kernel(const WgradParams p, const int nwg, const RowsArgs r) {
    __shared__ __attribute__((aligned(16))) float Ls[4 * 1024 + 4 * 64];
    if ((int)blockIdx.x < nwg) {
        wgrad_row_body<3, 3, DX, 4>(p, blockIdx.x, 0, Ls);
    } else {
        const int vb = ((int)blockIdx.x - nwg) * 4 + (threadIdx.x >> 6);      // block index of the stand-alone rows kernel
        const int lane = threadIdx.x & 63;
        rfft_rows_body(r.x, r.x_ld, r.R, r.npairs, r.col_weight, vb * 2 + (lane >> 5), lane & 31);
    }
}


// Epilogue of the kernels whose wave ty holds the four taps (ty, 0..3) of a 4x4 filter for one 32 x 32 (n, c) tile: slab, or
// -- single pixel split -- the gradient view itself, through LDS as [n][c][16] rows of 2 KB where the view is packed.
// Ls: >= 64 KB of LDS nobody reads any more (the caller has passed a barrier).
// NP: passes of the packed form (1: all 32 n rows at once, 64 KB; 2: 16 rows each, 32 KB).
template <int NP>
__device__ __forceinline__ void taps_epilogue(const WgradParams& p, f32x16 (&acc)[4], float bsum, float* Ls, int ty, int n0, int c0,
                                              bool do_bias) {
    constexpr int TW = 4, T = 16;
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
    const bool direct = (p.nslab == 1);
    const bool packed = direct && a.w_sc == T && (a.w_sn % 4) == 0 && ((((uintptr_t)a.dw) & 15) == 0) && g.ky0 == 0 && g.kx0 == 0 &&
                        g.ky_step == 1 && g.kx_step == 1 && g.KW == TW;
    if (packed) {
        // [n][c][ty][tx] in LDS (a lane's four taps of a filter row are 16 contiguous bytes), then straight copies of 2 KB rows.
        // Accumulator registers 16 h / NP .. hold the n rows 32 h / NP .. of the tile (mfma32_row), so a pass is a register range.
        constexpr int RP = 32 / NP, EP = 16 / NP;
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            if (h > 0) __syncthreads();
#pragma unroll
            for (int e = h * EP; e < (h + 1) * EP; ++e) {
                const f32x4 v = {acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
                *reinterpret_cast<f32x4*>(Ls + ((mfma32_row(e, lane) - h * RP) * 32 + l31) * T + ty * TW) = v;
            }
            __syncthreads();
#pragma unroll 4
            for (int i = tid; i < RP * 32 * T / 4; i += 256) {
                const int nl = h * RP + (i >> 7), rem = i & 127;
                float* dst = a.dw + (long long)(n0 + nl) * a.w_sn + (long long)c0 * T + 4 * rem;
                f32x4 v = *reinterpret_cast<const f32x4*>(Ls + 4 * i);
                if (a.accumulate & 1) v += *reinterpret_cast<const f32x4*>(dst);
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    } else {
        float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
#pragma unroll
        for (int tx = 0; tx < TW; ++tx) {
            const int t = ty * TW + tx;
            const int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + mfma32_row(e, lane), cc = c0 + l31;
                if (direct) {
                    float* dst = a.dw + (long long)n * a.w_sn + (long long)cc * a.w_sc + kidx;
                    *dst = (a.accumulate & 1) ? (*dst + acc[tx][e]) : acc[tx][e];
                } else {
                    slab[((long long)t * a.N + n) * a.C + cc] = acc[tx][e];
                }
            }
        }
    }
    if (do_bias) {
        __syncthreads();
        if (ty == 0) Ls[lane] = bsum;
        __syncthreads();
        if (tid < 32) {
            const float s2 = Ls[tid] + Ls[tid + 32];
            if (direct) {
                float* dst = a.db + n0 + tid;
                *dst = (a.accumulate & 2) ? (*dst + s2) : s2;
            } else {
                (a.ws + (long long)blockIdx.x * p.slab_stride)[(long long)p.T * a.N * a.C + n0 + tid] = s2;
            }
        }
    }
}

// ---- all-taps variant: 4x4 filters (the strided down convs of the discriminator trunk) --------------------------------
// wgrad_kernel gives every group of 1-3 taps a workgroup of its own (grid.z): each re-reads the whole cotangent and its
// own gather of the input -- 16 taps = 16 x (P + Q/4) through the L2s, 4.7 TB/s of fabric traffic at 72-76 TFLOP/s for
// the 64..256-channel layers -- and on the 4x4 / 2x2 / 1x1 maps of the deep layers, where one pixel split suffices, its
// workgroups scatter 4-byte values 64 bytes apart into the [n][c][4][4] gradient (16.8 MB written as 270 MB of sectors:
// 34 us for 0.27 GFLOP).  Here ONE workgroup owns a 32 x 32 (n, c) tile for ALL 16 taps: wave ty takes the four taps of
// filter row ty (4 x 16 accumulator registers), so the waves share the cotangent loads' cache lines, nobody splits pixels
// inside the workgroup and there is no cross-wave sum.  Operands go straight to registers as in the row-window kernel
// (lane = channel, k-step kk = pixels (kk, 16 + kk): one dword per lane and k-step, 128 contiguous bytes per 32 lanes;
// out-of-image taps are out-of-range buffer offsets = 0), the next 32-pixel chunk is in flight under the current chunk's
// 64 MFMAs.  With a single pixel split the tile is transposed through LDS to [n][c][16 taps] and leaves as 2 KB runs.
__global__ __launch_bounds__(256, 1) void wgrad_taps_kernel(const WgradParams p) {
    constexpr int TW = 4, T = 16;
    __shared__ __attribute__((aligned(16))) float Ls[32 * 32 * T];
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, ty = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int ntile = blockIdx.y / p.nCt, ctile = blockIdx.y % p.nCt;
    const int n0 = ntile * 32, c0 = ctile * 32;
    const PairSel ps = pair_select(p);
    const int m0 = ps.bx * p.ppw;                       // ppw: pixels per WORKGROUP here (a multiple of 32)
    const bool do_bias = (a.db != nullptr) && ctile == 0;
    const bool bias_wave = do_bias && ty == 0;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.P), (short)0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.Q), (short)0, (int)p.q_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const unsigned qtx = (unsigned)(g.tap_dx * a.q_ld * 4);
    const int dyrow = g.off_y + ty * g.tap_dy;

    f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    // This lane's 16 pixels of chunk mc are m0 + mc + 16 kh + kk, walked incrementally from one index decomposition.
    // The next chunk's loads (one P dword and four Q dwords per pixel, with their address arithmetic) are issued BETWEEN
    // the current chunk's MFMAs, one piece per MFMA: an in-order wave that issues them in a block first (first version)
    // spent ~3 k clocks there per chunk with the matrix pipe idle -- 57 TFLOP/s.
    struct Walk { int m, ox, oy, b; bool live; };
    auto walk_init = [&](int mc) {
        Walk w;
        w.m = m0 + mc + kh * 16;
        w.live = mc < p.ppw;
        pix_decompose(w.m, g.OW, g.OH, w.b, w.oy, w.ox);
        return w;
    };
    unsigned qbase = 0;
    bool rowok = false, pok = false;
    int ix0 = 0;
    float pa[16], qa[TW][16];
    // loads of pixel kk, tap tx of the chunk `w` walks through: the Q dword, and with the row's last tap the P dword
    // (ONE register set: an operand register is refilled right after the last MFMA that reads it)
    auto piece = [&](Walk& w, int kk, int tx) {
        if (tx == 0) {
            pok = w.live && w.m < p.M;
            const int iy = w.oy * g.in_sy + dyrow;
            ix0 = w.ox * g.in_sx + g.off_x;
            rowok = pok && ((unsigned)iy < (unsigned)g.IH);
            qbase = (unsigned)(((((long long)w.b * g.IH + iy) * g.IW + ix0) * a.q_ld + c0 + l31) * 4);
        }
        const bool colok = (unsigned)(ix0 + tx * g.tap_dx) < (unsigned)g.IW;
        const unsigned qoff = (rowok && colok) ? qbase + (unsigned)tx * qtx : OOB;
        qa[tx][kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, qoff, 0, 0));
        if (tx == TW - 1) {
            const unsigned poff = pok ? (unsigned)(((long long)w.m * a.p_ld + n0 + l31) * 4) : OOB;
            pa[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, poff, 0, 0));
            ++w.m;
            if (++w.ox == g.OW) {
                w.ox = 0;
                if (++w.oy == g.OH) { w.oy = 0; ++w.b; }
            }
        }
    };
    {
        Walk w = walk_init(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int tx = 0; tx < TW; ++tx) piece(w, kk, tx);
    }
    const HalfScale hs = half_scale_load(a);
    for (int mc = 0; mc < p.ppw; mc += 32) {
        const float hsc = half_scale_of(hs, m0 + mc);
        Walk w = walk_init(mc + 32);            // past the workgroup's range: every offset out of range, zeros
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float pk = pa[kk] * hsc;          // (mtd_wgrad_args.half_scale: 1 without it)
#pragma unroll
            for (int tx = 0; tx < TW; ++tx) {
                acc[tx] = mfma32(pk, qa[tx][kk], acc[tx]);
                // (the bias sum takes each P value where it is in a register anyway; summed at the top of the chunk it made
                // wave 0 wait for ALL of the chunk's loads before its first MFMA)
                if (tx == TW - 1) bsum += bias_wave ? pa[kk] : 0.f;
                piece(w, kk, tx);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    taps_epilogue<1>(p, acc, bsum, Ls, ty, n0, c0, do_bias);
}

// ---- halo-window variant: 4x4 / stride 2 / pad 1 on maps whose output side is a multiple of 8 (down1..3) -------------
// The LDS-staged kernel above runs these layers with one workgroup per tap (grid.z = 16): every tap re-reads the cotangent
// and its own gather of the input through the L2s -- 268 MB per launch for 42 MB of operands, 57 us at 75 TFLOP/s.  The
// all-taps kernel reads each operand dword straight into registers, 80 loads per 64 MFMAs and wave: 57-67 TFLOP/s.
// Here a workgroup owns a 32 x 32 (n, c) tile for all 16 taps (wave ty = filter row ty, as in the all-taps kernel) and
// walks 8 x 8 blocks of output pixels: the block's cotangent tile [64 px][32 n] and its 18 x 18 input window [324 px][32 c]
// arrive by LDS-DMA (49 one-KB instructions over the four waves, out-of-image pixels as out-of-range offsets = zeros),
// double-buffered, so the next block is in flight under this block's MFMAs and no operand passes through a register on
// its way in.  Every tap of every pixel is then an immediate offset into the window: per block and wave 32 + 128 LDS
// dword reads (lane = channel: 128 contiguous bytes per half-wave, conflict-free) for 128 MFMAs.  The input is read
// (18/16)^2 = 1.27 times, the cotangent once.
constexpr int S2_WPX = 18 * 18;                 // window pixels
constexpr int S2_QI = (S2_WPX + 7) / 8;         // 41 DMA instructions (8 pixels x 128 bytes each) for the window
constexpr int S2_PI = 8;                        // ... and 8 for the 64 cotangent pixels
constexpr int S2_NI = 52;                       // padded to 13 per wave (the rest land in a dummy KB): one vmcnt value for all

// DB: double-buffered, one workgroup per CU (100 KB of LDS); !DB: one buffer set, two workgroups per CU take turns
template <bool DB>
__global__ __launch_bounds__(256, DB ? 1 : 2) void wgrad_s2_kernel(const WgradParams p) {
    constexpr int TW = 4;
    typedef __attribute__((address_space(3))) float lds_f;
    // two buffer sets as SEPARATE LDS objects, the block loop unrolled by two: the compiler's wait insertion knows which
    // object a pending LDS-DMA writes, and with one array indexed by `buf` it put a vmcnt(0) -- a wait for the NEXT block's
    // loads too -- in front of every block's reads
    __shared__ __attribute__((aligned(1024))) float Qw0[S2_QI * 256];
    __shared__ __attribute__((aligned(1024))) float Qw1[S2_QI * 256];
    __shared__ __attribute__((aligned(1024))) float Pt0[S2_PI * 256];
    __shared__ __attribute__((aligned(1024))) float Pt1[S2_PI * 256];
    __shared__ __attribute__((aligned(1024))) float dummy[256];
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int ntile = blockIdx.y / p.nCt, ctile = blockIdx.y % p.nCt;
    const int n0 = ntile * 32, c0 = ctile * 32;
    const bool do_bias = (a.db != nullptr) && ctile == 0;
    const bool bias_wave = do_bias && ty == 0;
    const int nbx = g.OW >> 3, nby = g.OH >> 3;
    const int NB = g.B * nby * nbx;
    const PairSel ps = pair_select(p);
    const int blk0 = ps.bx * p.ppw;                     // ppw: 8 x 8 pixel blocks per WORKGROUP here
    const int blk1 = min(NB, blk0 + p.ppw);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.P), (short)0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.Q), (short)0, (int)p.q_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int rsub = lane >> 3, piece = lane & 7;

    auto stage = [&](int blk, float* Qd, float* Pd) {
        const int bx = blk % nbx;
        const int t2 = blk / nbx;
        const int by = t2 % nby, b = t2 / nby;
        const bool live = blk < blk1;
#pragma unroll
        for (int k = 0; k < S2_NI / 4; ++k) {
            const int i = ty + 4 * k;                   // instruction index, wave-uniform
            if (i < S2_QI) {
                const int wp = 8 * i + rsub;
                const int wy = wp / 18, wx = wp - wy * 18;
                const int iy = 16 * by - 1 + wy, ix = 16 * bx - 1 + wx;
                const bool ok = live & (wp < S2_WPX) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
                const unsigned voff = ok ? (unsigned)(((((long long)b * g.IH + iy) * g.IW + ix) * a.q_ld + c0 + piece * 4) * 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (lds_f*)(Qd + i * 256), 16, voff, 0, 0, 0);
            } else if (i < S2_QI + S2_PI) {
                const int r = i - S2_QI;                // output row of the block; rsub = column
                const long long m = ((long long)b * g.OH + 8 * by + r) * g.OW + 8 * bx + rsub;
                const unsigned voff = live ? (unsigned)((m * a.p_ld + n0 + piece * 4) * 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_f*)(Pd + r * 256), 16, voff, 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_f*)&dummy[0], 16, OOB, 0, 0, 0);
            }
        }
    };

    f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    // one block from LDS: k-step kk = output pixels (2 kk, 2 kk + 1) of the block = row kk >> 2, columns 2 (kk & 3) + kh.
    // The operands of the NEXT two k-steps (2 + 8 dword reads) are issued before the eight MFMAs of the current two: read on
    // demand (the compiler's order) every pair of MFMAs waited a full LDS round trip -- 55 % of the MFMA rate.
    auto compute = [&](const float* Pbuf, const float* Qbuf) {
        const float* Pb = Pbuf + l31 + 32 * kh;
        const float* Qb = Qbuf + (ty * 18 + 2 * kh) * 32 + l31;
        float pa[2][2], qa[2][2][TW];
        auto load_group = [&](int gi, int slot) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kk = 2 * gi + u;
                pa[slot][u] = Pb[64 * kk];
#pragma unroll
                for (int tx = 0; tx < TW; ++tx) qa[slot][u][tx] = Qb[((2 * (kk >> 2)) * 18 + 4 * (kk & 3) + tx) * 32];
            }
        };
        load_group(0, 0);
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) {
            if (gi + 1 < 16) load_group(gi + 1, (gi + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (bias_wave) bsum += pa[gi & 1][u];
#pragma unroll
                for (int tx = 0; tx < TW; ++tx) acc[tx] = mfma32(pa[gi & 1][u], qa[gi & 1][u][tx], acc[tx]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Bare s_barrier, not __syncthreads(): the latter's workgroup-scope fence makes the compiler wait for EVERY outstanding
    // LDS-DMA (vmcnt(0): they are LDS writes) -- the next block's too, which is the overlap this loop exists for.  What the
    // barriers need is stated by hand: before the first, this wave's DMA of the CURRENT block has landed (13 newer ones may
    // be in flight); before the second, its LDS reads have returned (the MFMAs consumed them).
    if constexpr (!DB) {
        for (int blk = blk0; blk < blk1; ++blk) {
            stage(blk, Qw0, Pt0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(Pt0, Qw0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
    stage(blk0, Qw0, Pt0);
    for (int blk = blk0; blk < blk1; blk += 2) {
        stage(blk + 1, Qw1, Pt1);                       // (past the range: zeros)
        asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        compute(Pt0, Qw0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // every read of this buffer is done before it is staged again
        if (blk + 1 < blk1) {
            stage(blk + 2, Qw0, Pt0);
            asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(Pt1, Qw1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the trailing (all out-of-range) stage
    __syncthreads();
    static_assert(S2_QI * 256 >= 16 * 32 * 16, "epilogue staging: 16 n rows per pass fit one window buffer");
    taps_epilogue<2>(p, acc, bsum, Qw0, ty, n0, c0, do_bias);
}

// ---- block-window variant -------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions on the 8x8, 4x4 and 2x2 feature maps of the discriminator trunk (the layers
// with 512 x 512 ... 256 x 1024 channels).  A lane's 16 pixels (k-steps) are two rows of an 8x8 image, a whole 4x4
// image or four 2x2 images, so every tap of every pixel is a compile-time shift inside a small register window of the
// input: 32 (8x8: two rows + one halo row either side) or 16 loads per 32-pixel chunk instead of 144, no LDS, no
// per-tap address arithmetic -- and k-steps whose tap falls into the zero padding for every lane are not issued at all
// (8 % / 31 % / 56 % of the MFMAs for 8x8 / 4x4 / 2x2).
template <int W, int NW>
__global__ __launch_bounds__(NW * 64, (W == 8 || NW == 8) ? 1 : 2) void wgrad_blk_kernel(const WgradParams p) {
    constexpr int T = 9;
    constexpr int HALO = (W == 8) ? 8 : 0;
    constexpr int NQ = 16 + 2 * HALO;
    __shared__ __attribute__((aligned(16))) float Ls[NW * 1024 + NW * 64];
    const mtd_wgrad_args& a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int ntile = blockIdx.y / p.nCt, ctile = blockIdx.y % p.nCt;
    const int n0 = ntile * 32, c0 = ctile * 32;
    const PairSel ps = pair_select(p);
    const int mwave0 = (ps.bx * NW + wave) * p.ppw;
    const HalfScale hs = half_scale_load(a);
    const bool do_bias = (a.db != nullptr) && ctile == 0;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.P), (short)0, (int)p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ps.Q), (short)0, (int)p.q_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int pstep = a.p_ld * 4, qstep = a.q_ld * 4;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum = 0.f;

    float af[16], an[16], qf[NQ];
    float qn[(W == 8) ? 1 : NQ];          // 4x4 / 2x2: plain double buffer; 8x8 recycles window rows in place
    unsigned pbase = OOB, qbase = OOB;
    bool up_ok = false, dn_ok = false;
    auto setup = [&](int mc) {
        const int m = mwave0 + mc + kh * 16;
        const bool valid = (mc < p.ppw) && (m < p.M);
        pbase = valid ? (unsigned)(((long long)m * a.p_ld + n0 + l31) * 4) : OOB;
        qbase = valid ? (unsigned)(((long long)m * a.q_ld + c0 + l31) * 4) : OOB;
        if (W == 8) {
            const int y0 = (m & 63) >> 3;
            up_ok = valid && y0 > 0;
            dn_ok = valid && y0 < 6;
        }
    };
    auto load_p = [&](float* dst) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            dst[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, pbase, kk * pstep, 0));
    };
    // window entries [j0, j1) of the chunk described by the current setup()
    auto load_q = [&](float* dst, int j0, int j1) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            if (j < j0 || j >= j1) continue;
            const int jj = j - HALO;
            const bool ok = (qbase != OOB) && (jj < 0 ? up_ok : (jj >= 16 ? dn_ok : true));
            const unsigned off = ok ? qbase + (unsigned)(jj * qstep) : OOB;
            dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, off, 0, 0));
        }
    };
    auto mfma_taprow = [&](int ty) {
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int x = kk % W, xx = x + tx - 1;
                if (xx < 0 || xx >= W) continue;
                if (W <= 4) {
                    const int y = (kk / W) % W, yy = y + ty - 1;
                    if (yy < 0 || yy >= W) continue;
                }
                acc[ty * 3 + tx] = mfma32(af[kk], qf[kk + (ty - 1) * W + (tx - 1) + HALO], acc[ty * 3 + tx]);
            }
        }
    };

    setup(0);
    load_p(af);
    load_q(qf, 0, NQ);
    for (int mc = 0; mc < p.ppw; mc += 32) {
        __builtin_amdgcn_sched_barrier(0);
        setup(mc + 32);
        load_p(an);
        if (W != 8) load_q(qn, 0, NQ);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) bsum += af[kk];
        if (a.half_scale) {                      // (uniform; the bias sum took the unscaled cotangent)
            const float hsc = half_scale_of(hs, mwave0 + mc);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) af[kk] *= hsc;
        }
        mfma_taprow(0);
        __builtin_amdgcn_sched_barrier(0);
        if (W == 8) load_q(qf, 0, 8);            // window row 0 is dead after filter row 0
        __builtin_amdgcn_sched_barrier(0);
        mfma_taprow(1);
        __builtin_amdgcn_sched_barrier(0);
        if (W == 8) load_q(qf, 8, 16);
        __builtin_amdgcn_sched_barrier(0);
        mfma_taprow(2);
        __builtin_amdgcn_sched_barrier(0);
        if (W == 8) load_q(qf, 16, 32);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) af[kk] = an[kk];
        if (W != 8) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) qf[j] = qn[j];
        }
    }
    reg_kernel_epilogue<T, NW>(p, acc, bsum, Ls, n0, c0, do_bias, blockIdx.x);
}

// out[g][idx] = sum_{s in group g} in[s][idx]     (group size gs), fixed order
// scalar forms of the two kernels below, for the thin layers (N == 1 or C == 1: slab strides are not multiples of 4)
// sum of slabs k0 .. k1-1 at element idx, in slab order, eight loads in flight
__device__ __forceinline__ float slab_sum1(const float* __restrict__ in, long long stride_in, int k0, int k1, long long idx) {
    float s = 0.f;
    const float* base = in + idx;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = base[(long long)(k + j) * stride_in];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < k1; ++k) s += base[(long long)k * stride_in];
    return s;
}

__global__ __launch_bounds__(256) void slab_group_sum_scalar_kernel(const float* __restrict__ in, float* __restrict__ out, int nslab, int gs,
                                                                    long long count, long long stride_in, long long stride_out) {
    const int grp = blockIdx.y;
    const int s0 = grp * gs, s1 = min(nslab, s0 + gs);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (long long)gridDim.x * 256)
        out[(long long)grp * stride_out + idx] = slab_sum1(in, stride_in, s0, s1, idx);
}

__global__ __launch_bounds__(256) void wgrad_finish_scalar_kernel(const WgradParams p, const float* __restrict__ in, int nslab, long long stride_in) {
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const long long nw = (long long)p.T * a.N * a.C;
    const long long count = nw + (a.db ? a.N : 0);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (long long)gridDim.x * 256) {
        const float s = slab_sum1(in, stride_in, 0, nslab, idx);
        if (idx < nw) {
            int c = (int)(idx % a.C);
            long long t2 = idx / a.C;
            int n = (int)(t2 % a.N);
            int tap = (int)(t2 / a.N);
            int ty = tap / g.TW, tx = tap % g.TW;
            int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
            float* dst = a.dw + (long long)n * a.w_sn + (long long)c * a.w_sc + kidx;
            *dst = (a.accumulate & 1) ? (*dst + s) : s;
        } else {
            float* dst = a.db + (idx - nw);
            *dst = (a.accumulate & 2) ? (*dst + s) : s;
        }
    }
}

// The two stages above in ONE launch for the thin layers (round 6: 25 launch pairs per step on the weight-gradient stream): thread
// (element e = tid & 15, group g = tid >> 4) sums group g of element blockIdx.x * 16 + e, the sixteen group sums meet in LDS and are added
// in group order -- the association of slab_group_sum_scalar_kernel followed by wgrad_finish_scalar_kernel, bit for bit.  nslab <= 16 gs.
__global__ __launch_bounds__(256) void wgrad_finish_scalar2_kernel(const WgradParams p, const float* __restrict__ in, int nslab, long long stride_in, int gs) {
    __shared__ float part[16][17];
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const long long nw = (long long)p.T * a.N * a.C;
    const long long count = nw + (a.db ? a.N : 0);
    const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long long idx = (long long)blockIdx.x * 16 + e;
    const int ng = (nslab + gs - 1) / gs;
    float s = 0.f;
    if (idx < count && grp < ng) s = slab_sum1(in, stride_in, grp * gs, min(nslab, (grp + 1) * gs), idx);
    part[grp][e] = s;
    __syncthreads();
    if (grp != 0 || idx >= count) return;
    float t = 0.f;
    for (int k = 0; k < ng; ++k) t += part[k][e];
    if (idx < nw) {
        int c = (int)(idx % a.C);
        long long t2 = idx / a.C;
        int n = (int)(t2 % a.N);
        int tap = (int)(t2 / a.N);
        int ty = tap / g.TW, tx = tap % g.TW;
        int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
        float* dst = a.dw + (long long)n * a.w_sn + (long long)c * a.w_sc + kidx;
        *dst = (a.accumulate & 1) ? (*dst + t) : t;
    } else {
        float* dst = a.db + (idx - nw);
        *dst = (a.accumulate & 2) ? (*dst + t) : t;
    }
}

// sum of slabs k0 .. k1-1 at float4 index i4, in slab order; the loads of eight slabs are in flight together (a plain
// `for k: s += in[k]` loop of unknown length serialises one memory round trip per slab)
__device__ __forceinline__ f32x4 slab_sum4(const float* __restrict__ in, long long stride_in, int k0, int k1, long long i4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = in + 4 * i4;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (long long)(k + j) * stride_in);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    if (k + 4 <= k1) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (long long)(k + j) * stride_in);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[j];
        k += 4;
    }
    for (; k < k1; ++k) s += *reinterpret_cast<const f32x4*>(base + (long long)k * stride_in);
    return s;
}

// out[g][idx] = sum_{s in group g} in[s][idx]     (group size gs), fixed order.  count and both strides are multiples of 4
// (N % 32 == 0) and the workspace is 16-byte aligned (checked by the caller).
__global__ __launch_bounds__(256) void slab_group_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int nslab, int gs,
                                                             long long count, long long stride_in, long long stride_out) {
    const int grp = blockIdx.y;
    const int s0 = grp * gs, s1 = min(nslab, s0 + gs);
    const long long count4 = count >> 2;
    for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < count4; i4 += (long long)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(out + (long long)grp * stride_out + 4 * i4) = slab_sum4(in, stride_in, s0, s1, i4);
}

// final: sum <= gs slabs and scatter into the strided weight-gradient view (+ bias gradient).  A thread owns four
// consecutive c of one (tap, n); index arithmetic in 32 bits (T * N * C < 2^31, checked by the caller).
// (pair launches, gridDim.y == 2: row y sums slabs y nslab .. (y + 1) nslab - 1 into its own gradient -- a.dw, dw2 -- and row 0
// sums the bias rows of ALL slabs)
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const WgradParams p, const float* __restrict__ in, int nslab, long long stride_in,
                                                           float* __restrict__ dw2) {
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const unsigned nw4 = (unsigned)(p.T * a.N * a.C) >> 2;
    const unsigned count4 = nw4 + ((a.db && blockIdx.y == 0) ? ((unsigned)a.N >> 2) : 0u);
    const unsigned c4n = (unsigned)a.C >> 2;
    float* const dw = blockIdx.y ? dw2 : a.dw;
    const float* const in_w = in + (long long)blockIdx.y * nslab * stride_in;
    for (unsigned i4 = blockIdx.x * 256 + threadIdx.x; i4 < count4; i4 += gridDim.x * 256) {
        const f32x4 s = i4 < nw4 ? slab_sum4(in_w, stride_in, 0, nslab, i4) : slab_sum4(in, stride_in, 0, nslab * (int)gridDim.y, i4);
        if (i4 < nw4) {
            const unsigned t2 = i4 / c4n;
            const unsigned c = (i4 - t2 * c4n) << 2;
            const unsigned tap = t2 / (unsigned)a.N;
            const unsigned n = t2 - tap * (unsigned)a.N;
            const int ty = (int)tap / g.TW, tx = (int)tap % g.TW;
            const int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
            float* dst = dw + (long long)n * a.w_sn + (long long)c * a.w_sc + kidx;
            if (a.accumulate & 1) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = dst[(long long)j * a.w_sc];
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[(long long)j * a.w_sc] = o[j] + s[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[(long long)j * a.w_sc] = s[j];
            }
        } else {
            float* dst = a.db + ((i4 - nw4) << 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = (a.accumulate & 2) ? (dst[j] + s[j]) : s[j];
        }
    }
}

// the two reduce stages and the scatter in one launch (workgroup of 16 float4 columns x SQ * SQ slab runs, common.h
// block_slab_sum): the 256-slab sums of the 32-channel 3x3 layers cost one 5 us launch instead of two
// scatter one summed float4 (four consecutive c of one (tap, n), or four bias entries) into the strided gradient views
__device__ __forceinline__ void finish_scatter4(const mtd_wgrad_args& a, float* dw, int T, unsigned i4, const f32x4 s) {
    const mtd_geom& g = a.g;
    const unsigned nw4 = (unsigned)(T * a.N * a.C) >> 2;
    const unsigned c4n = (unsigned)a.C >> 2;
    if (i4 < nw4) {
        const unsigned t2 = i4 / c4n;
        const unsigned c = (i4 - t2 * c4n) << 2;
        const unsigned tap = t2 / (unsigned)a.N;
        const unsigned n = t2 - tap * (unsigned)a.N;
        const int ty = (int)tap / g.TW, tx = (int)tap % g.TW;
        const int kidx = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
        float* dst = dw + (long long)n * a.w_sn + (long long)c * a.w_sc + kidx;
        if (a.accumulate & 1) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = dst[(long long)j * a.w_sc];
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(long long)j * a.w_sc] = o[j] + s[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[(long long)j * a.w_sc] = s[j];
        }
    } else {
        float* dst = a.db + ((i4 - nw4) << 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = (a.accumulate & 2) ? (dst[j] + s[j]) : s[j];
    }
}

// (pair launches, gridDim.y == 2: as wgrad_finish_kernel; a workgroup's 16 columns are all weights or all bias: N, C multiples of 32)
template <int SQ>
__global__ __launch_bounds__(16 * SQ * SQ) void wgrad_reduce_finish_kernel(const WgradParams p, const float* __restrict__ in, int nslab,
                                                                           long long stride_in, float* __restrict__ dw2) {
    __shared__ f32x4 red[16 * SQ * (SQ + 1)];
    const mtd_wgrad_args& a = p.a;
    const unsigned nw4 = (unsigned)(p.T * a.N * a.C) >> 2;
    const unsigned count4 = nw4 + (a.db ? ((unsigned)a.N >> 2) : 0u);
    const bool bias_cols = blockIdx.x * 16 >= nw4;                 // workgroup-uniform
    if (bias_cols && blockIdx.y != 0) return;
    const unsigned i4 = blockIdx.x * 16 + (threadIdx.x & 15);
    const f32x4 s = bias_cols ? block_slab_sum<SQ>(in, stride_in, nslab * (int)gridDim.y, i4, i4 < count4, red)
                              : block_slab_sum<SQ>(in + (long long)blockIdx.y * nslab * stride_in, stride_in, nslab, i4, i4 < count4, red);
    if ((threadIdx.x >> 4) != 0 || i4 >= count4) return;
    finish_scatter4(a, blockIdx.y ? dw2 : a.dw, p.T, i4, s);
}

// the same for many layers in one launch: workgroup -> (layer, 16-column chunk) through the table's block prefix sums
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(const mtd_wgrad_reduce_desc* __restrict__ table, int count) {
    __shared__ f32x4 red[16 * 8 * 9];
    int li = 0, hi = count - 1;                   // last layer whose first_block <= blockIdx.x (uniform scalar loads)
    while (li < hi) {
        const int mid = (li + hi + 1) >> 1;
        if ((int)blockIdx.x >= table[mid].first_block) li = mid;
        else hi = mid - 1;
    }
    const mtd_wgrad_reduce_desc& d = table[li];
    const mtd_wgrad_args& a = d.a;
    const unsigned count4 = ((unsigned)(d.T * a.N * a.C) >> 2) + (a.db ? ((unsigned)a.N >> 2) : 0u);
    const unsigned i4 = (blockIdx.x - d.first_block) * 16 + (threadIdx.x & 15);
    const f32x4 s = block_slab_sum<8>(a.ws, d.slab_stride, d.nslab, i4, i4 < count4, red);
    if ((threadIdx.x >> 4) != 0 || i4 >= count4) return;
    finish_scatter4(a, a.dw, d.T, i4, s);
}

#include "conv_wgrad_wino.h"
#include "conv_wgrad_wino_s2.h"
#include "conv_wgrad_wino32.h"

struct WPlan { int cfg, WN, WC, TG, ppw, nsplit, ntg, nw; };

int g_wforce_cfg = -1, g_wforce_split = -1;     // tuning hook (mtd_conv_wgrad_override)
int g_wforce_nw = 0;                            // waves per workgroup of the register-operand kernels (env MTD_WGRAD_NW, lab only)
constexpr int NWCFG = 7;
const int kWcfgWN[NWCFG] = {1, 1, 2, 1, 1, 1, 2};
const int kWcfgWC[NWCFG] = {1, 1, 2, 1, 1, 1, 2};
const int kWcfgTG[NWCFG] = {9, 4, 1, 8, 3, 1, 3};
// (cfg 13: wgrad_taps_kernel.  A 64 x 64 tile with a whole filter row per workgroup, wgrad_kernel<2, 2, 4>, needs 256
// accumulator registers and spilled 600 bytes per lane: not instantiated.)

// row-window kernel: stride 1, rows of a multiple of 16 output pixels, 3x3 (unit tap spacing) or 1x1
bool row_window_ok(const mtd_wgrad_args& a) {
    const mtd_geom& g = a.g;
    if (g.in_sy != 1 || g.in_sx != 1 || (g.OW % 16) != 0) return false;
    if (g.TH == 1 && g.TW == 1) return true;
    return g.TH == 3 && g.TW == 3 && (g.tap_dx == 1 || g.tap_dx == -1);
}

// block-window kernel: 3x3 / stride 1 / pad 1 on square 8x8, 4x4 or 2x2 maps (forward tap order)
int block_window_w(const mtd_wgrad_args& a) {
    const mtd_geom& g = a.g;
    if (g.in_sy != 1 || g.in_sx != 1 || g.TH != 3 || g.TW != 3 || g.tap_dy != 1 || g.tap_dx != 1) return 0;
    if (g.off_y != -1 || g.off_x != -1 || g.IH != g.OH || g.IW != g.OW || g.OH != g.OW) return 0;
    return (g.OW == 8 || g.OW == 4 || g.OW == 2) ? g.OW : 0;
}

// g_wplan_div: 2 while planning ONE of the two problems of a pair launch (half the workgroup targets); a parameter, not a
// global, so that concurrent callers (main thread + autograd's backward thread, several devices) cannot see each other's value
WPlan make_wplan(const mtd_wgrad_args& a, const int g_wplan_div = 1) {
    WPlan pl{};
    const int T = a.g.TH * a.g.TW;
    const long long M = geom_pixels(a.g);
    // LDS-staged kernels (strided convs and feature maps narrower than 16 pixels); choices from tools/census.py --sweep-wgrad
    if (T == 1 && a.N % 64 == 0 && a.C % 64 == 0) pl.cfg = 2;
    else if (T <= 4) pl.cfg = 1;
    else if (T <= 9) pl.cfg = (M <= 2048) ? 4 : 0;              // few pixels, many tiles: 3 taps per wave, no pixel split
    else {
        // 4x4 taps (round 6, tools/wgrad_s2_small_probe.py: both halves of a paired pass are one launch now, so down4 has 1024 pixels
        // and down5 256): 64 x 64 tiles with one tap each from 1024 pixels on (108 us against 140 for 32 x 32 tiles x 3 taps), 32 x 32
        // tiles with one tap each up to 256 pixels (42 us against 51)
        static const int env_t16 = [] { const char* e = mtd_lab_env("MTD_WGRAD_T16_PLAN"); return e ? atoi(e) : 1; }();      // (0: round 5's thresholds)
        if (!env_t16) pl.cfg = (M >= 2048 && a.N % 64 == 0 && a.C % 64 == 0) ? 2 : 4;
        else if (M >= 1024 && a.N % 64 == 0 && a.C % 64 == 0) pl.cfg = 2;
        else pl.cfg = (M <= 256) ? 5 : 4;
    }
    // Winograd F(2x2, 3x3) on ONE 32 x 32 block (conv_wgrad_wino32.h): the generator's 32 -> 32 layers on maps of at least
    // MTD_WGRAD_WINO32_MIN_HW pixels a side (where the row-window kernel is the plan otherwise).  cfg 19; ppw = chunks of 16 tiles per slice.
    static const int env_w32 = [] { const char* e = mtd_lab_env("MTD_WGRAD_WINO32"); return e ? atoi(e) : 1; }();
    static const int env_w32_hw = [] { const char* e = mtd_lab_env("MTD_WGRAD_WINO32_MIN_HW"); return e ? atoi(e) : 32; }();
    if (((env_w32 && g_wforce_cfg == -1) || g_wforce_cfg == 19) && wgrad_wino32_ok(a) && a.g.OH >= env_w32_hw && a.g.OW >= env_w32_hw) {
        const long long tiles = (long long)a.g.B * (a.g.OH / 2) * (a.g.OW / 2);
        const long long chunks = (tiles + W32_T - 1) / W32_T;
        long long ns = 256 / g_wplan_div;
        if (g_wforce_split > 0) ns = g_wforce_split;
        if (ns > chunks / 4) ns = chunks / 4;
        if (ns > chunks) ns = chunks;
        if (ns >= 2) {
            const long long cps = (chunks + ns - 1) / ns;
            ns = (chunks + cps - 1) / cps;
            if (ns >= 2) {
                pl.cfg = 19;
                pl.WN = 1; pl.WC = 1; pl.TG = T; pl.ntg = 1; pl.nw = 8;
                pl.ppw = (int)cps;
                pl.nsplit = (int)ns;
                return pl;
            }
        }
    }
    // Winograd F(3x3, 2x2) form of the 4x4 / stride-2 layers (conv_wgrad_wino_s2.h): output maps of at least MTD_WGRAD_WINO_S2_MIN_HW
    // pixels a side.  cfg 18; planned like cfg 16 (ppw = chunks of eight tiles per pixel split, always through slabs).
    static const int env_ws2 = [] { const char* e = mtd_lab_env("MTD_WGRAD_WINO_S2"); return e ? atoi(e) : 1; }();
    static const int env_ws2_hw = [] { const char* e = mtd_lab_env("MTD_WGRAD_WINO_S2_MIN_HW"); return e ? atoi(e) : 8; }();
    if (((env_ws2 && g_wforce_cfg == -1) || g_wforce_cfg == 18) && wgrad_wino_s2_ok(a) && a.g.OH >= env_ws2_hw && a.g.OW >= env_ws2_hw) {
        pl.cfg = 18;
        pl.WN = 2; pl.WC = 2; pl.TG = T; pl.ntg = 1; pl.nw = 8;
        const long long blocks = wgrad_wino_s2_blocks(a);
        const long long chunks = (wgrad_wino_s2_tiles(a, a.g.B) + WGW_T - 1) / WGW_T;
        long long ns = (256 + blocks - 1) / blocks;
        if (g_wforce_split > 0) ns = g_wforce_split;
        if (ns > chunks / 4) ns = chunks / 4;
        if (ns < 2) ns = 2;
        if (ns > chunks) ns = chunks;
        const long long cps = (chunks + ns - 1) / ns;
        ns = (chunks + cps - 1) / cps;
        if (ns >= 2) {
            pl.ppw = (int)cps;
            pl.nsplit = (int)ns;
            return pl;
        }
        pl = WPlan{};
        pl.cfg = (M >= 1024 && a.N % 64 == 0 && a.C % 64 == 0) ? 2 : ((M <= 256) ? 5 : 4);
    }
    static const int env_s2 = [] { const char* e = mtd_lab_env("MTD_WGRAD_S2"); return e ? atoi(e) : 1; }();
    if (a.g.TH == 4 && a.g.TW == 4 && a.g.in_sy == 2 && a.g.in_sx == 2 && a.g.off_y == -1 && a.g.off_x == -1 && a.g.tap_dy == 1 &&
        a.g.tap_dx == 1 && (a.g.OH % 8) == 0 && (a.g.OW % 8) == 0 && ((env_s2 && g_wforce_cfg == -1) || g_wforce_cfg == 15)) {
        // halo-window kernel: 8 x 8 pixel blocks; ~512 workgroups of the single-buffer form, two per CU (51 KB of LDS each), which take
        // turns on the matrix cores: 46.6-49.6 us per layer against 51-54 for 256 double-buffered workgroups and 57 for wgrad_kernel<2,2,1>
        pl.cfg = 15;
        pl.WN = 1; pl.WC = 1; pl.TG = T; pl.ntg = 1; pl.nw = 4;
        const long long tiles = (long long)(a.N / 32) * (a.C / 32);
        const long long NB = (long long)a.g.B * (a.g.OH / 8) * (a.g.OW / 8);
        static const int env_s2wgs = [] { const char* e = mtd_lab_env("MTD_WGRAD_S2_WGS"); return e ? atoi(e) : 512; }();
        long long ns = (env_s2wgs / g_wplan_div + tiles - 1) / tiles;
        if (g_wforce_split > 0) ns = g_wforce_split;
        if (ns > NB) ns = NB;
        if (ns < 1) ns = 1;
        const long long bpw = (NB + ns - 1) / ns;
        ns = (NB + bpw - 1) / bpw;
        pl.ppw = (int)bpw;
        pl.nsplit = (int)ns;
        return pl;
    }
    static const int env_taps = [] { const char* e = mtd_lab_env("MTD_WGRAD_TAPS"); return e ? atoi(e) : 1; }();
    static const int env_taps_maxm = [] { const char* e = mtd_lab_env("MTD_WGRAD_TAPS_MAXM"); return e ? atoi(e) : 128; }();
    if (a.g.TH == 4 && a.g.TW == 4 && ((env_taps && g_wforce_cfg == -1 && M <= env_taps_maxm) || g_wforce_cfg == 13)) {
        // all-taps kernel: ~one workgroup per CU; pixels per workgroup a multiple of 32
        pl.cfg = 13;
        pl.WN = 1; pl.WC = 1; pl.TG = T; pl.ntg = 1; pl.nw = 4;
        const long long tiles = (long long)(a.N / 32) * (a.C / 32);
        static const int env_wgs = [] { const char* e = mtd_lab_env("MTD_WGRAD_TAPS_WGS"); return e ? atoi(e) : 256; }();
        long long ns = (env_wgs / g_wplan_div + tiles - 1) / tiles;
        if (g_wforce_split > 0) ns = g_wforce_split;
        const long long max_splits = (M + 31) / 32;
        if (ns > max_splits) ns = max_splits;
        if (ns < 1) ns = 1;
        long long ppw = (M + ns - 1) / ns;
        ppw = ((ppw + 31) / 32) * 32;
        ns = (M + ppw - 1) / ppw;
        pl.ppw = (int)ppw;
        pl.nsplit = (int)ns;
        return pl;
    }
    // Winograd F(2x2, 3x3) form (conv_wgrad_wino.h): 3x3 stride-1 layers with N, C multiples of 64 on maps of at least
    // MTD_WGRAD_WINO_MIN_HW pixels a side.  cfg 16; ppw = chunks of eight tiles per pixel split.
    static const int env_wino = [] { const char* e = mtd_lab_env("MTD_WGRAD_WINO"); return e ? atoi(e) : 1; }();
    static const int env_wino_hw = [] { const char* e = mtd_lab_env("MTD_WGRAD_WINO_MIN_HW"); return e ? atoi(e) : 8; }();
    if (((env_wino && g_wforce_cfg == -1) || g_wforce_cfg == 16) && wgrad_wino_ok(a) && a.g.OH >= env_wino_hw && a.g.OW >= env_wino_hw) {
        pl.cfg = 16;
        pl.WN = 2; pl.WC = 2; pl.TG = T; pl.ntg = 1; pl.nw = 8;
        const long long blocks = wgrad_wino_blocks(a);            // 64 x 64 (F(2x2)) or 64 x 32 (F(2x4)) blocks of (n, c)
        const long long chunks = (wgrad_wino_tiles(a, a.g.B) + WGW_T - 1) / WGW_T;
        long long ns = (256 + blocks - 1) / blocks;              // about one workgroup per CU
        if (g_wforce_split > 0) ns = g_wforce_split;
        if (ns > chunks / 4) ns = chunks / 4;                    // at least four chunks per slice
        if (ns < 2) ns = 2;                                      // (always through slabs: the kernel has no direct-store form)
        if (ns > chunks) ns = chunks;
        const long long cps = (chunks + ns - 1) / ns;
        ns = (chunks + cps - 1) / cps;
        if (ns >= 2) {
            pl.ppw = (int)cps;
            pl.nsplit = (int)ns;
            return pl;
        }
        pl = WPlan{};
        pl.cfg = (M <= 2048) ? 4 : 0;
    }
    const int bw = block_window_w(a);
    if ((row_window_ok(a) || bw) && g_wforce_cfg != -2) {   // override cfg -2: keep the LDS-staged kernels (A/B comparison)
        pl.cfg = bw ? (bw == 8 ? 10 : (bw == 4 ? 11 : 12)) : ((T == 1) ? 9 : (a.g.tap_dx > 0 ? 7 : 8));
        pl.WN = 1; pl.WC = 1; pl.TG = T; pl.ntg = 1;
        long long tiles = (long long)(a.N / 32) * (a.C / 32);
        long long want_splits = (256 / g_wplan_div + tiles - 1) / tiles;      // sweep: ~one workgroup per CU, longer pixel runs per wave
        if (g_wforce_split > 0) want_splits = g_wforce_split;
        long long max_splits = (M + 127) / 128;
        long long ns = want_splits < 1 ? 1 : want_splits;
        if (ns > max_splits) ns = max_splits;
        // four waves per workgroup; eight (two per SIMD, same workgroup count and slab traffic) measured 1 % faster
        // standalone and no different inside the step, so it stays a lab switch (MTD_WGRAD_NW=8)
        pl.nw = (g_wforce_nw == 8 && bw != 8) ? 8 : 4;    // (the 8x8 window kernel needs more than 256 registers)
        long long ppw = (M + ns * pl.nw - 1) / (ns * pl.nw);
        ppw = ((ppw + 31) / 32) * 32;
        ns = (M + ppw * pl.nw - 1) / (ppw * pl.nw);
        pl.ppw = (int)ppw;
        pl.nsplit = (int)ns;
        return pl;
    }
    if (g_wforce_cfg >= 0 && g_wforce_cfg < NWCFG && a.N % (32 * kWcfgWN[g_wforce_cfg]) == 0 && a.C % (32 * kWcfgWC[g_wforce_cfg]) == 0)
        pl.cfg = g_wforce_cfg;
    pl.WN = kWcfgWN[pl.cfg];
    pl.WC = kWcfgWC[pl.cfg];
    pl.TG = kWcfgTG[pl.cfg];
    pl.ntg = (T + pl.TG - 1) / pl.TG;
    pl.nw = 4;
    long long tiles = (long long)(a.N / (32 * pl.WN)) * (a.C / (32 * pl.WC)) * pl.ntg;
    // aim for >= 512 workgroups; every wave gets a multiple of 32 pixels
    long long want_splits = (512 / g_wplan_div + tiles - 1) / tiles;
    if (g_wforce_split > 0) want_splits = g_wforce_split;
    long long max_splits = (M + 127) / 128;             // at least 32 px per wave
    long long ns = want_splits < 1 ? 1 : want_splits;
    if (ns > max_splits) ns = max_splits;
    long long ppw = (M + ns * 4 - 1) / (ns * 4);
    ppw = ((ppw + 31) / 32) * 32;
    ns = (M + ppw * 4 - 1) / (ppw * 4);
    pl.ppw = (int)ppw;
    pl.nsplit = (int)ns;
    return pl;
}

bool is_direct(const mtd_wgrad_args& a) { return (a.N % 32) || (a.C % 32); }

int check_wargs(const mtd_wgrad_args& a) {
    if (!a.p || !a.q || !a.dw) return MTD_EINVAL;
    if (a.N <= 0 || a.C <= 0) return MTD_EINVAL;
    if (is_direct(a)) {
        if (a.N != 1 && a.C != 1) return MTD_EINVAL;
        const mtd_geom& g = a.g;
        if (g.B <= 0 || g.TH <= 0 || g.TW <= 0 || g.TH * g.TW > 16) return MTD_EINVAL;
        if (a.p_ld < a.N || a.q_ld < a.C) return MTD_EINVAL;
        return MTD_OK;
    }
    const mtd_geom& g = a.g;
    if (g.B <= 0 || g.IH <= 0 || g.IW <= 0 || g.OH <= 0 || g.OW <= 0) return MTD_EINVAL;
    if (g.TH <= 0 || g.TW <= 0 || g.TH * g.TW > 16) return MTD_EINVAL;
    if (geom_pixels(g) > (1ll << 30)) return MTD_EINVAL;
    if (a.p_ld < a.N || a.q_ld < a.C || (a.p_ld % 4) || (a.q_ld % 4)) return MTD_EINVAL;
    if (!aligned16(a.p) || !aligned16(a.q)) return MTD_EALIGN;
    if (a.half_scale && (!a.half_scale2 || a.m_first <= 0 || (a.m_first % 32) || a.m_first >= geom_pixels(g))) return MTD_EINVAL;
    return MTD_OK;
}

constexpr int GS = 64;   // slabs summed per reduce stage (eight loads in flight per thread: 64 slabs cost less than a second launch)

size_t wgrad_ws_floats(const mtd_wgrad_args& a, int nsplit) {
    const long long T = a.g.TH * a.g.TW;
    const long long stride = T * a.N * a.C + a.N;
    long long total = (long long)nsplit * stride;
    long long ns = nsplit;
    while (ns > GS) {           // intermediate stages
        ns = (ns + GS - 1) / GS;
        total += ns * stride;
    }
    return (size_t)total;
}

}  // namespace

#ifndef MTD_NO_API      // (conv_c32_bwd.hip includes this file for its kernels and helpers only)
int mtd_direct_wgrad_launch(const mtd_wgrad_args* a, int* nslab_out, long long slab_stride, void* stream);
int mtd_direct_wgrad_nslab(const mtd_wgrad_args* a);

extern "C" int mtd_conv_wgrad_override(int cfg, int nsplit) {
    g_wforce_cfg = cfg;
    g_wforce_split = nsplit;
    return MTD_OK;
}

extern "C" size_t mtd_conv_wgrad_ws_bytes(const mtd_wgrad_args* a) {
    if (!a || check_wargs(*a) != MTD_OK) return 0;
    if (is_direct(*a)) return wgrad_ws_floats(*a, mtd_direct_wgrad_nslab(a)) * sizeof(float);
    WPlan pl = make_wplan(*a);
    return wgrad_ws_floats(*a, pl.nsplit) * sizeof(float);
}

// The plan's kernel for these arguments: index into the profiler's weight-gradient name table (16 = the Winograd kernel, which
// executes 4/9 of the layer's multiplications), -1 for the direct (vector-ALU) kernels, MTD_EINVAL for invalid arguments.
// Host-side flop accounting (kernels.wgrad) asks this instead of mirroring the plan.
extern "C" int mtd_conv_wgrad_plan_cfg(const mtd_wgrad_args* a) {
    if (!a || check_wargs(*a) != MTD_OK) return MTD_EINVAL;
    if (is_direct(*a)) return -1;
    return make_wplan(*a).cfg;
}

// plans whose kernels implement the pair form (pair_select): wgrad_kernel<> (0-6), the block-window kernels (10-12), the all-taps
// kernel (13), the stride-2 halo-window kernel (15).  (16, the Winograd kernel, has its own: mtd_conv_wgrad_pair.)
static bool wgrad_cfg_pairs(int cfg) { return (cfg >= 0 && cfg < NWCFG) || (cfg >= 10 && cfg <= 13) || cfg == 15; }

// the slab-producing kernel of one layer; fills p, nsplit, direct
// (rows != nullptr: also run the forward row transform `rows` describes, inside the same launch if the plan is the row-window
// kernel on one (n, c) tile -- *rows_done says whether it was)
// (pair: a describes ONE of two equal problems laid out back to back in p / q; the launch covers both, see WgradParams::pair_ns)
// plans whose kernels apply mtd_wgrad_args.half_scale: wgrad_kernel<> (0-6), the block-window kernels (10-12), the all-taps kernel (13)
static bool wgrad_cfg_half_scale(int cfg) { return (cfg >= 0 && cfg < NWCFG) || (cfg >= 10 && cfg <= 13); }

extern "C" int mtd_conv_wgrad_half_scale_ok(const mtd_wgrad_args* a) {
    if (!a || !a->half_scale || check_wargs(*a) != MTD_OK || is_direct(*a)) return 0;
    return wgrad_cfg_half_scale(make_wplan(*a).cfg) ? 1 : 0;
}

static int wgrad_partial(const mtd_wgrad_args* a, void* stream, WgradParams& p, int& nsplit, bool& direct_out,
                         const RowsArgs* rows = nullptr, bool* rows_done = nullptr, bool pair = false) {
    if (!a) return MTD_EINVAL;
    static const int env_nw = [] { const char* e = mtd_lab_env("MTD_WGRAD_NW"); return e ? atoi(e) : 0; }();
    if (env_nw == 4 || env_nw == 8) g_wforce_nw = env_nw;
    int rc = check_wargs(*a);
    if (rc != MTD_OK) return rc;
    const bool direct = is_direct(*a);
    direct_out = direct;
    WPlan pl{};
    if (direct) nsplit = mtd_direct_wgrad_nslab(a);
    else if (!pair) { pl = make_wplan(*a); nsplit = pl.nsplit; }
    // mtd_wgrad_args.half_scale: the register-operand kernels only (their K loops walk whole 32-pixel chunks of one half)
    if (a->half_scale && (direct || pair || rows || !wgrad_cfg_half_scale(pl.cfg))) return MTD_EINVAL;
    if (pair) {
        if (direct) return MTD_EINVAL;
        pl = make_wplan(*a, 2);
        if (!wgrad_cfg_pairs(pl.cfg)) return MTD_EINVAL;
        p.pair_ns = pl.nsplit;
        p.pair_p_off = geom_pixels(a->g) * a->p_ld;
        p.pair_q_off = (long long)a->g.B * a->g.IH * a->g.IW * a->q_ld;
        pl.nsplit *= 2;
        nsplit = pl.nsplit;
    }
    if (!a->ws || a->ws_bytes < wgrad_ws_floats(*a, nsplit) * sizeof(float)) return MTD_EWS;
    p.a = *a;
    p.M = (int)geom_pixels(a->g);
    p.T = a->g.TH * a->g.TW;
    p.nslab = nsplit;
    p.slab_stride = (long long)p.T * a->N * a->C + a->N;
    {
        const mtd_geom& gg = a->g;
        for (int t = 0; t < gg.TH * gg.TW; ++t) {
            const int ty = t / gg.TW, tx = t % gg.TW;
            p.tap_dy[t] = ty * gg.tap_dy;
            p.tap_dx[t] = tx * gg.tap_dx;
            p.tap_delta[t] = (int)((((long long)(ty * gg.tap_dy) * gg.IW + tx * gg.tap_dx) * a->q_ld) * 4);
        }
        const long long pb = (((long long)p.M - 1) * a->p_ld + a->N) * 4;
        const long long qb = (((long long)gg.B * gg.IH * gg.IW - 1) * a->q_ld + a->C) * 4;
        if (pb >= (1ll << 31) || qb >= (1ll << 31)) return MTD_EINVAL;
        p.p_bytes = (unsigned)pb;
        p.q_bytes = (unsigned)qb;
    }
    hipStream_t s = (hipStream_t)stream;
    if (direct) {
        p.ppw = 0;
        p.nCt = 1;
        int ns2 = 0;
        rc = mtd_direct_wgrad_launch(a, &ns2, p.slab_stride, stream);
        if (rc != MTD_OK) return rc;
        if (ns2 != nsplit) return MTD_EINVAL;
    } else {
        p.ppw = pl.ppw;
        p.nCt = a->C / (32 * pl.WC);
        dim3 grid(pl.nsplit, (a->N / (32 * pl.WN)) * p.nCt, pl.ntg);
        const int np = pair ? 2 : 1;         // problems in this launch
        const int prof = mtd_prof_begin(1, pl.cfg, pl.nsplit, np * geom_pixels(a->g), a->N, a->C, a->g.TH * a->g.TW, s,
                                            4.0 * np * ((double)geom_pixels(a->g) * a->N + (double)a->g.B * a->g.IH * a->g.IW * a->C + (double)a->g.TH * a->g.TW * a->N * a->C));
        // Row-window kernel: 64 KB of dynamic LDS nobody uses caps it at ONE workgroup (one wave per SIMD) per CU.  Alone in
        // a stream that changes nothing (generator step: 31.7 us per launch either way); in the full step, where it runs on a
        // side stream beside the data-gradient chain, a second workgroup on a CU took the slots of the other stream's
        // kernel: 44.47 -> 44.03 ms per step (three A/B runs each, tools/wgrad_pad_full.sh).  MTD_WGRAD_LDS_PAD=0 is the old launch.
        static const unsigned lds_pad = [] { const char* e = mtd_lab_env("MTD_WGRAD_LDS_PAD"); return e ? (unsigned)atoi(e) : 65536u; }();
        if (pl.cfg == 16) {
            WgradWinoParams wp;
            wp.w = p;
            wp.tiles_x = a->g.OW / 2;
            wp.tiles_per_image = (a->g.OH / 2) * wp.tiles_x;
            wp.ntiles = a->g.B * wp.tiles_per_image;
            wp.chunks_per_split = pl.ppw;
            wp.ns_first = wp.first_tiles = 0;
            wp.p_add = nullptr;
            MTD_LAUNCH(wgrad_wino_kernel, dim3(pl.nsplit, (a->N / 64) * (a->C / 64)), dim3(512), 0, s, wp);
            mtd_prof_end(prof, s);
            MTD_LAUNCH_CHECK();
            return MTD_OK;
        }
        if (pl.cfg == 19) {
            WgradWinoParams wp;
            wp.w = p;
            wp.tiles_x = a->g.OW / 2;
            wp.tiles_per_image = (a->g.OH / 2) * wp.tiles_x;
            wp.ntiles = a->g.B * wp.tiles_per_image;
            wp.chunks_per_split = pl.ppw;
            wp.ns_first = wp.first_tiles = 0;
            wp.p_add = nullptr;
            MTD_LAUNCH(wgrad_wino32_kernel, dim3(pl.nsplit), dim3(512), 0, s, wp, a->g.tap_dy < 0 ? 1 : 0);
            mtd_prof_end(prof, s);
            MTD_LAUNCH_CHECK();
            return MTD_OK;
        }
        if (pl.cfg == 18) {
            WgradWinoParams wp;
            wp.w = p;
            wp.tiles_x = (a->g.OW + 2) / 3;
            wp.tiles_per_image = ((a->g.OH + 2) / 3) * wp.tiles_x;
            wp.ntiles = a->g.B * wp.tiles_per_image;
            wp.chunks_per_split = pl.ppw;
            wp.ns_first = wp.first_tiles = 0;
            wp.p_add = nullptr;
            MTD_LAUNCH(wgrad_wino_s2_kernel, dim3(pl.nsplit, (unsigned)wgrad_wino_s2_blocks(*a)), dim3(512), 0, s, wp);
            mtd_prof_end(prof, s);
            MTD_LAUNCH_CHECK();
            return MTD_OK;
        }
        switch (pl.cfg) {
            case 0: MTD_LAUNCH((wgrad_kernel<1, 1, 9>), grid, dim3(256), 0, s, p); break;
            case 1: MTD_LAUNCH((wgrad_kernel<1, 1, 4>), grid, dim3(256), 0, s, p); break;
            case 2: MTD_LAUNCH((wgrad_kernel<2, 2, 1>), grid, dim3(256), 0, s, p); break;
            case 3: MTD_LAUNCH((wgrad_kernel<1, 1, 8>), grid, dim3(256), 0, s, p); break;
            case 4: MTD_LAUNCH((wgrad_kernel<1, 1, 3>), grid, dim3(256), 0, s, p); break;
            case 7: if (pl.nw == 8) MTD_LAUNCH((wgrad_row_kernel<3, 3, 1, 8>), grid, dim3(512), 0, s, p);
                    else if (rows && grid.y == 1 && grid.z == 1) {
                        const dim3 fg(grid.x + (unsigned)((rows->npairs + 7) / 8));
                        MTD_LAUNCH((wgrad_row_rfft_kernel<1>), fg, dim3(256), 0, s, p, (int)grid.x, *rows);
                        *rows_done = true;
                    } else MTD_LAUNCH((wgrad_row_kernel<3, 3, 1, 4>), grid, dim3(256), lds_pad, s, p);
                    break;
            case 8: if (pl.nw == 8) MTD_LAUNCH((wgrad_row_kernel<3, 3, -1, 8>), grid, dim3(512), 0, s, p);
                    else if (rows && grid.y == 1 && grid.z == 1) {
                        const dim3 fg(grid.x + (unsigned)((rows->npairs + 7) / 8));
                        MTD_LAUNCH((wgrad_row_rfft_kernel<-1>), fg, dim3(256), 0, s, p, (int)grid.x, *rows);
                        *rows_done = true;
                    } else MTD_LAUNCH((wgrad_row_kernel<3, 3, -1, 4>), grid, dim3(256), lds_pad, s, p);
                    break;
            case 9: if (pl.nw == 8) MTD_LAUNCH((wgrad_row_kernel<1, 1, 1, 8>), grid, dim3(512), 0, s, p);
                    else MTD_LAUNCH((wgrad_row_kernel<1, 1, 1, 4>), grid, dim3(256), 0, s, p);
                    break;
            case 10: MTD_LAUNCH((wgrad_blk_kernel<8, 4>), grid, dim3(256), 0, s, p); break;
            case 11: if (pl.nw == 8) MTD_LAUNCH((wgrad_blk_kernel<4, 8>), grid, dim3(512), 0, s, p);
                     else MTD_LAUNCH((wgrad_blk_kernel<4, 4>), grid, dim3(256), 0, s, p);
                     break;
            case 12: if (pl.nw == 8) MTD_LAUNCH((wgrad_blk_kernel<2, 8>), grid, dim3(512), 0, s, p);
                     else MTD_LAUNCH((wgrad_blk_kernel<2, 4>), grid, dim3(256), 0, s, p);
                     break;
            case 13: MTD_LAUNCH(wgrad_taps_kernel, grid, dim3(256), 0, s, p); break;
            case 15: {
                static const int env_db = [] { const char* e = mtd_lab_env("MTD_WGRAD_S2_DB"); return e ? atoi(e) : 0; }();
                if (env_db) MTD_LAUNCH((wgrad_s2_kernel<true>), grid, dim3(256), 0, s, p);
                else MTD_LAUNCH((wgrad_s2_kernel<false>), grid, dim3(256), 0, s, p);
                break;
            }
            case 5: MTD_LAUNCH((wgrad_kernel<1, 1, 1>), grid, dim3(256), 0, s, p); break;
            default: MTD_LAUNCH((wgrad_kernel<2, 2, 3>), grid, dim3(256), 0, s, p); break;
        }
        mtd_prof_end(prof, s);
        MTD_LAUNCH_CHECK();
    }
    return MTD_OK;
}

extern "C" int mtd_conv_wgrad_slabs(const mtd_wgrad_args* a, int* nslab, long long* slab_stride, void* stream) {
    if (!a || !nslab || !slab_stride) return MTD_EINVAL;
    if (is_direct(*a)) return MTD_EINVAL;           // vector-ALU layers reduce immediately (mtd_conv_wgrad)
    WgradParams p;
    int nsplit = 0;
    bool direct = false;
    int rc = wgrad_partial(a, stream, p, nsplit, direct);
    if (rc != MTD_OK) return rc;
    *nslab = nsplit == 1 ? 0 : nsplit;              // single split: the kernel wrote dw / db itself
    *slab_stride = p.slab_stride;
    return MTD_OK;
}

extern "C" int mtd_conv_wgrad_slabs_rfft(const mtd_wgrad_args* a, int* nslab, long long* slab_stride, const float* x, int x_ld, float* R,
                                         int B, int col_weight, void* stream) {
    if (!a || !nslab || !slab_stride || !x || !R || B <= 0 || x_ld < 32) return MTD_EINVAL;
    if (is_direct(*a)) return MTD_EINVAL;
    WgradParams p;
    int nsplit = 0;
    bool direct = false, rows_done = false;
    const RowsArgs rows{x, x_ld, R, B * 32, col_weight};
    int rc = wgrad_partial(a, stream, p, nsplit, direct, &rows, &rows_done);
    if (rc != MTD_OK) return rc;
    *nslab = nsplit == 1 ? 0 : nsplit;
    *slab_stride = p.slab_stride;
    if (!rows_done) return mtd_rfft_rows(x, x_ld, R, B, col_weight, stream);       // another plan: two launches
    return MTD_OK;
}

static int reduce_desc_ok(const mtd_wgrad_reduce_desc& d) {
    const mtd_wgrad_args& a = d.a;
    if (!a.ws || !a.dw || !aligned16(a.ws) || (a.N % 32) || (a.C % 32) || a.N <= 0 || a.C <= 0) return 0;
    if (d.T <= 0 || d.nslab < 2 || d.nslab > 4096 || (d.slab_stride % 4) || (long long)d.T * a.N * a.C >= (1ll << 31)) return 0;
    if (d.slab_stride < (long long)d.T * a.N * a.C + (a.db ? a.N : 0)) return 0;
    return 1;
}

extern "C" int mtd_conv_wgrad_reduce_blocks(const mtd_wgrad_reduce_desc* d) {
    if (!d || !reduce_desc_ok(*d)) return 0;
    const long long units = ((long long)d->T * d->a.N * d->a.C + (d->a.db ? d->a.N : 0)) / 4;
    return (int)((units + 15) / 16);
}

extern "C" int mtd_conv_wgrad_reduce_multi(const mtd_wgrad_reduce_desc* table_dev, const mtd_wgrad_reduce_desc* table_host, int count,
                                           void* stream) {
    if (!table_dev || !table_host || count <= 0 || count > 4096) return MTD_EINVAL;
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        const int nb = mtd_conv_wgrad_reduce_blocks(&table_host[i]);
        if (nb <= 0 || table_host[i].first_block != (int)blocks) return MTD_EINVAL;
        blocks += nb;
    }
    if (blocks > (1ll << 30)) return MTD_EINVAL;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, table_dev, count);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

static int wgrad_fused_reduce_enabled() {
    static const int env_fused = [] { const char* e = mtd_lab_env("MTD_WGRAD_FUSED_REDUCE"); return e ? atoi(e) : 1; }();
    return env_fused;
}
static int wgrad_reduce_slabs_impl(const WgradParams& p, const float* cur, int ns, float* next, bool direct, hipStream_t s, float* dw2);
static int wgrad_reduce_slabs(const WgradParams& p, const float* cur, int ns, float* next, bool direct, hipStream_t s) {
    return wgrad_reduce_slabs_impl(p, cur, ns, next, direct, s, nullptr);
}
// the 2 ns slabs of a pair launch: slabs 0 .. ns - 1 into p.a.dw, the others into dw2, both bias rows into p.a.db
static int wgrad_reduce_pair(WgradParams& p, const float* cur, int ns, float* next, float* dw2, int accumulate, hipStream_t s) {
    const long long units = p.slab_stride / 4;
    if (ns <= GS || (wgrad_fused_reduce_enabled() && ns <= 1024 && units <= 16 * 4096))
        return wgrad_reduce_slabs_impl(p, cur, ns, next, false, s, dw2);                    // one launch for both
    int rc = wgrad_reduce_slabs_impl(p, cur, ns, next, false, s, nullptr);
    if (rc != MTD_OK) return rc;
    p.a.dw = dw2;
    p.a.accumulate = accumulate | 2;             // the bias gradient of the second range joins the first's
    return wgrad_reduce_slabs_impl(p, cur + (long long)ns * p.slab_stride, ns, next, false, s, nullptr);
}

extern "C" int mtd_conv_wgrad(const mtd_wgrad_args* a, void* stream) {
    WgradParams p;
    int nsplit = 0;
    bool direct = false;
    int rc = wgrad_partial(a, stream, p, nsplit, direct);
    if (rc != MTD_OK) return rc;
    if (!direct && nsplit == 1) return MTD_OK;      // single split: the kernel wrote dw / db itself
    return wgrad_reduce_slabs(p, a->ws, nsplit, a->ws + (long long)nsplit * p.slab_stride, direct, (hipStream_t)stream);
}

// ---- the two batch halves of a paired discriminator pass in ONE launch (discriminator_path.wgrad_sn: each half has its own
// spectral-norm sigma, u, v, so its raw weight gradient is needed by itself).  a describes the whole batch; images
// [0, b_first) give a->dw, images [b_first, B) give dw2; both bias gradients go to a->db (the second is accumulated).
// The slab-producing kernel runs once with its slices aligned to the image ranges -- twice the work per workgroup of two
// launches that each fill the chip, i.e. half the prologues, epilogues, slabs -- and each range's slabs are summed into
// its own gradient.  Winograd plan only (wgrad_wino_kernel); mtd_conv_wgrad_pair_ok says whether a layer qualifies.
static int g_wpair_mode = -1;            // -1: MTD_WGRAD_PAIR (default 1); set by mtd_conv_wgrad_pair_mode (tests, lab)
static bool wgrad_pair_plan(const mtd_wgrad_args& a, int b_first, int& ns_half, int& cps) {
    if (check_wargs(a) != MTD_OK || is_direct(a)) return false;
    if (b_first <= 0 || 2 * b_first != a.g.B) return false;
    static const int env_pair_default = [] { const char* e = mtd_lab_env("MTD_WGRAD_PAIR"); return e ? atoi(e) : 1; }();
    const int env_pair = g_wpair_mode >= 0 ? g_wpair_mode : env_pair_default;
    if (!env_pair) return false;
    mtd_wgrad_args h = a;
    h.g.B = b_first;
    const int cfg = make_wplan(h).cfg;
    if (cfg == 18) {                       // the stride-2 Winograd kernel: the pair form of cfg 16 (slices aligned to the image ranges)
        const long long blocks = wgrad_wino_s2_blocks(a);
        const long long chunks = (wgrad_wino_s2_tiles(a, b_first) + WGW_T - 1) / WGW_T;
        long long ns = (128 + blocks - 1) / blocks;
        if (ns > chunks / 4) ns = chunks / 4;
        if (ns < 1) ns = 1;
        const long long c = (chunks + ns - 1) / ns;
        ns = (chunks + c - 1) / c;
        ns_half = -(int)ns;                // (negative: cfg 18)
        cps = (int)c;
        return true;
    }
    if (cfg != 16) {                       // the general kernels: planned inside wgrad_partial
        // By default only the stride-2 halo-window kernel (cfg 15: `down` layers with output maps of at least 8x8: 14-21 us less
        // per pair).  The small-map kernels lose: their single launches have one pixel split and write the gradient themselves,
        // a pair launch has two slabs per (n, c) tile and a reduce (down4 132 -> 210 us, down6 26 -> 102 us, conv5x 47 -> 66 us;
        // tools/wgrad_pair_probe.py).  MTD_WGRAD_PAIR=2: Winograd kernel only, 3: every kernel with the pair form (lab, tests).
        if (env_pair == 2 || (env_pair != 3 && cfg != 15)) return false;
        ns_half = cps = 0;
        return wgrad_cfg_pairs(cfg);
    }
    const long long blocks = wgrad_wino_blocks(a);
    const long long chunks = (wgrad_wino_tiles(a, b_first) + WGW_T - 1) / WGW_T;
    long long ns = (128 + blocks - 1) / blocks;                  // the two ranges together: about one workgroup per CU
    if (ns > chunks / 4) ns = chunks / 4;
    if (ns < 1) ns = 1;
    const long long c = (chunks + ns - 1) / ns;
    ns = (chunks + c - 1) / c;
    ns_half = (int)ns;
    cps = (int)c;
    return true;
}

// 0: never pair, 1: the default rule, 2: Winograd kernel only, 3: every kernel that has the pair form, -1: back to the
// environment's choice.  Returns the previous mode.
extern "C" int mtd_conv_wgrad_pair_mode(int mode) {
    const int prev = g_wpair_mode;
    g_wpair_mode = mode;
    return prev;
}

// 0: no pair form; 1: mtd_conv_wgrad_pair; 2: also mtd_conv_wgrad_pair_sum (the Winograd plan: a second cotangent added on load)
extern "C" int mtd_conv_wgrad_pair_ok(const mtd_wgrad_args* a, int b_first) {
    int ns, cps;
    if (!a || !wgrad_pair_plan(*a, b_first, ns, cps)) return 0;
    return ns > 0 ? 2 : 1;                 // (the stride-2 Winograd kernel, ns < 0, has no second cotangent)
}

extern "C" size_t mtd_conv_wgrad_pair_ws_bytes(const mtd_wgrad_args* a, int b_first) {
    int ns, cps;
    if (!a || !wgrad_pair_plan(*a, b_first, ns, cps)) return 0;
    if (ns == 0) {
        mtd_wgrad_args h = *a;
        h.g.B = b_first;
        ns = make_wplan(h, 2).nsplit;
    }
    if (ns < 0) ns = -ns;
    return wgrad_ws_floats(*a, 2 * ns) * sizeof(float);
}

extern "C" int mtd_conv_wgrad_pair_sum(const mtd_wgrad_args* a, const float* p_add, float* dw2, int b_first, void* stream);
extern "C" int mtd_conv_wgrad_pair(const mtd_wgrad_args* a, float* dw2, int b_first, void* stream) {
    return mtd_conv_wgrad_pair_sum(a, nullptr, dw2, b_first, stream);
}

// as mtd_conv_wgrad_pair with the gradients taken from a->p + p_add (p_add: same shape, pixel stride and alignment as a->p; NULL:
// none).  Only where mtd_conv_wgrad_pair_ok says 2.
extern "C" int mtd_conv_wgrad_pair_sum(const mtd_wgrad_args* a, const float* p_add, float* dw2, int b_first, void* stream) {
    int ns_half = 0, cps = 0;
    if (!a || !dw2 || a->half_scale || !wgrad_pair_plan(*a, b_first, ns_half, cps)) return MTD_EINVAL;
    if (p_add && (ns_half <= 0 || !aligned16(p_add))) return MTD_EINVAL;
    const bool s2w = ns_half < 0;          // the stride-2 Winograd kernel (cfg 18)
    if (s2w) ns_half = -ns_half;
    if (ns_half == 0) {                    // one of the general kernels over both problems
        mtd_wgrad_args h = *a;
        h.g.B = b_first;
        WgradParams p;
        int ns = 0;
        bool direct = false;
        int rc = wgrad_partial(&h, stream, p, ns, direct, nullptr, nullptr, true);
        if (rc != MTD_OK) return rc;
        return wgrad_reduce_pair(p, a->ws, ns / 2, a->ws + (long long)ns * p.slab_stride, dw2, a->accumulate, (hipStream_t)stream);
    }
    const int nsplit = 2 * ns_half;
    if (!a->ws || a->ws_bytes < wgrad_ws_floats(*a, nsplit) * sizeof(float)) return MTD_EWS;
    WgradWinoParams wp;
    WgradParams& p = wp.w;
    p.a = *a;
    p.M = (int)geom_pixels(a->g);
    p.T = s2w ? 16 : 9;
    p.nslab = nsplit;
    p.slab_stride = (long long)p.T * a->N * a->C + a->N;
    for (int t = 0; t < 16; ++t) p.tap_dy[t] = p.tap_dx[t] = p.tap_delta[t] = 0;
    {
        const long long pb = (((long long)p.M - 1) * a->p_ld + a->N) * 4;
        const long long qb = (((long long)a->g.B * a->g.IH * a->g.IW - 1) * a->q_ld + a->C) * 4;
        if (pb >= (1ll << 31) || qb >= (1ll << 31)) return MTD_EINVAL;
        p.p_bytes = (unsigned)pb;
        p.q_bytes = (unsigned)qb;
    }
    p.ppw = cps;
    p.nCt = a->C / 64;
    wp.tiles_x = s2w ? (a->g.OW + 2) / 3 : a->g.OW / 2;
    wp.tiles_per_image = (s2w ? (a->g.OH + 2) / 3 : a->g.OH / 2) * wp.tiles_x;
    wp.ntiles = a->g.B * wp.tiles_per_image;
    wp.chunks_per_split = cps;
    wp.ns_first = ns_half;
    wp.first_tiles = b_first * wp.tiles_per_image;
    wp.p_add = p_add;
    hipStream_t s = (hipStream_t)stream;
    const int prof = mtd_prof_begin(1, s2w ? 18 : 16, nsplit, geom_pixels(a->g), a->N, a->C, p.T, s,
                                    4.0 * ((double)geom_pixels(a->g) * a->N + (double)a->g.B * a->g.IH * a->g.IW * a->C + 2.0 * p.T * a->N * a->C));
    if (s2w) MTD_LAUNCH(wgrad_wino_s2_kernel, dim3(nsplit, (unsigned)wgrad_wino_s2_blocks(*a)), dim3(512), 0, s, wp);
    else MTD_LAUNCH(wgrad_wino_kernel, dim3(nsplit, (a->N / 64) * (a->C / 64)), dim3(512), 0, s, wp);
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    return wgrad_reduce_pair(p, a->ws, ns_half, a->ws + (long long)nsplit * p.slab_stride, dw2, a->accumulate, s);
}

// staged, order-fixed reduction of ns slabs at cur into p.a.dw / p.a.db (staging area: next)
// (dw2 != nullptr: the pair form of the one-launch paths, see wgrad_reduce_pair)
static int wgrad_reduce_slabs_impl(const WgradParams& p, const float* cur, int ns, float* next, bool direct, hipStream_t s, float* dw2) {
    const mtd_wgrad_args* a = &p.a;
    const long long count = p.slab_stride;
    const bool vec = !direct;      // MFMA layers: N, C multiples of 32, so every slab offset is a multiple of 4 floats
    if (vec && (!aligned16(a->ws) || (long long)p.T * a->N * a->C >= (1ll << 31))) return MTD_EINVAL;   // float4 reads, 32-bit indices
    const long long units = vec ? count / 4 : count;
    // 16 < ns <= 1024 slabs of a layer small enough that 16-column workgroups still fill the chip's launch slots quickly:
    // one fused launch.  (Large layers have few slabs and keep the one-thread-per-float4 finish kernel.)
    const int env_fused = wgrad_fused_reduce_enabled();
    if (vec && env_fused && ns > 16 && ns <= 1024 && units <= 16 * 4096) {
        const int bx = (int)((units + 15) / 16);
        if (ns > 128) hipLaunchKernelGGL((wgrad_reduce_finish_kernel<8>), dim3(bx, dw2 ? 2 : 1), dim3(1024), 0, s, p, cur, ns, p.slab_stride, dw2);
        else hipLaunchKernelGGL((wgrad_reduce_finish_kernel<4>), dim3(bx, dw2 ? 2 : 1), dim3(256), 0, s, p, cur, ns, p.slab_stride, dw2);
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    static const int env_scalar2 = [] { const char* e = mtd_lab_env("MTD_WGRAD_SCALAR2"); return e ? atoi(e) : 1; }();
    if (env_scalar2 && !vec && !dw2 && ns > GS && ns <= 16 * GS) {          // thin layers: both stages in one launch (same association)
        hipLaunchKernelGGL(wgrad_finish_scalar2_kernel, dim3((unsigned)((count + 15) / 16)), dim3(256), 0, s, p, cur, ns, p.slab_stride, GS);
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    if (dw2 && ns > GS) return MTD_EINVAL;       // (wgrad_reduce_pair sends only the one-launch cases here)
    while (ns > GS) {
        int ng = (ns + GS - 1) / GS;
        int bx = (int)((units + 255) / 256);
        if (bx > 1024) bx = 1024;
        if (vec) hipLaunchKernelGGL(slab_group_sum_kernel, dim3(bx, ng), dim3(256), 0, s, cur, next, ns, GS, count, p.slab_stride, p.slab_stride);
        else hipLaunchKernelGGL(slab_group_sum_scalar_kernel, dim3(bx, ng), dim3(256), 0, s, cur, next, ns, GS, count, p.slab_stride, p.slab_stride);
        MTD_LAUNCH_CHECK();
        cur = next;
        next = next + (long long)ng * p.slab_stride;
        ns = ng;
    }
    {
        int bx = (int)((units + 255) / 256);
        if (bx > 2048) bx = 2048;
        if (vec) hipLaunchKernelGGL(wgrad_finish_kernel, dim3(bx, dw2 ? 2 : 1), dim3(256), 0, s, p, cur, ns, p.slab_stride, dw2);
        else hipLaunchKernelGGL(wgrad_finish_scalar_kernel, dim3(bx), dim3(256), 0, s, p, cur, ns, p.slab_stride);
        MTD_LAUNCH_CHECK();
    }
    return MTD_OK;
}
#endif  // MTD_NO_API
