// Winograd F(2x2, 3x3) convolution / data gradient on fp32 MFMA for the 3x3 stride-1 "same" layers with >= 64 channels
// (arch/Ours/networks.py:181-221 trunk conv{l}1/2, :230-301 decoder *_dconv{l}1/2, and their autograd data gradients).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplications per output tile and (c, n) pair instead of 36: 2.25x fewer MFMA flops than the implicit GEMM of
// conv_igemm.hip, whose large layers already run at the rate this chip sustains for fp32 MFMA loops that stream operands
//  -- the one lever left there is the count of multiplications.  fp32 throughout; F(2x2, 3x3) has transform
// entries 0, +-1, +-1/2 and costs ~1e-6 relative error, far inside the 1e-3 parity bound.
//
// GEMM view: for each of the 16 transform positions xi an independent  M/4 (tiles) x N x C  product
//     acc_xi[tile][n] += U_xi[tile][c] * W_xi[c][n],    U = B^T d B (input transform),  W = G g G^T (weight transform).
// A 512-thread workgroup owns 32 tiles (= 128 output pixels) x BN = 32 NB output channels for ALL 16 positions: wave w holds
// positions 2w, 2w+1 as 2 x NB accumulator blocks of 32 x 32 (NB = 4: 128 registers).  K runs in chunks of 8 channels:
//   * weights: transformed once per optimizer step by wino_weights_kernel into  Uw[xi][C/8][N][8]  -- a chunk's
//     (n, 8 c) block is contiguous, so a wave's B fragments for one (xi, n block) are ONE fully coalesced 1 KB load
//     (lane (n, kh) takes channels 4 kh .. 4 kh + 3: k-step s of the chunk uses channels (s, 4 + s)); straight from L2
//     into registers, the next chunk's loads in flight under this chunk's MFMAs.  No LDS for weights.
//   * input: the transform needs VALU work, so U goes through LDS: thread (tile, channel) loads its own 4 x 4 patch from
//     global memory (buffer loads: out-of-image pixels are out-of-range offsets and arrive as zeros = the padding; the
//     2x overlap of neighbouring patches is served by L1 / L2), applies B^T . B (32 additions) and writes the 16 values
//     to  As[xi][tile][8 c]; the A fragment of a k-chunk is one ds_read_b128 per position.  Waves 0-3 transform the even
//     chunks, waves 4-7 the odd ones, each two chunks ahead of the MFMAs; one workgroup barrier per chunk.
//   * epilogue: the 16 positions of a tile sit in 8 different waves, so the accumulators meet in LDS one n block at a
//     time (16 x 32 x 32 floats), thread (tile, n) applies A^T . A (24 additions) and runs the conv epilogue of
//     conv_igemm.hip (scale, bias, adds, activation, mask -- epilogue_value()) on its 2 x 2 output pixels.
//   * split-K over channel ranges for grids that would leave CUs empty: partial OUTPUT tiles (the transform is linear) go
//     to slabs [split][M][N] and conv_igemm.hip's splitk_epilogue_kernel finishes them, as for the implicit GEMM.
// Roofline: fp32 MFMA.  Algorithmic flops of the layer = 2 M N C 9 (what the caller asked for); executed MFMA flops =
// 2 M N C 4; both are reported to the launch profiler's reader (bench.py).
//
// Round 4: F(2x4, 3x3) -- F(2,3) down the rows as before, F(4,3) ALONG them (template parameter PX = 6, the patch width): 2 x 4
// output tiles from 4 x 6 patches, 24 positions = three per wave, 24 multiplications per 8 output pixels = 3 per pixel instead
// of 4: a quarter of the MFMA work of the large layers gone (executed = 2 M N C 3).  Everything that made the F(2x2) kernel
// fast carries over unchanged because the patch still has FOUR rows: a row per lane of a quad, the column transform by DPP,
// the K loop's single path, the weight fragments straight from L2.  A thread loads six pixels of its row instead of four and
// applies the 6-point B^T of F(4,3) (interpolation points 0, +-1, +-2, inf) in registers.  fp32 error against float64 on the
// discriminator's layers 1.4e-6 .. 2.5e-6 of the output's max-abs (F(2x2): 3e-7 .. 6e-7, direct fp32: 3e-7; a full F(4x4, 3x3)
// would be 6e-6 .. 1.3e-5 -- tools/winograd_f24_probe.py), far inside the 1e-3 parity bound.  Maps whose width is a multiple
// of 4 (and at least wino_plan's threshold) take it; NB = 2 only (three positions x two n blocks = 96 accumulator registers).
#define MTD_NO_API 1
#include "conv_igemm.hip"

#ifndef W_SPREAD
// How the K step's load instructions are issued.  0 (rounds 3-4): in bursts -- PX patch loads + PW NB weight loads at the top of the
// step, PW NB more in the middle.  1: the first burst spread over the first half's MFMAs (two MFMAs per load).  2 (round 5, the
// default): the second burst too, one load per four MFMAs of the second half.  An in-order wave whose load the address unit cannot
// accept issues nothing else meanwhile, and all eight waves of the workgroup burst together after the barrier: the matrix pipe
// stood while the queue drained (MFMA busy 51.5 % + TA busy 47.2 % = 99 % of wino_conv_kernel<2, false, 6>'s time:
// profiles/r5_full_step_pipes_per_kernel.txt).  Measured (profiles/r5_load_spreading.txt): 53.3 -> 49.75 us on the step's average
// launch, full step 29.05 -> 28.47 ms.
#define W_SPREAD 2
#endif
#ifndef W_TRANSPOSED
#define W_TRANSPOSED 1      // accumulator blocks transposed: the epilogue's exchange image is written as 16-byte vectors
#endif
#ifndef W_AUX
#define W_AUX 0      // cache policy of the weight-fragment loads (lab: 2 = nt, streaming: the weights of a workgroup are read once)
#endif
namespace {

constexpr int WT = 32;        // tiles per workgroup
constexpr int WALD = 20;      // LDS row stride (floats) of As[xi][tile][16 c]: 16-byte aligned, b128 reads of 16 rows hit 16 x 4 distinct banks
constexpr int WXLD = 36;      // LDS row stride (floats) of the exchange image X[xi][tile][32 n]: 16-byte aligned rows

// x -> three bf16 pieces (the upper halves of h, m, l) with x = hi + mid + lo EXACTLY: bit masks and exact subtractions
// (conv_winograd_split.h says what for)
__device__ __forceinline__ void w3_split(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const float r = x - __builtin_bit_cast(float, xb & 0xffff0000u);
    const unsigned rb = __builtin_bit_cast(unsigned, r);
    const float q = r - __builtin_bit_cast(float, rb & 0xffff0000u);
    h = xb;
    m = rb;
    l = __builtin_bit_cast(unsigned, q);
}

struct WinoParams {
    IgemmParams p;            // args, M, split-K (c_per_split in channels), buffer extents, out_identity
    int ntiles, tiles_x, tiles_per_image;
    int nchunk;               // C / 8
    int xcd_order;            // 0: workgroup = blockIdx; 1, 2: XCD-contiguous orders (see the kernel)
    unsigned w_bytes;         // extent of the transformed weights (4 PX N C floats)
};

// ---- weight transform:  Uw[xi][C/8][N][8] = (G g G^T)[xi],  g[a][b] = W(n, c, kmap[a * 3 + b])  (a, b = correlation position
// of the tap: input offset -1 + a, -1 + b).  One thread per (n, c).
struct WinoWDesc {
    const float* src; float* dst;
    long long sn, sc, st;     // W(n, c, kidx) = src[n * sn + c * sc + kidx * st]
    int N, C;
    int kmap[9];
    int px;                   // low four bits: patch width of the transform along x: 6 = F(4,3), anything else = F(2,3) (4);
                              // bit 4: the split-bf16 form  Uw3[xi][C/16][plane 0..2][N][16 c]  (conv_winograd_split.h)
};

__global__ __launch_bounds__(256) void wino_weights_kernel(const WinoWDesc* __restrict__ tab, int count) {
    for (int d = blockIdx.y; d < count; d += gridDim.y) {
        const WinoWDesc w = tab[d];
        const long long total = (long long)w.N * w.C;
        const int PXr = (w.px & 15) == 6 ? 6 : 4;
        const bool split = (w.px & 16) != 0;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            // destination order: c8 (split form: c16) fastest within (n), so consecutive threads write consecutive elements
            const int c8 = split ? (int)(i & 15) : (int)(i & 7);
            const long long r = split ? i >> 4 : i >> 3;
            const int n = (int)(r % w.N);
            const int ck = (int)(r / w.N);
            const int c = split ? ck * 16 + c8 : ck * 8 + c8;
            const float* s = w.src + (long long)n * w.sn + (long long)c * w.sc;
            float g[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = s[(long long)w.kmap[a * 3 + b] * w.st];
            // t = G g  (4 x 3), u = t Gx^T (4 x PX);  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] down the rows,
            // Gx = G (F(2,3)) or [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1] (F(4,3)) along them
            float t[4][3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                t[0][b] = g[0][b];
                t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
                t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
                t[3][b] = g[2][b];
            }
            if (split) {
                // the transformed value, split exactly into three bf16 pieces (conv_winograd_split.h), plane p at
                // ((xi (C/16) + ck) 3 + p) N 16 + n 16 + c16  (in bf16 elements)
                unsigned short* d16 = reinterpret_cast<unsigned short*>(w.dst);
                const long long plane = (long long)w.N * 16;
                const long long xs16 = (long long)(w.C / 16) * 3 * plane;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
                    float u[6];
                    if (PXr == 6) {
                        const float e = (t0 + t2) * (1.f / 6.f), f = t1 * (1.f / 6.f);
                        const float h = t0 * (1.f / 24.f) + t2 * (1.f / 6.f), k = t1 * (1.f / 12.f);
                        u[0] = 0.25f * t0; u[1] = -e - f; u[2] = -e + f; u[3] = h + k; u[4] = h - k; u[5] = t2;
                    } else {
                        u[0] = t0; u[1] = 0.5f * (t0 + t1 + t2); u[2] = 0.5f * (t0 - t1 + t2); u[3] = t2; u[4] = u[5] = 0.f;
                    }
                    for (int b = 0; b < PXr; ++b) {
                        unsigned hh, mm, ll;
                        w3_split(u[b], hh, mm, ll);
                        unsigned short* o = d16 + (long long)(a * PXr + b) * xs16 + (long long)ck * 3 * plane + (long long)n * 16 + c8;
                        o[0] = (unsigned short)(hh >> 16);
                        o[plane] = (unsigned short)(mm >> 16);
                        o[2 * plane] = (unsigned short)(ll >> 16);
                    }
                }
                continue;
            }
            const long long xs = (long long)(w.C / 8) * w.N * 8;          // stride between positions xi = a * PX + b
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float* o = w.dst + (((long long)(a * PXr) * (w.C / 8) + ck) * w.N + n) * 8 + c8;
                const float t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
                if (PXr == 6) {
                    const float e = (t0 + t2) * (1.f / 6.f), f = t1 * (1.f / 6.f);           // (t0 + t2) / 6, t1 / 6
                    const float h = t0 * (1.f / 24.f) + t2 * (1.f / 6.f), k = t1 * (1.f / 12.f);
                    o[0] = 0.25f * t0;
                    o[xs] = -e - f;
                    o[2 * xs] = -e + f;
                    o[3 * xs] = h + k;
                    o[4 * xs] = h - k;
                    o[5 * xs] = t2;
                } else {
                    o[0] = t0;
                    o[xs] = 0.5f * (t0 + t1 + t2);
                    o[2 * xs] = 0.5f * (t0 - t1 + t2);
                    o[3 * xs] = t2;
                }
            }
        }
    }
}

// F(2,3) across the quad of lanes that hold the four rows of a patch, four channels at once: row ti of B^T r = r[0] - r[2] |
// r[1] + r[2] | r[2] - r[1] | r[1] - r[3], as  r[perm0] + qsign r[perm1]  with qsign = +1 for ti = 1, else -1.  Written out as
// v_mul_f32_dpp + v_add_f32_dpp -- each takes its permuted operand directly: two instructions per value where the compiler's
// form is two v_mov_b32_dpp and an fma (a vector instruction is paid in full beside fp32 MFMAs: DESIGN 3.8: tools/overlap_probe.hip).  The s_nop covers
// the two wait states between a vector write of r and its first DPP read.
__device__ __forceinline__ f32x4 wino_quad_rows(f32x4 r, float qsign) {
    float u0, u1, u2, u3, t0, t1, t2, t3;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %4, %8, %12 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %5, %9, %12 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %6, %10, %12 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %7, %11, %12 quad_perm:[2,2,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %8, %4 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %9, %5 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %10, %6 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %11, %7 quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(qsign));
    return f32x4{u0, u1, u2, u3};
}

// ---- the convolution
// LEAN (NB = 2): at most 128 registers, so TWO workgroups share a CU (2 x 80 KB of LDS) and one's prologue / epilogue / waits run
// under the other's MFMAs -- for the layers with few K steps (C <= 128) or N = 64, where a lone workgroup per CU spends as long
// outside its K loop as inside.  One weight-fragment register set instead of two.
// PX: patch width.  4 = F(2x2, 3x3): 16 positions, two per wave; 6 = F(2x4, 3x3): 24 positions, three per wave (NB = 2, not LEAN).
// (the kernel's body as a function of the workgroup's place (bx, by, bz) in a grid (gx, gy, gz): wino_conv_kernel runs it on its own
// grid, wino_conv_multi_kernel -- round 6 -- on either half of a grid that holds TWO problems of one shape)
template <int NB, bool LEAN, int PX>
__device__ __forceinline__ void wino_conv_body(const WinoParams& wp, int bx, int by, int bz, const int gx, const int gy, const int gz) {
    constexpr int NP = 4 * PX;            // transform positions xi = PX * (patch row) + (patch column)
    constexpr int PW = NP / 8;            // positions per wave
    constexpr int TWX = PX - 2;           // output pixels per tile row
    static_assert(PX == 4 || (PX == 6 && NB <= 2 && !LEAN), "F(2x4, 3x3): NB <= 2, one workgroup per CU");
    // As: two buffers of NP positions x 32 tiles x 16 channels (row stride WALD floats, 40 / 60 KB each); the exchange image of
    // the epilogue (NP x 32 x WXLD floats = 72 / 108 KB) reuses the same memory after the K loop.
    constexpr int AS_BUF = NP * WT * WALD;
    constexpr int X_SIZE = NP * WT * WXLD;
    __shared__ __attribute__((aligned(16))) float Ls[(2 * AS_BUF > X_SIZE) ? 2 * AS_BUF : X_SIZE];
    const IgemmParams& p = wp.p;
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    // Which (tile block, n block, K slice) is this workgroup's?  Consecutive workgroups go round-robin over the eight XCDs, each
    // with an L2 of its own, so with the plain mapping every L2 fetches every weight slice AND every input slice of the launch.
    // xcd_order gives each XCD one contiguous run of an order in which the workgroups of a run share operands:
    //   1 (weights >= input bytes: the small maps) = the dispatch order itself, tile block fastest -- an XCD holds a few
    //     (n block, K slice) weight slices, each fetched by ONE L2 and used by all its tile blocks;
    //   2 (input > weights: the large maps) = tile block slowest, n block fastest -- an XCD holds a contiguous run of image rows
    //     (the halo rows of neighbouring tile blocks meet in its L2) and each input slice is fetched once for all its n blocks.
    if (wp.xcd_order) {
        const int v = xcd_contiguous_block(bx + gx * (by + gy * bz), gx * gy * gz);
        if (wp.xcd_order == 1) {
            bz = v / (gx * gy);
            const int r = v - bz * (gx * gy);
            by = r / gx;
            bx = r - by * gx;
        } else {
            bx = v / (gy * gz);
            const int r = v - bx * (gy * gz);
            bz = r / gy;
            by = r - bz * gy;
        }
    }
    const int tile0 = bx * WT;
    const int n0 = by * (32 * NB);
    const int zk = bz;
    // K runs in steps of 16 channels (one transform per thread and step), each step two MFMA sub-chunks of 8
    const int st_beg = zk * (p.c_per_split >> 4);
    const int st_end = min(wp.nchunk >> 1, st_beg + (p.c_per_split >> 4));
    const int nst = st_end - st_beg;
    const int st_last = st_end - 1;

    // ---- transform role: thread (tile tt, channel quad tq, patch row ti) -- the four rows of a patch are the four lanes of a
    // quad.  A thread loads its row as PX 16-byte vectors (4 channels x PX pixels: a quarter of the vector-memory and LDS
    // instructions of a dword-per-lane form, which had the K loop waiting on instruction issue), applies B along the row in
    // registers and B^T across the quad with DPP, and stores row ti of the 4 x PX result as PX 16-byte vectors.
    const int ti = tid & 3, tq = (tid >> 2) & 3, tt = tid >> 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    unsigned pbase, pvalid = 0;            // offset of patch pixel (ti, 0), channels 4 tq ..; validity of the row's PX pixels
    {
        const int tg = tile0 + tt;
        const bool tv = tg < wp.ntiles;
        int b, ty, tx;
        pix_decompose(tg, wp.tiles_x, g.OH >> 1, b, ty, tx);           // (image, tile row, tile column)
        const int iy = 2 * ty - 1 + ti;
        // (pixel (iy, TWX tx - 1) may lie outside the image: the offset is formed modulo 2^32, every VALID pixel's is in range)
        pbase = (unsigned)(((((long long)b * g.IH + iy) * g.IW + (TWX * tx - 1)) * a.in_ld + 4 * tq) * 4);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int ix = TWX * tx - 1 + j;
            if (tv & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) pvalid |= 1u << j;
        }
    }
    const int px_b = a.in_ld * 4;          // byte displacement of one pixel
    f32x4 d[PX];
    auto load_patch = [&](int st) {        // step st (absolute, clamped by the caller): channels 16 st + 4 tq .. + 3
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const unsigned vo = ((pvalid >> j) & 1u) ? pbase + (unsigned)(j * px_b) : 0x80000000u;
            d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, st * 64, 0));
        }
    };
    // B^T d B -> As[xi = PX ti + j][tt][4 tq ..]: columns j0, j0 + 1 of the result (the K loop places the pairs between MFMA groups)
    const float qsign = ti == 1 ? 1.f : -1.f;
    auto row_value = [&](int j) -> f32x4 {        // (d B)[j] along the row
        if constexpr (PX == 4) {
            return j == 0 ? d[0] - d[2] : (j == 1 ? d[1] + d[2] : (j == 2 ? d[2] - d[1] : d[1] - d[3]));
        } else {
            // F(4,3):  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
            // (one fma or add per term: 12 instructions per channel for the six values; negations folded into the constants)
            f32x4 r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (j == 0) r[c] = fmaf(4.f, d[0][c], fmaf(-5.f, d[2][c], d[4][c]));
                else if (j == 1) r[c] = fmaf(-4.f, d[2][c], d[4][c]) + fmaf(-4.f, d[1][c], d[3][c]);
                else if (j == 2) r[c] = fmaf(-4.f, d[2][c], d[4][c]) - fmaf(-4.f, d[1][c], d[3][c]);
                else if (j == 3) r[c] = fmaf(2.f, d[3][c] - d[1][c], d[4][c] - d[2][c]);
                else if (j == 4) r[c] = fmaf(-2.f, d[3][c] - d[1][c], d[4][c] - d[2][c]);
                else r[c] = fmaf(4.f, d[1][c], fmaf(-5.f, d[3][c], d[5][c]));
            }
            return r;
        }
    };
    auto transform_cols = [&](float* As, int j0) {
        float* o = As + tt * WALD + 4 * tq;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
            const f32x4 rj = row_value(j);
            // across the quad: (B^T .)[ti] = r[0]-r[2] | r[1]+r[2] | r[2]-r[1] | r[1]-r[3]
            *reinterpret_cast<f32x4*>(o + (PX * ti + j) * (WT * WALD)) = wino_quad_rows(rj, qsign);
        }
    };
    auto transform_store = [&](float* As) {
#pragma unroll
        for (int j0 = 0; j0 < PX; j0 += 2) transform_cols(As, j0);
    };

    // ---- MFMA role: positions PW wave .. PW wave + PW - 1; B fragments straight from the transformed weights
    // (buffer loads: the lane's part of the address is ONE register for the whole launch, position and chunk go through the scalar
    // offset, the n block through the instruction's immediate -- no vector-ALU address arithmetic in the K loop, where every
    // vector instruction is paid in full beside the fp32 MFMAs)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), (short)0, (int)wp.w_bytes, 0x00020000);
    const unsigned w_lane = (unsigned)(((n0 + l31) * 8 + kh * 4) * 4);
    const int xi_stride_b = wp.nchunk * a.N * 32;                  // bytes between positions
    const int w_pos0 = PW * wave * xi_stride_b;
    auto load_b = [&](int ck, f32x4 (&bf)[PW][NB]) {
#pragma unroll
        for (int x = 0; x < PW; ++x)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[x][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, w_lane + (unsigned)(nb * 1024), w_pos0 + x * xi_stride_b + ck * a.N * 32, W_AUX));
    };
    f32x16 acc[PW][NB];
#pragma unroll
    for (int x = 0; x < PW; ++x)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][nb][e] = 0.f;
    f32x4 af[PW];                                                // this sub-chunk's A fragments, one per position
    auto load_af = [&](const float* Ac, int u) {
#pragma unroll
        for (int x = 0; x < PW; ++x) af[x] = *reinterpret_cast<const f32x4*>(Ac + ((PW * wave + x) * WT + l31) * WALD + u * 8 + kh * 4);
    };
    auto mfma_group = [&](int s, const f32x4 (&bf)[PW][NB]) {     // k-step s of the sub-chunk in af: PW NB MFMAs
#pragma unroll
        for (int x = 0; x < PW; ++x)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)      // (W_TRANSPOSED: the block transposed -- rows = channels, columns = tiles; same products, same sums)
                acc[x][nb] = W_TRANSPOSED ? mfma32(bf[x][nb][s], af[x][s], acc[x][nb]) : mfma32(af[x][s], bf[x][nb][s], acc[x][nb]);
    };

    // ---- prologue: step 0 transformed into buffer 0, step 1's patch in flight, step 0's first weights in registers.
    // Every load below is issued unconditionally (indices clamped to the slice's last step): the loop then has ONE path, and
    // the waits the compiler places count exactly the loads that are younger than the registers an MFMA needs.  (With the
    // loads under `if (k + 1 < n)` the path that skips them set the wait counts for all, and every chunk stood for a full
    // memory round trip of its own prefetches: half speed.)
    f32x4 b0[PW][NB], b1[PW][NB];      // (LEAN never touches b1)
    if (nst > 0) {
        load_patch(st_beg);
        load_b(2 * st_beg, b0);
        transform_store(Ls);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): nothing of the prologue is pending inside the loop
    __syncthreads();
    // One K step.  The patch of step j + 1 is requested at the TOP of step j and transformed at its END, between the last MFMA
    // groups: a whole step of MFMAs (8192 clocks of the pipe) covers its way from HBM.  (Requested in the middle of a step and
    // transformed early in the next, the transform stood for it: 25 of 140 us on the 64-channel layers.)  Weight fragments:
    // b1 (second half of this step) at the top, b0 (first half of the next step) after the first half's MFMAs -- each a
    // sub-chunk of MFMAs ahead of its use.  vmcnt counts in issue order: patch, b1 | b0; the scheduling fences keep the
    // compiler from sinking the loads next to their uses.
#pragma unroll 1
    for (int j = 0; j < nst; ++j) {
        const int st = st_beg + j;
        const float* Ac = Ls + (j & 1) * AS_BUF;
        float* An = Ls + ((j + 1) & 1) * AS_BUF;
        load_af(Ac, 0);
        load_patch(min(st + 1, st_last));
        if constexpr (!LEAN) load_b(2 * st + 1, b1);
        if (!(W_SPREAD) || LEAN) __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, b0); mfma_group(1, b0); mfma_group(2, b0); mfma_group(3, b0);
        if constexpr (W_SPREAD && !LEAN) {
            // the step's PX + PW NB load instructions spread over the first half's MFMAs, two MFMAs per load, instead of one
            // burst at the top (W_SPREAD above)
            __builtin_amdgcn_sched_group_barrier(0x100, PW, 0);
#pragma unroll
            for (int i = 0; i < PX + PW * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * PW * NB - 2 * (PX + PW * NB), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LEAN) load_b(2 * st + 1, b0);            // one register set: requested right before its use, the other
        load_af(Ac, 1);                                        // workgroup of the CU has the matrix pipe meanwhile
        __builtin_amdgcn_sched_barrier(0);
        // second half: its 4 PW NB MFMAs and the transform of step j + 1 (vector ALU, DPP, PX LDS stores) in ONE scheduling region,
        // interleaved one MFMA : a few VALU operations -- in separate clusters both waves of a SIMD reach their VALU cluster
        // together (they run the same program between the same barriers) and the matrix pipe stands
        if constexpr (LEAN) {
            mfma_group(0, b0); mfma_group(1, b0); mfma_group(2, b0); mfma_group(3, b0);
            transform_store(An);
        } else {
            load_b(2 * min(st + 1, st_last), b0);              // (the first half's MFMAs have read b0)
            if (W_SPREAD < 2) __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, b1); mfma_group(1, b1); mfma_group(2, b1); mfma_group(3, b1);
            transform_store(An);
        }
        if constexpr (PX == 4) {
#pragma unroll
            for (int i = 0; i < 8 * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NB == 4 ? 3 : 6, 0);     // a few VALU / DPP operations
                if (W_SPREAD >= 2 && !LEAN && (i & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // one of the next step's weight loads
                if ((i & (NB == 4 ? 7 : 3)) == (NB == 4 ? 7 : 3)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // an LDS store
            }
        } else {
            // 24 MFMAs over ~150 vector-ALU / DPP operations (six row values of 4 channels at 2-5 operations each, 24 x 3 across
            // the quad) and six LDS stores
#pragma unroll
            for (int i = 0; i < 4 * PW * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NB == 2 ? 7 : 14, 0);
                if (W_SPREAD >= 2 && (i & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((i & (NB == 2 ? 3 : 1)) == (NB == 2 ? 3 : 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LEAN) load_b(2 * min(st + 1, st_last), b0);
        __syncthreads();
    }

    // ---- epilogue: one n block at a time through the exchange image X[xi][tile][n] (row stride WXLD: 16-byte aligned rows).
    // Thread (tile tl, channel quad nq, output row ei) finishes TWX pixels x 4 channels: A^T m A on 16-byte vectors, then the conv
    // epilogue in epilogue_value()'s order.  Its operands -- up to three tensors -- are requested BEFORE the accumulators go to
    // LDS, so their way from memory is under the exchange; everything moves as 16-byte vectors (the dword form of this
    // epilogue cost the data gradients, with two adds and a mask, up to 60 us per launch).
    const ScalePair sp = load_scale(a);
    const int ei = tid & 1, enq = (tid >> 1) & 7, etl = tid >> 4;
    int epix;                                                      // left pixel of the thread's output row (-1: no such tile)
    {
        const int tg = tile0 + etl;
        int bimg, ty, tx;
        pix_decompose(tg, wp.tiles_x, g.OH >> 1, bimg, ty, tx);
        epix = tg < wp.ntiles ? (bimg * g.OH + 2 * ty + ei) * g.OW + TWX * tx : -1;
    }
    const bool evalid = epix >= 0;
    float esc[TWX];
#pragma unroll
    for (int q = 0; q < TWX; ++q) esc[q] = pick_scale(sp, epix < 0 ? 0 : epix + q);
    const bool vec = (p.wide & 1) != 0;                            // every epilogue operand row 16-byte aligned (wide_epilogue_ok)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {        // (unrolled: a run-time index into the accumulators would put them in scratch memory)
        const int n = n0 + nb * 32 + 4 * enq;
        f32x4 e1[TWX], e2[TWX], em[TWX], bias4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < TWX; ++q) {
            e1[q] = f32x4{-0.0f, -0.0f, -0.0f, -0.0f};
            e2[q] = e1[q];
            em[q] = f32x4{1.f, 1.f, 1.f, 1.f};
        }
        if (p.splitk == 1 && evalid) {
            if (vec) {
                if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n);
                if (a.add1) {
#pragma unroll
                    for (int q = 0; q < TWX; ++q) e1[q] = *reinterpret_cast<const f32x4*>(a.add1 + (long long)(epix + q) * a.add1_ld + n);
                }
                if (a.add2) {
#pragma unroll
                    for (int q = 0; q < TWX; ++q) e2[q] = *reinterpret_cast<const f32x4*>(a.add2 + (long long)(epix + q) * a.add2_ld + n);
                }
                if (a.mask) {
#pragma unroll
                    for (int q = 0; q < TWX; ++q) em[q] = *reinterpret_cast<const f32x4*>(a.mask + (long long)(epix + q) * a.mask_ld + n);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (a.bias) bias4[c] = a.bias[n + c];
#pragma unroll
                    for (int q = 0; q < TWX; ++q) {
                        if (a.add1) e1[q][c] = a.add1[(long long)(epix + q) * a.add1_ld + n + c];
                        if (a.add2) e2[q][c] = a.add2[(long long)(epix + q) * a.add2_ld + n + c];
                        if (a.mask) em[q][c] = a.mask[(long long)(epix + q) * a.mask_ld + n + c];
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < PW; ++x) {
            if constexpr (W_TRANSPOSED) {
                // round 5: the accumulator blocks are computed TRANSPOSED (mfma(b, a): a lane holds, for ITS tile l31, channels
                // 8 g + 4 kh .. + 3 in registers 4 g .. 4 g + 3), so the exchange image is written as four 16-byte vectors per
                // block instead of sixteen dwords: 12 / 8 LDS stores per n block and wave instead of 48 / 32
                float* X = Ls + ((PW * wave + x) * WT + l31) * WXLD + 4 * kh;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<f32x4*>(X + 8 * gq) = f32x4{acc[x][nb][4 * gq], acc[x][nb][4 * gq + 1], acc[x][nb][4 * gq + 2], acc[x][nb][4 * gq + 3]};
            } else {
                float* X = Ls + (PW * wave + x) * (WT * WXLD) + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) X[mfma32_row(e, lane) * WXLD] = acc[x][nb][e];
            }
        }
        __syncthreads();
        f32x4 y[TWX];
        {
            f32x4 t[PX];
#pragma unroll
            for (int b = 0; b < PX; ++b) {
                const float* col = Ls + (b * WT + etl) * WXLD + 4 * enq;              // position xi = PX a + b at col + a * PX * WT * WXLD
                const f32x4 m1 = *reinterpret_cast<const f32x4*>(col + 1 * PX * WT * WXLD);
                const f32x4 m2 = *reinterpret_cast<const f32x4*>(col + 2 * PX * WT * WXLD);
                const f32x4 m03 = *reinterpret_cast<const f32x4*>(col + (ei ? 3 : 0) * PX * WT * WXLD);
                t[b] = ei ? m1 - m2 - m03 : m03 + m1 + m2;                             // row ei of A^T m
            }
            if constexpr (PX == 4) {
                y[0] = t[0] + t[1] + t[2];
                y[1] = t[1] - t[2] - t[3];
            } else {
                // A^T of F(4,3) = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
                const f32x4 s12 = t[1] + t[2], d12 = t[1] - t[2], s34 = t[3] + t[4], d34 = t[3] - t[4];
                y[0] = t[0] + s12 + s34;
                y[1] = d12 + 2.f * d34;
                y[2] = s12 + 4.f * s34;
                y[3] = d12 + 8.f * d34 + t[5];
            }
        }
        if (evalid) {
            if (p.splitk > 1) {
                float* slab = a.ws + (long long)zk * ((long long)p.M * a.N) + n;
#pragma unroll
                for (int q = 0; q < TWX; ++q) {
                    if (vec && (p.wide & 2)) *reinterpret_cast<f32x4*>(slab + (long long)(epix + q) * a.N) = y[q];
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c) slab[(long long)(epix + q) * a.N + c] = y[q][c];
                }
            } else {
#pragma unroll
                for (int q = 0; q < TWX; ++q) {
                    const float sc = esc[q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = y[q][c] * sc + bias4[c];
                        if (a.act == MTD_ACT_RELU_ADD) v = v > 0.f ? v : 0.f;      // the residual operands AFTER the activation
                        v += e1[q][c];
                        v += e2[q][c];
                        y[q][c] = v;
                    }
                }
                if (a.act == MTD_ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < TWX; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[q][c] = y[q][c] > 0.f ? y[q][c] : 0.f;
                } else if (a.act == MTD_ACT_LRELU) {
#pragma unroll
                    for (int q = 0; q < TWX; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[q][c] = y[q][c] > 0.f ? y[q][c] : 0.2f * y[q][c];
                }
                if (a.mask) {
                    const float slope = a.mask_slope;
#pragma unroll
                    for (int q = 0; q < TWX; ++q)
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[q][c] *= (em[q][c] > 0.f) ? 1.f : slope;
                }
#pragma unroll
                for (int q = 0; q < TWX; ++q) {
                    float* o = a.out + (long long)(epix + q) * a.out_ld + n;
                    if (vec) *reinterpret_cast<f32x4*>(o) = y[q];
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = y[q][c];
                }
            }
        }
        __syncthreads();
    }
}

template <int NB, bool LEAN = false, int PX = 4>
__global__ __launch_bounds__(512, LEAN ? 4 : 1) void wino_conv_kernel(const WinoParams wp) {
    wino_conv_body<NB, LEAN, PX>(wp, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, (int)gridDim.y, (int)gridDim.z);
}

// Round 6: TWO or THREE convs of one shape in one grid -- the mirror layers of the discriminator's pixel-level and restoration decoders
// (networks.py:420-467: s_dconv{l}k and r_dconv{l}k have the same channels on the same map; their inputs, weights and epilogue operands
// differ).  On the 2x2 ... 8x8 maps a single layer fills a fraction of the chip even with its split of K; two of them in one launch are
// twice the workgroups for one launch latency (and one slab-sum launch for both).  grid.z = 2 x the split of K: the first half of it
// is problem 0.  (A branch per problem, not an index: a dynamic index into the kernel arguments would put them in scratch.)
constexpr int WINO_MULTI_MAX = 3;
struct WinoMulti { WinoParams p[WINO_MULTI_MAX]; int count; };

template <int NB, bool LEAN = false, int PX = 4>
__global__ __launch_bounds__(512, LEAN ? 4 : 1) void wino_conv_multi_kernel(const WinoMulti mp) {
    const int gz = (int)gridDim.z / mp.count;                  // grid.z = count x the split of K: problem s owns [s gz, (s + 1) gz)
    const int set = __builtin_amdgcn_readfirstlane((int)blockIdx.z / gz);
    const int bz = (int)blockIdx.z - set * gz;
    switch (set) {
        case 0: wino_conv_body<NB, LEAN, PX>(mp.p[0], (int)blockIdx.x, (int)blockIdx.y, bz, (int)gridDim.x, (int)gridDim.y, gz); break;
        case 1: wino_conv_body<NB, LEAN, PX>(mp.p[1], (int)blockIdx.x, (int)blockIdx.y, bz, (int)gridDim.x, (int)gridDim.y, gz); break;
        default: wino_conv_body<NB, LEAN, PX>(mp.p[2], (int)blockIdx.x, (int)blockIdx.y, bz, (int)gridDim.x, (int)gridDim.y, gz); break;
    }
}

#include "conv_wino_c32.h"
#include "conv_winograd_split.h"

int wino_patch_w_of(const mtd_conv_args& a);

// the kernel's domain: 3x3, stride 1, "same" size, even height and width, every tap within one pixel of the output position,
// output pixel == launch pixel, C a multiple of 16, N a multiple of 64 (or C = N = 32 in the F(2x4) form), the input view inside 32-bit byte offsets
bool wino_eligible(const mtd_conv_args& a) {
    const mtd_geom& g = a.g;
    if (g.TH != 3 || g.TW != 3 || g.in_sy != 1 || g.in_sx != 1) return false;
    if (g.IH != g.OH || g.IW != g.OW || (g.OH & 1) || (g.OW & 1)) return false;
    if (!(g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW)) return false;
    for (int i = 0; i < 3; ++i) {
        const int dy = g.off_y + i * g.tap_dy, dx = g.off_x + i * g.tap_dx;
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    }
    if (g.tap_dy == 0 || g.tap_dx == 0) return false;
    if (a.C % 16) return false;
    if (a.out2 && !(a.N == 32 && a.C == 32 && a.mask)) return false;      // (a second output: the persistent 32 -> 32 kernel's MASKED2 form only; mtd_conv_winograd_ok checks that it takes the launch)
    // N a multiple of 64; or the generator's 32 -> 32 channel layers in the F(2x4) form: the persistent kernel of conv_wino_c32.h
    // (wino_c32_takes), else this kernel's 32-channel workgroups (NB = 1).  (Whether conv() sends them here is the host's
    // threshold, kernels.WINO_C32_MIN_HW: whole-slice inference yes, the 64 x 64 training patches no -- DESIGN 3.8.)
    if ((a.N % 64) && !(a.N == 32 && a.C == 32 && wino_patch_w_of(a) == 6)) return false;
    if (a.act == MTD_ACT_RELU_ADD && !((a.N % 64) != 0 && !a.mask)) return false;      // residual after the activation: that form only
    return true;
}

struct WinoPlan { int nb, lean, splitk, c_per_split, px; };

// Which transform along x does this layer take?  4 = the patch width of F(2x2, 3x3), 6 = F(2x4, 3x3): maps whose width is a
// multiple of 4 and at least MTD_WINO_F4_MIN_W (default 8: on the 4-pixel-wide maps a tile row is one tile and the transformed
// weights -- 24 / 9 of the filter instead of 16 / 9 -- are what the launch streams).  MTD_WINO_F4=0 switches the form off.
int g_f4_min_w = -1;          // -1: not yet read from the environment; 0: the form is off
int wino_patch_w(const mtd_conv_args& a);
int wino_patch_w_of(const mtd_conv_args& a) { return wino_patch_w(a); }
int wino_patch_w(const mtd_conv_args& a) {
    if (g_f4_min_w < 0) {
        const char* off = mtd_lab_env("MTD_WINO_F4");
        const char* mw = mtd_lab_env("MTD_WINO_F4_MIN_W");
        g_f4_min_w = (off && atoi(off) == 0) ? 0 : (mw ? atoi(mw) : 8);
    }
    const int pxw = (g_f4_min_w > 0 && (a.g.OW % 4) == 0 && a.g.OW >= g_f4_min_w) ? 6 : 4;
    // the split-bf16 kernel (conv_winograd_split.h) takes every layer whose N is a multiple of 64 (bit 4 of the code); the
    // generator's 32 -> 32 layers keep the fp32 forms (the persistent kernel of conv_wino_c32.h)
    return pxw | ((mtd_option(MTD_OPT_WINO_SPLIT) && (a.N % 64) == 0) ? 16 : 0);
}

WinoPlan wino_plan(const mtd_conv_args& a, int pxcode, int sets = 1) {
    WinoPlan pl{};
    const int px = pxcode & 15;
    const bool split3 = (pxcode & 16) != 0;
    pl.px = px;
    const int tile_px = 2 * (px - 2);                            // output pixels per tile
    pl.nb = (a.N % 128 == 0 && px == 4) ? 4 : 2;
    // F(2x4): a tile block is 256 pixels x 64 channels; where that leaves the grid short (the mid-size maps: 4096 .. 16384 pixels)
    // 32-channel workgroups (NB = 1) can stand in for a split of K -- no slabs, no finishing launch, but every 32 output channels
    // repeat the input transform, and a transform instruction is paid in full beside the fp32 MFMAs.  The form won
    // 0.2 ms per step while the transform cost 170 vector instructions per K step; at 116 the split of K is ahead by 0.15 ms
    // (29.15 against 29.31 ms), so it is off by default now (MTD_WINO_F4_NB1=1: on).
    static const int env_f4_nb1 = [] { const char* e = mtd_lab_env("MTD_WINO_F4_NB1"); return e ? atoi(e) : 0; }();
    if (px == 6 && env_f4_nb1) {
        const long long t = geom_pixels(a.g) / tile_px;
        if (((t + WT - 1) / WT) * (a.N / 64) <= 128 && a.C >= 128) pl.nb = 1;
    }
    if (a.N % 64) pl.nb = 1;                                     // (F(2x4) only: wino_eligible)
    if (split3) pl.nb = 2;                                       // the split-bf16 kernel: 64-channel workgroups only
    // (lab, MTD_WINO_NB2_MAXC=64: the narrow form with its lean variant for layers with four K steps whatever their N -- 5 % less time
    // for those launches (123 -> 116 us, 226 -> 213 us), 0.08 ms per step, but the input is then read per 64 instead of per 128 output
    // channels: 62 -> 80 MB of fabric traffic per launch.  Off.)
    static const int env_nb2_c = [] { const char* e = mtd_lab_env("MTD_WINO_NB2_MAXC"); return e ? atoi(e) : 0; }();
    if (a.C <= env_nb2_c) pl.nb = 2;
    const long long tiles = geom_pixels(a.g) / tile_px;
    long long blocks = ((tiles + WT - 1) / WT) * (a.N / (32 * pl.nb));
    if (blocks < 192 && pl.nb == 4 && a.N % 64 == 0 && !split3) {          // more, narrower workgroups before splitting K
        pl.nb = 2;
        blocks = ((tiles + WT - 1) / WT) * (a.N / 64);
    }
    const int chunks = a.C / 16;                                 // K steps of 16 channels
    // K steps per slice at least: 2 since the end of round 5 (rounds 3-5: 4).  In the concurrent step the layers this frees -- C = 64 ... 128 on
    // grids of 64 ... 128 workgroups -- gain more from the second half of the chip than the extra slab costs: 27.45 -> 27.33 ms in four A/B
    // pairs (1: 27.37 / 27.44, 3: 27.42 / 27.40; a cap on the split or a target of 384 / 512 workgroups loses 0.4 ... 1.8 ms)
    static const int env_sk_steps = [] { const char* e = mtd_lab_env("MTD_WINO_SPLITK_MINSTEPS"); return e ? atoi(e) : 2; }();
    // (sets > 1: the group form.s grid holds that many problems and the split of K is planned for the whole grid: wino_group_sets)
    const long long grid_blocks = blocks * sets;
    int sk = grid_blocks <= 128 ? (int)(256 / grid_blocks) : 1;
    if (sk > chunks / env_sk_steps) sk = chunks / env_sk_steps;
    if (sk > 16) sk = 16;
    if (sk < 1) sk = 1;
    static const int env_sk = [] { const char* e = mtd_lab_env("MTD_WINO_SPLITK"); return e ? atoi(e) : 0; }();
    if (env_sk > 0) sk = env_sk < chunks ? env_sk : chunks;
    const int cps = (chunks + sk - 1) / sk;
    pl.splitk = (chunks + cps - 1) / cps;
    pl.c_per_split = cps * 16;
    // two lean workgroups per CU where a workgroup has few K steps and the grid has at least two per CU (MTD_WINO_LEAN: 0 never,
    // 1 by this rule, 2 whenever NB = 2)
    static const int env_lean = [] { const char* e = mtd_lab_env("MTD_WINO_LEAN"); return e ? atoi(e) : 1; }();
    const long long grid = ((tiles + WT - 1) / WT) * (a.N / (32 * pl.nb)) * pl.splitk;
    pl.lean = px == 4 && pl.nb == 2 && env_lean && (env_lean == 2 || (cps <= 8 && grid >= 512)) && !split3;
    return pl;
}

// The persistent 32 -> 32 channel kernel (conv_wino_c32.h) takes a layer of the F(2x4) form with one residual operand at most,
// no scales, no mask, 16-byte aligned rows everywhere and buffers inside 31-bit byte offsets.  MTD_WINO_C32_KERNEL=0: the
// general kernel's 32-channel workgroups instead (lab switch).
bool wino_c32_takes(const mtd_conv_args& a, int pxcode) {
    const int px = pxcode;       // (a split code, 20 / 22, never matches 6: N % 64 == 0 there)
    static const int env_on = [] { const char* e = mtd_lab_env("MTD_WINO_C32_KERNEL"); return e ? atoi(e) : 1; }();
    if (!env_on || px != 6 || a.C != 32 || a.N != 32) return false;
    if (a.scale || a.scale2 || a.add2 || (a.out2 && !a.mask)) return false;
    if (a.mask && (a.act == MTD_ACT_RELU_ADD || !aligned16(a.mask) || (a.mask_ld % 4))) return false;
    if (a.out2 && (!aligned16(a.out2) || (a.out2_ld % 4))) return false;
    if (!wide_epilogue_ok(a) || !aligned16(a.in) || (a.in_ld % 4)) return false;
    const long long M = geom_pixels(a.g);
    if (M / 8 >= (1ll << 23)) return false;
    if (((M - 1) * a.out_ld + a.N) * 4 >= (1ll << 31)) return false;
    if (a.add1 && ((M - 1) * a.add1_ld + a.N) * 4 >= (1ll << 31)) return false;
    if (a.mask && ((M - 1) * a.mask_ld + a.N) * 4 >= (1ll << 31)) return false;
    if (a.out2 && ((M - 1) * a.out2_ld + a.N) * 4 >= (1ll << 31)) return false;
    return true;
}

// the patch width the transformed weights in a->w were built for travels in a->w_st (6: F(2x4, 3x3); anything else: 4)
// (bit 4: the split-bf16 form)
inline int wino_args_px(const mtd_conv_args& a) { return ((a.w_st & 15) == 6 ? 6 : 4) | ((a.w_st & 16) && a.w_st < 32 ? 16 : 0); }

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------
// Transformed weights of `count` conv views in one launch.  desc[i]: the weight view W(n, c, kidx) = src[n sn + c sc + kidx st]
// and the geometry it will be used with (its taps decide which filter entry sits at which correlation position: a forward
// conv and the data gradient of the same layer need different transforms); dst: 16 * N * C floats, layout [xi][C/8][N][8].
extern "C" size_t mtd_winograd_weight_floats(int N, int C) { return (N > 0 && C > 0) ? (size_t)36 * N * C : 0; }      /* enough for every form (split: 24 N C x 3 bf16) */

extern "C" int mtd_winograd_weights(const mtd_wino_weight_desc* table_dev, const mtd_wino_weight_desc* table_host, int count, void* stream) {
    static_assert(sizeof(mtd_wino_weight_desc) == sizeof(WinoWDesc), "descriptor layouts must agree");
    if (!table_dev || !table_host || count <= 0) return MTD_EINVAL;
    long long most = 0;
    for (int i = 0; i < count; ++i) {
        const mtd_wino_weight_desc& d = table_host[i];
        if (!d.src || !d.dst || d.N <= 0 || d.C <= 0 || (d.C % 8) || !(d.px == 0 || d.px == 4 || d.px == 6 || d.px == 20 || d.px == 22)) return MTD_EINVAL;
        if ((d.px & 16) && (d.C % 16)) return MTD_EINVAL;
        for (int k = 0; k < 9; ++k)
            if (d.kmap[k] < 0 || d.kmap[k] > 15) return MTD_EINVAL;
        const long long t = (long long)d.N * d.C;
        most = t > most ? t : most;
    }
    int gx = (int)((most + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(wino_weights_kernel, dim3(gx, count < 64 ? count : 64), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const WinoWDesc*>(table_dev), count);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// kmap for a geometry: correlation position (a, b) (input offset -1 + a, -1 + b) -> index of the filter entry in the kh x kw plane
extern "C" int mtd_winograd_kmap(const mtd_geom* g, int* kmap9) {
    if (!g || !kmap9 || g->TH != 3 || g->TW != 3) return MTD_EINVAL;
    for (int i = 0; i < 9; ++i) kmap9[i] = -1;
    for (int ty = 0; ty < 3; ++ty)
        for (int tx = 0; tx < 3; ++tx) {
            const int dy = g->off_y + ty * g->tap_dy, dx = g->off_x + tx * g->tap_dx;
            if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return MTD_EINVAL;
            kmap9[(dy + 1) * 3 + (dx + 1)] = (g->ky0 + ty * g->ky_step) * g->KW + (g->kx0 + tx * g->kx_step);
        }
    for (int i = 0; i < 9; ++i)
        if (kmap9[i] < 0) return MTD_EINVAL;
    return MTD_OK;
}

extern "C" int mtd_conv_winograd_ok(const mtd_conv_args* a) {
    if (!a || !a->in || !a->w || !a->out) return 0;
    if (a->C <= 0 || a->N <= 0 || a->in_ld < a->C || a->out_ld < a->N) return 0;
    if (!wino_eligible(*a)) return 0;
    const long long npix = (long long)a->g.B * a->g.IH * a->g.IW;
    if (((npix - 1) * a->in_ld + a->C) * 4 >= (1ll << 31)) return 0;
    if (geom_pixels(a->g) * a->N >= (1ll << 31)) return 0;
    if ((long long)144 * a->N * a->C >= (1ll << 31)) return 0;          // (the transformed weights inside 31-bit byte offsets, split form included)
    // a mask on a 32 -> 32 layer / a second output: only the persistent kernel carries them in this form (the general kernel's
    // 32-channel workgroups take a mask, never a second output -- and the caller's MASKED2 launches must not end up there)
    if (a->out2 && !wino_c32_takes(*a, wino_patch_w(*a))) return 0;
    return 1;
}

// The transform this layer's weights are to be built for (mtd_wino_weight_desc.px, and a->w_st of the conv launch): 6 =
// F(2x4, 3x3), 4 = F(2x2, 3x3); 0 if the layer is not in the Winograd kernel's domain.
extern "C" int mtd_conv_winograd_patch_w(const mtd_conv_args* a) {
    if (!mtd_conv_winograd_ok(a)) return 0;
    return wino_patch_w(*a);
}

// Tuning / test hook: narrowest map that takes F(2x4, 3x3) (0: never; the default is 8, or MTD_WINO_F4_MIN_W / MTD_WINO_F4=0 from
// the environment).  Returns the previous value.  Callers that cache mtd_conv_winograd_patch_w's answers drop them.
extern "C" int mtd_conv_winograd_f4_min_w(int min_w) {
    mtd_conv_args probe{};
    (void)wino_patch_w(probe);                       // (reads the environment once)
    const int old = g_f4_min_w;
    if (min_w >= 0) g_f4_min_w = min_w;
    return old;
}

extern "C" size_t mtd_conv_winograd_ws_bytes(const mtd_conv_args* a) {
    if (!mtd_conv_winograd_ok(a)) return 0;
    const WinoPlan pl = wino_plan(*a, wino_args_px(*a));
    return pl.splitk > 1 ? (size_t)pl.splitk * (size_t)geom_pixels(a->g) * a->N * sizeof(float) : 0;
}

// the launch parameters of one problem (shared by mtd_conv_winograd and mtd_conv_winograd_pair)
static int wino_fill(const mtd_conv_args* a, const WinoPlan& pl, int pxcode, WinoParams& wp) {
    const int px = pxcode & 15;
    const bool split3 = (pxcode & 16) != 0;
    IgemmParams& p = wp.p;
    p.a = *a;
    p.M = (int)geom_pixels(a->g);
    p.splitk = pl.splitk;
    p.c_per_split = pl.c_per_split;
    {
        const long long npix = (long long)a->g.B * a->g.IH * a->g.IW;
        p.in_bytes = (unsigned)(((npix - 1) * a->in_ld + a->C) * 4);
    }
    p.w_bytes = 0;
    for (int t = 0; t < 16; ++t) p.tap_dy[t] = p.tap_dx[t] = p.tap_delta[t] = p.tap_kidx[t] = 0;
    p.out_identity = 1;
    p.out_linear = 1;
    p.xcd_map = 0;
    p.nt_store = 0;
    p.fin = 0;
    p.wide = (wide_epilogue_ok(*a) ? 1 : 0) | ((pl.splitk > 1 && aligned16(a->ws) && (a->N % 4) == 0) ? 2 : 0);
    wp.tiles_x = a->g.OW / (px - 2);
    wp.tiles_per_image = (a->g.OH / 2) * wp.tiles_x;
    wp.ntiles = a->g.B * wp.tiles_per_image;
    wp.nchunk = a->C / 8;
    wp.w_bytes = (unsigned)((long long)4 * px * a->N * a->C * (split3 ? 6 : 4));
    {
        static const int env_xcd = [] { const char* e = mtd_lab_env("MTD_WINO_XCD"); return e ? atoi(e) : -1; }();
        const double wbytes = 4.0 * px * a->C * a->N * 4, ibytes = (double)p.M * a->C * 4;
        wp.xcd_order = env_xcd >= 0 ? env_xcd : (wbytes >= ibytes ? 1 : 2);
    }
    if (pl.splitk > 1) {
        const size_t need = (size_t)pl.splitk * (size_t)p.M * a->N * sizeof(float);
        if (!a->ws || a->ws_bytes < need) return MTD_EWS;
    }
    return MTD_OK;
}

// a: as for mtd_conv_igemm, except that a->w points to the TRANSFORMED weights of this view and geometry
// (mtd_winograd_weights; a->w_sn / w_sc / w_st are ignored).  Same epilogue, same split-K workspace contract.
extern "C" int mtd_conv_winograd(const mtd_conv_args* a, void* stream) {
    if (!mtd_conv_winograd_ok(a)) return MTD_EINVAL;
    if (!aligned16(a->w)) return MTD_EALIGN;
    const int pxcode = wino_args_px(*a);
    const int px = pxcode & 15;
    const bool split3 = (pxcode & 16) != 0;
    if (px == 6 && (a->g.OW % 4)) return MTD_EINVAL;
    if (split3 && (a->N % 64)) return MTD_EINVAL;
    const WinoPlan pl = wino_plan(*a, pxcode);
    WinoParams wp;
    {
        const int rc = wino_fill(a, pl, pxcode, wp);
        if (rc != MTD_OK) return rc;
    }
    IgemmParams& p = wp.p;
    hipStream_t s = (hipStream_t)stream;
    if (wino_c32_takes(*a, pxcode)) {
        C32Params cp;
        cp.wp = wp;
        cp.tiles_y = a->g.OH / 2;
        cp.nblocks = (wp.ntiles + C32_T - 1) / C32_T;
        cp.out_bytes = (unsigned)((((long long)p.M - 1) * a->out_ld + a->N) * 4);
        cp.add_bytes = a->add1 ? (unsigned)((((long long)p.M - 1) * a->add1_ld + a->N) * 4) : 0u;
        cp.mask_bytes = a->mask ? (unsigned)((((long long)p.M - 1) * a->mask_ld + a->N) * 4) : 0u;
        cp.out2_bytes = a->out2 ? (unsigned)((((long long)p.M - 1) * a->out2_ld + a->N) * 4) : 0u;
        const int per = (cp.nblocks + 7) / 8;                    // blocks per XCD; one workgroup per CU: 32 per XCD
        const int slots = per < 32 ? per : 32;
        const dim3 pgrid(8 * slots);
        const int step = C32_T * slots;                          // tiles between two blocks of a workgroup's walk
        cp.d_tx = step % wp.tiles_x;
        cp.d_ty = (step / wp.tiles_x) % cp.tiles_y;
        cp.d_img = step / wp.tiles_per_image;
        const int prof = mtd_prof_begin(0, a->mask ? (a->add1 ? 33 : 32) : (a->add1 ? 26 : 25), 1, p.M, a->N, a->C, 9, s, algorithmic_bytes(a));
        if (a->mask && a->add1) MTD_LAUNCH((wino_c32_kernel<true, true>), pgrid, dim3(512), 0, s, cp);
        else if (a->mask) MTD_LAUNCH((wino_c32_kernel<false, true>), pgrid, dim3(512), 0, s, cp);
        else if (a->add1) MTD_LAUNCH((wino_c32_kernel<true>), pgrid, dim3(512), 0, s, cp);
        else MTD_LAUNCH((wino_c32_kernel<false>), pgrid, dim3(512), 0, s, cp);
        mtd_prof_end(prof, s);
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    const dim3 grid((wp.ntiles + WT - 1) / WT, a->N / (32 * pl.nb), pl.splitk);
    // (one profiler id per INSTANTIATION -- 14: <2, false, 4>, 15: <4, false, 4>, 22: <2, true, 4>, 23: <2, false, 6>, 24: <1, false, 6>; 25, 26: wino_c32_kernel<false / true, false>; 32, 33: wino_c32_kernel<false / true, true> -- so that a record's name is
    // one kernel symbol of a rocprofv3 table)
    // (27, 28: wino_conv3_kernel<6 / 4>, the split-bf16 forms)
    const int prof = mtd_prof_begin(0, split3 ? (px == 6 ? 27 : 28) : px == 6 ? (pl.nb == 1 ? 24 : 23) : (pl.nb == 4 ? 15 : (pl.lean ? 22 : 14)), pl.splitk, p.M, a->N, a->C, 9, s, algorithmic_bytes(a));
    if (split3 && px == 6) MTD_LAUNCH((wino_conv3_kernel<6>), grid, dim3(512), 0, s, wp);
    else if (split3) MTD_LAUNCH((wino_conv3_kernel<4>), grid, dim3(512), 0, s, wp);
    else if (px == 6 && pl.nb == 1) MTD_LAUNCH((wino_conv_kernel<1, false, 6>), grid, dim3(512), 0, s, wp);
    else if (px == 6) MTD_LAUNCH((wino_conv_kernel<2, false, 6>), grid, dim3(512), 0, s, wp);
    else if (pl.nb == 4) MTD_LAUNCH((wino_conv_kernel<4>), grid, dim3(512), 0, s, wp);
    else if (pl.lean) MTD_LAUNCH((wino_conv_kernel<2, true>), grid, dim3(512), 0, s, wp);
    else MTD_LAUNCH((wino_conv_kernel<2>), grid, dim3(512), 0, s, wp);
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        const long long total = (long long)p.M * a->N;
        const bool vec = splitk_vec_ok(*a, p.M);
        int blocks = (int)(((vec ? total / 4 : total) + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (vec) hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(splitk_epilogue_scalar_kernel, dim3(blocks), dim3(256), 0, s, p);
        MTD_LAUNCH_CHECK();
    }
    return MTD_OK;
}

// ---- two or three convs of one shape in ONE launch (wino_conv_multi_kernel): a[0 .. count) as for mtd_conv_winograd, same geometry, N,
// C and weight form; inputs, weights, outputs, epilogue operands and workspaces their own.  The general fp32 kernels only (not the
// persistent 32-channel kernel, not the split-bf16 form): mtd_conv_winograd_group_ok says whether a group qualifies.
// The group's split of K is planned for the WHOLE grid (count problems: 1 / count of the slices per problem of a single launch -- fewer
// slabs, and another grouping of the K sum: a group equals single launches up to rounding, a few 1e-6).  Measured in the step for pairs
// against a plan per problem: 26.15 -> 25.80 ms (three A/B pairs; planning for 3 or 4 problems' worth of grid: +0.2 ms).
// Lab: MTD_WINO_PAIR_SPLIT = n plans every group as if it held n problems (1: per problem).
static int wino_group_sets(int count) {
    static const int v = [] { const char* e = mtd_lab_env("MTD_WINO_PAIR_SPLIT"); return (e && atoi(e) > 0) ? atoi(e) : 0; }();
    return v > 0 ? v : count;
}

extern "C" int mtd_conv_winograd_group_ok(const mtd_conv_args* a, int count) {
    if (!a || count < 2 || count > WINO_MULTI_MAX) return 0;
    const int px = wino_args_px(a[0]);
    const long long M = geom_pixels(a[0].g);
    for (int i = 0; i < count; ++i) {
        if (!mtd_conv_winograd_ok(&a[i])) return 0;
        if (__builtin_memcmp(&a[0].g, &a[i].g, sizeof(mtd_geom)) != 0 || a[0].N != a[i].N || a[0].C != a[i].C) return 0;
        if (wino_args_px(a[i]) != px || wino_c32_takes(a[i], px) || !aligned16(a[i].w)) return 0;
    }
    if ((px & 16) || (a[0].N % 64)) return 0;
    if ((px & 15) == 6 && (a[0].g.OW % 4)) return 0;
    const WinoPlan pl = wino_plan(a[0], px, wino_group_sets(count));
    if ((px & 15) == 6 && pl.nb != 2) return 0;
    // the problems' slab sums go through ONE launch of the 16-byte epilogue: all need it (else: single launches)
    if (pl.splitk > 1)
        for (int i = 0; i < count; ++i)
            if (!splitk_vec_ok(a[i], M)) return 0;
    return 1;
}

extern "C" int mtd_conv_winograd_group(const mtd_conv_args* a, int count, void* stream) {
    if (!mtd_conv_winograd_group_ok(a, count)) return MTD_EINVAL;
    const int pxcode = wino_args_px(a[0]);
    const int px = pxcode & 15;
    const WinoPlan pl = wino_plan(a[0], pxcode, wino_group_sets(count));
    WinoMulti mp;
    mp.count = count;
    double bytes = 0.0;
    for (int i = 0; i < WINO_MULTI_MAX; ++i) {
        const int rc = wino_fill(&a[i < count ? i : 0], pl, pxcode, mp.p[i]);
        if (rc != MTD_OK) return rc;
        mp.p[i].xcd_order = mp.p[0].xcd_order;
        if (i < count) bytes += algorithmic_bytes(&a[i]);
    }
    const IgemmParams& p = mp.p[0].p;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((mp.p[0].ntiles + WT - 1) / WT, a[0].N / (32 * pl.nb), count * pl.splitk);
    // (profiler ids 34-37: wino_conv_multi_kernel<2, false, 6>, <2, false, 4>, <4, false, 4>, <2, true, 4>)
    const int prof = mtd_prof_begin(0, px == 6 ? 34 : (pl.nb == 4 ? 36 : (pl.lean ? 37 : 35)), pl.splitk, (long long)count * p.M, a[0].N, a[0].C, 9, s, bytes);
    if (px == 6) MTD_LAUNCH((wino_conv_multi_kernel<2, false, 6>), grid, dim3(512), 0, s, mp);
    else if (pl.nb == 4) MTD_LAUNCH((wino_conv_multi_kernel<4>), grid, dim3(512), 0, s, mp);
    else if (pl.lean) MTD_LAUNCH((wino_conv_multi_kernel<2, true>), grid, dim3(512), 0, s, mp);
    else MTD_LAUNCH((wino_conv_multi_kernel<2>), grid, dim3(512), 0, s, mp);
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        IgemmMulti em;
        for (int i = 0; i < MULTI_MAX; ++i) em.p[i] = mp.p[i < count ? i : 0].p;      // (rows of the grid: blockIdx.y < count)
        const long long total = (long long)p.M * a[0].N;
        int blocks = (int)((total / 4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_epilogue_multi_kernel, dim3(blocks, count), dim3(256), 0, s, em);
        MTD_LAUNCH_CHECK();
    }
    return MTD_OK;
}

#include "conv_wino_s2.h"
