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
#define MTD_NO_API 1
#include "conv_igemm.hip"

namespace {

constexpr int WT = 32;        // tiles per workgroup
constexpr int WALD = 20;      // LDS row stride (floats) of As[xi][tile][16 c]: 16-byte aligned, b128 reads of 16 rows hit 16 x 4 distinct banks
constexpr int WXLD = 33;      // LDS row stride of the exchange image X[xi][tile][32 n]

struct WinoParams {
    IgemmParams p;            // args, M, split-K (c_per_split in channels), buffer extents, out_identity
    int ntiles, tiles_x, tiles_per_image;
    int nchunk;               // C / 8
};

// ---- weight transform:  Uw[xi][C/8][N][8] = (G g G^T)[xi],  g[a][b] = W(n, c, kmap[a * 3 + b])  (a, b = correlation position
// of the tap: input offset -1 + a, -1 + b).  One thread per (n, c).
struct WinoWDesc {
    const float* src; float* dst;
    long long sn, sc, st;     // W(n, c, kidx) = src[n * sn + c * sc + kidx * st]
    int N, C;
    int kmap[9];
    int pad_;
};

__global__ __launch_bounds__(256) void wino_weights_kernel(const WinoWDesc* __restrict__ tab, int count) {
    for (int d = blockIdx.y; d < count; d += gridDim.y) {
        const WinoWDesc w = tab[d];
        const long long total = (long long)w.N * w.C;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            // destination order: c8 fastest within (n), so consecutive threads write consecutive floats
            const int c8 = (int)(i & 7);
            const long long r = i >> 3;
            const int n = (int)(r % w.N);
            const int ck = (int)(r / w.N);
            const int c = ck * 8 + c8;
            const float* s = w.src + (long long)n * w.sn + (long long)c * w.sc;
            float g[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = s[(long long)w.kmap[a * 3 + b] * w.st];
            // t = G g  (4 x 3), u = t G^T (4 x 4);  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
            float t[4][3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                t[0][b] = g[0][b];
                t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
                t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
                t[3][b] = g[2][b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
                float* o = w.dst + (((long long)(a * 4) * (w.C / 8) + ck) * w.N + n) * 8 + c8;
                const long long xs = (long long)(w.C / 8) * w.N * 8;      // stride between positions xi
                o[0] = u0;
                o[xs] = u1;
                o[2 * xs] = u2;
                o[3 * xs] = u3;
            }
        }
    }
}

// ---- the convolution
template <int NB>
__global__ __launch_bounds__(512, 1) void wino_conv_kernel(const WinoParams wp) {
    // As: two buffers of 16 positions x 32 tiles x 16 channels (row stride WALD floats, 40 KB each); the exchange image of the
    // epilogue (16 x 32 x WXLD floats = 66 KB) reuses the same memory after the K loop.
    constexpr int AS_BUF = 16 * WT * WALD;
    constexpr int X_SIZE = 16 * WT * WXLD;
    __shared__ __attribute__((aligned(16))) float Ls[(2 * AS_BUF > X_SIZE) ? 2 * AS_BUF : X_SIZE];
    const IgemmParams& p = wp.p;
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int tile0 = blockIdx.x * WT;
    const int n0 = blockIdx.y * (32 * NB);
    const int zk = blockIdx.z;
    // K runs in steps of 16 channels (one transform per thread and step), each step two MFMA sub-chunks of 8
    const int st_beg = zk * (p.c_per_split >> 4);
    const int st_end = min(wp.nchunk >> 1, st_beg + (p.c_per_split >> 4));
    const int nst = st_end - st_beg;
    const int st_last = st_end - 1;

    // ---- transform role: thread (tile tt, channel tc of the step).  The 16 patch pixels are ONE lane offset (out of range
    // for a tile past the end) + a scalar displacement per pixel, with a 16-bit validity mask for the image border.
    const int tt = tid >> 4, tc = tid & 15;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    unsigned pbase, pvalid = 0;
    {
        const int tg = tile0 + tt;
        const bool tv = tg < wp.ntiles;
        const int b = tg / wp.tiles_per_image;
        const int r = tg - b * wp.tiles_per_image;
        const int ty = r / wp.tiles_x, tx = r - ty * wp.tiles_x;
        // offset of patch pixel (0, 0) = image pixel (2 ty - 1, 2 tx - 1), which may lie outside: formed modulo 2^32, every
        // VALID pixel's offset pbase + displacement is inside the buffer
        pbase = (unsigned)(((((long long)b * g.IH + (2 * ty - 1)) * g.IW + (2 * tx - 1)) * a.in_ld + tc) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = 2 * ty - 1 + i, ix = 2 * tx - 1 + j;
                if (tv & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) pvalid |= 1u << (i * 4 + j);
            }
    }
    const int row_b = g.IW * a.in_ld * 4, px_b = a.in_ld * 4;      // byte displacements of one image row / one pixel
    float d[16];
    auto load_patch = [&](int st) {        // step st (absolute, clamped by the caller): channels 16 st + tc
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned vo = ((pvalid >> (i * 4 + j)) & 1u) ? pbase + (unsigned)(i * row_b + j * px_b) : 0x80000000u;
                d[i * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, vo, st * 64, 0));
            }
    };
    // B^T d B -> As[xi][tt][tc], in three pieces that the K loop places between groups of MFMAs
    float t[16];
    auto transform_cols = [&]() {             // columns: B^T d
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d0 = d[j], d1 = d[4 + j], d2 = d[8 + j], d3 = d[12 + j];
            t[j] = d0 - d2;
            t[4 + j] = d1 + d2;
            t[8 + j] = d2 - d1;
            t[12 + j] = d1 - d3;
        }
    };
    auto transform_rows = [&](float* As, int i0) {   // rows i0, i0 + 1: (.) B, and the stores
        float* o = As + tt * WALD + tc;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            const float t0 = t[4 * i], t1 = t[4 * i + 1], t2 = t[4 * i + 2], t3 = t[4 * i + 3];
            o[(4 * i + 0) * (WT * WALD)] = t0 - t2;
            o[(4 * i + 1) * (WT * WALD)] = t1 + t2;
            o[(4 * i + 2) * (WT * WALD)] = t2 - t1;
            o[(4 * i + 3) * (WT * WALD)] = t1 - t3;
        }
    };
    auto transform_store = [&](float* As) {
        transform_cols();
        transform_rows(As, 0);
        transform_rows(As, 2);
    };

    // ---- MFMA role: positions 2 wave, 2 wave + 1; B fragments straight from the transformed weights
    const float* wbase = a.w + ((long long)(2 * wave) * wp.nchunk * a.N + (n0 + l31)) * 8 + kh * 4;
    const long long xi_stride = (long long)wp.nchunk * a.N * 8;
    auto load_b = [&](int ck, f32x4 (&bf)[2][NB]) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[x][nb] = *reinterpret_cast<const f32x4*>(wbase + x * xi_stride + ((long long)ck * a.N + nb * 32) * 8);
    };
    f32x16 acc[2][NB];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][nb][e] = 0.f;
    f32x4 af[2][2];                                              // [sub-chunk][position]
    auto load_af = [&](const float* Ac) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int x = 0; x < 2; ++x) af[u][x] = *reinterpret_cast<const f32x4*>(Ac + ((2 * wave + x) * WT + l31) * WALD + u * 8 + kh * 4);
    };
    auto mfma_group = [&](int u, int s, const f32x4 (&bf)[2][NB]) {      // k-step s of sub-chunk u: 2 NB MFMAs
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[x][nb] = mfma32(af[u][x][s], bf[x][nb][s], acc[x][nb]);
    };

    // ---- prologue: step 0 transformed into buffer 0, step 1's patch in flight, step 0's first weights in registers.
    // Every load below is issued unconditionally (indices clamped to the slice's last step): the loop then has ONE path, and
    // the waits the compiler places count exactly the loads that are younger than the registers an MFMA needs.  (With the
    // loads under `if (k + 1 < n)` the path that skips them set the wait counts for all, and every chunk stood for a full
    // memory round trip of its own prefetches: half speed.)
    f32x4 b0[2][NB], b1[2][NB];
    if (nst > 0) {
        load_patch(st_beg);
        load_b(2 * st_beg, b0);
        transform_store(Ls);
        load_patch(min(st_beg + 1, st_last));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): nothing of the prologue is pending inside the loop
    __syncthreads();
    // One K step.  Issue order of the loads (vmcnt counts in this order): b1 (second half of this step), the patch of step
    // j + 2, b0 (first half of step j + 1) -- each is needed a whole group of MFMAs after it was requested.  The transform of
    // step j + 1 (vector ALU + LDS stores) sits in three pieces between the MFMA groups of the first half; the scheduling
    // fences keep the compiler from sinking the loads next to their uses (it did: every chunk then waited for its own loads).
#pragma unroll 1
    for (int j = 0; j < nst; ++j) {
        const int st = st_beg + j;
        const float* Ac = Ls + (j & 1) * AS_BUF;
        float* An = Ls + ((j + 1) & 1) * AS_BUF;
        load_af(Ac);
        load_b(2 * st + 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, 0, b0);
        __builtin_amdgcn_sched_barrier(0);
        transform_cols();
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, 1, b0);
        __builtin_amdgcn_sched_barrier(0);
        transform_rows(An, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, 2, b0);
        __builtin_amdgcn_sched_barrier(0);
        transform_rows(An, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(0, 3, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_patch(min(st + 2, st_last));
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, 0, b1);
        mfma_group(1, 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        load_b(2 * min(st + 1, st_last), b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, 2, b1);
        mfma_group(1, 3, b1);
        __syncthreads();
    }

    // ---- epilogue: one n block at a time through the exchange image X[xi][tile][n].  A thread finishes channel n = tid & 31
    // of tiles (tid >> 5) and (tid >> 5) + 16: A^T m A, then the conv epilogue on its 2 x 4 output pixels -- operands first
    // (uniform tests around whole batches of loads, as in conv_igemm.hip's epi_group), arithmetic in epilogue_value()'s order.
    const ScalePair sp = load_scale(a);
    const int en = tid & 31;
    long long pix0[2];                                            // top-left output pixel of the thread's two tiles (-1: none)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tg = tile0 + (tid >> 5) + 16 * h;
        const int bimg = tg / wp.tiles_per_image;
        const int r = tg - bimg * wp.tiles_per_image;
        const int ty = r / wp.tiles_x, tx = r - ty * wp.tiles_x;
        pix0[h] = tg < wp.ntiles ? ((long long)bimg * g.OH + 2 * ty) * g.OW + 2 * tx : -1;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {        // (unrolled: a run-time index into the accumulators would put them in scratch memory)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float* X = Ls + (2 * wave + x) * (WT * WXLD) + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) X[mfma32_row(e, lane) * WXLD] = acc[x][nb][e];
        }
        __syncthreads();
        const int n = n0 + nb * 32 + en;
        float y[8];
        long long pp[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tl = (tid >> 5) + 16 * h;
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = Ls[(xi * WT + tl) * WXLD + en];
            float t2[2][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                t2[0][b] = m[b] + m[4 + b] + m[8 + b];
                t2[1][b] = m[4 + b] - m[8 + b] - m[12 + b];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                y[h * 4 + 2 * i] = t2[i][0] + t2[i][1] + t2[i][2];
                y[h * 4 + 2 * i + 1] = t2[i][1] - t2[i][2] - t2[i][3];
                pp[h * 4 + 2 * i] = pix0[h] + (long long)i * g.OW;
                pp[h * 4 + 2 * i + 1] = pix0[h] + (long long)i * g.OW + 1;
            }
        }
        if (p.splitk > 1) {
            float* slab = a.ws + (long long)zk * ((long long)p.M * a.N) + n;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (pix0[q >> 2] >= 0) slab[pp[q] * a.N] = y[q];
        } else {
            const float bias_n = a.bias ? a.bias[n] : 0.f;
            float e1[8], e2[8], em[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { e1[q] = -0.0f; e2[q] = -0.0f; em[q] = 1.f; }
            if (a.add1) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (pix0[q >> 2] >= 0) e1[q] = a.add1[pp[q] * a.add1_ld + n];
            }
            if (a.add2) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (pix0[q >> 2] >= 0) e2[q] = a.add2[pp[q] * a.add2_ld + n];
            }
            if (a.mask) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (pix0[q >> 2] >= 0) em[q] = a.mask[pp[q] * a.mask_ld + n];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float v = y[q] * pick_scale(sp, (int)pp[q]) + bias_n;
                v += e1[q];
                v += e2[q];
                y[q] = v;
            }
            if (a.act == MTD_ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = y[q] > 0.f ? y[q] : 0.f;
            } else if (a.act == MTD_ACT_LRELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = y[q] > 0.f ? y[q] : 0.2f * y[q];
            }
            if (a.mask) {
                const float slope = a.mask_slope;
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] *= (em[q] > 0.f) ? 1.f : slope;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (pix0[q >> 2] >= 0) a.out[pp[q] * a.out_ld + n] = y[q];
        }
        __syncthreads();
    }
}

// the kernel's domain: 3x3, stride 1, "same" size, even height and width, every tap within one pixel of the output position,
// output pixel == launch pixel, C a multiple of 16, N a multiple of 64, the input view inside 32-bit byte offsets
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
    if ((a.C % 16) || (a.N % 64) || a.out2) return false;
    return true;
}

struct WinoPlan { int nb, splitk, c_per_split; };

WinoPlan wino_plan(const mtd_conv_args& a) {
    WinoPlan pl{};
    pl.nb = (a.N % 128 == 0) ? 4 : 2;
    const long long tiles = geom_pixels(a.g) / 4;
    long long blocks = ((tiles + WT - 1) / WT) * (a.N / (32 * pl.nb));
    if (blocks < 192 && pl.nb == 4 && a.N % 64 == 0) {          // more, narrower workgroups before splitting K
        pl.nb = 2;
        blocks = ((tiles + WT - 1) / WT) * (a.N / 64);
    }
    const int chunks = a.C / 16;                                 // K steps of 16 channels
    int sk = blocks <= 128 ? (int)(256 / blocks) : 1;
    if (sk > chunks / 4) sk = chunks / 4;                        // at least four steps per slice
    if (sk > 16) sk = 16;
    if (sk < 1) sk = 1;
    static const int env_sk = [] { const char* e = getenv("MTD_WINO_SPLITK"); return e ? atoi(e) : 0; }();
    if (env_sk > 0) sk = env_sk < chunks ? env_sk : chunks;
    const int cps = (chunks + sk - 1) / sk;
    pl.splitk = (chunks + cps - 1) / cps;
    pl.c_per_split = cps * 16;
    return pl;
}

}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------
// Transformed weights of `count` conv views in one launch.  desc[i]: the weight view W(n, c, kidx) = src[n sn + c sc + kidx st]
// and the geometry it will be used with (its taps decide which filter entry sits at which correlation position: a forward
// conv and the data gradient of the same layer need different transforms); dst: 16 * N * C floats, layout [xi][C/8][N][8].
extern "C" size_t mtd_winograd_weight_floats(int N, int C) { return (N > 0 && C > 0) ? (size_t)16 * N * C : 0; }

extern "C" int mtd_winograd_weights(const mtd_wino_weight_desc* table_dev, const mtd_wino_weight_desc* table_host, int count, void* stream) {
    static_assert(sizeof(mtd_wino_weight_desc) == sizeof(WinoWDesc), "descriptor layouts must agree");
    if (!table_dev || !table_host || count <= 0) return MTD_EINVAL;
    long long most = 0;
    for (int i = 0; i < count; ++i) {
        const mtd_wino_weight_desc& d = table_host[i];
        if (!d.src || !d.dst || d.N <= 0 || d.C <= 0 || (d.C % 8)) return MTD_EINVAL;
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
    return 1;
}

extern "C" size_t mtd_conv_winograd_ws_bytes(const mtd_conv_args* a) {
    if (!mtd_conv_winograd_ok(a)) return 0;
    const WinoPlan pl = wino_plan(*a);
    return pl.splitk > 1 ? (size_t)pl.splitk * (size_t)geom_pixels(a->g) * a->N * sizeof(float) : 0;
}

// a: as for mtd_conv_igemm, except that a->w points to the TRANSFORMED weights of this view and geometry
// (mtd_winograd_weights; a->w_sn / w_sc / w_st are ignored).  Same epilogue, same split-K workspace contract.
extern "C" int mtd_conv_winograd(const mtd_conv_args* a, void* stream) {
    if (!mtd_conv_winograd_ok(a)) return MTD_EINVAL;
    if (!aligned16(a->w)) return MTD_EALIGN;
    const WinoPlan pl = wino_plan(*a);
    WinoParams wp;
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
    p.wide = 0;
    wp.tiles_x = a->g.OW / 2;
    wp.tiles_per_image = (a->g.OH / 2) * wp.tiles_x;
    wp.ntiles = a->g.B * wp.tiles_per_image;
    wp.nchunk = a->C / 8;
    if (pl.splitk > 1) {
        const size_t need = (size_t)pl.splitk * (size_t)p.M * a->N * sizeof(float);
        if (!a->ws || a->ws_bytes < need) return MTD_EWS;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((wp.ntiles + WT - 1) / WT, a->N / (32 * pl.nb), pl.splitk);
    const int prof = mtd_prof_begin(0, 14, pl.splitk, p.M, a->N, a->C, 9, s, algorithmic_bytes(a));
    if (pl.nb == 4) MTD_LAUNCH((wino_conv_kernel<4>), grid, dim3(512), 0, s, wp);
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
