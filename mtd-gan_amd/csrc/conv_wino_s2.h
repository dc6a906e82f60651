// Winograd F(3x3, 2x2) on fp32 MFMA for the 4x4 / stride-2 / padding-1 layers of the discriminator's encoder
// (arch/Ours/networks.py:185-215 down1..3: Conv2d(k4, s2, p1)) -- the forward conv and the four-parity data gradient.
// (Included by conv_winograd.hip: same tile machinery as wino_conv_kernel -- 32 tiles x 64 channels per workgroup, the four
// rows of a 4 x 4 patch in the four lanes of a quad, weights straight from L2, the accumulators met in an LDS exchange image.)
//
// Polyphase view of the FORWARD conv: with the input padded by one pixel, input pixel 2 o + k (k = 0..3) of output o is pixel
// (o + j) of phase p of the padded input, k = 2 j + p: the 4 x 4 stride-2 conv over C channels IS a 2 x 2 stride-1 "valid" conv
// over the 4 C channels (py, px, c) of the space-to-depth image of the padded input.  F(3x3, 2x2) then takes 16 multiplications
// per 3 x 3 output tile and channel pair instead of 36: 2.25x fewer MFMA flops than the implicit GEMM over 16 taps (less the
// ragged last tile of a 32- / 16- / 8-pixel row: 33 / 18 / 9 pixels computed).  The space-to-depth image is never formed: a
// K step's channels (one phase, 16 channels) are read from the NHWC input at pixel stride 2 from the phase's own origin.
// DATA GRADIENT: each of the four input-parity classes (geom_dgrad_s2) already is a 2 x 2 stride-1 conv over the cotangent
// (one "phase", pixel stride 1) whose result lands on every other pixel of the input gradient; up to four classes run as ONE
// grid (sets).
//   patch row r (0..3) of tile row ty:  input row  ps (3 ty + r) + base_y + py,   ps = 2 / 1,  base = off (forward), off - 1 (class:
//   tap_d = -1, correlation index j = 1 - tap);   interpolation points 0, +-1, inf as for F(2,3): the SAME input transform B^T,
//   G = [1 0; .5 .5; .5 -.5; 0 1],   A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 -1].   fp32 error like F(2x2, 3x3)'s (3e-7 .. 6e-7 of max-abs).
namespace {

struct W32Set { const float* w; float* ws; int base_y, base_x, out_oy, out_ox; };

struct W32Params {
    IgemmParams p;            // set 0's args; M = launch pixels of ONE set; split-K (c_per_split in channels of the 4 C / C sum)
    int ntiles, tiles_x, tiles_y;
    int nchunk;               // (groups C) / 8
    int groups, gsteps, ps;   // phases in the K sum (4 / 1), K steps of 16 channels per phase, input pixel stride
    int xcd_order, nsets;
    unsigned w_bytes;
    W32Set set[4];
};

struct Wino32WDesc {
    const float* src; float* dst;
    long long sn, sc, st;     // W(n, c, kidx) = src[n sn + c sc + kidx st]
    int N, C, groups;
    int kmap[16];             // [group][jy][jx] -> kidx of the filter entry at correlation position (jy, jx) of that phase
};

// Uw[xi = 4 a + b][(groups C) / 8][N][8] = (G g G^T)[a][b],  g[jy][jx] = W(n, c, kmap[grp][jy][jx]).  One thread per (n, k).
__global__ __launch_bounds__(256) void wino32_weights_kernel(const Wino32WDesc* __restrict__ tab, int count) {
    for (int d = blockIdx.y; d < count; d += gridDim.y) {
        const Wino32WDesc w = tab[d];
        const int K = w.groups * w.C;
        const long long total = (long long)w.N * K;
        const long long xs = (long long)(K / 8) * w.N * 8;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int c8 = (int)(i & 7);
            const long long r = i >> 3;
            const int n = (int)(r % w.N);
            const int ck = (int)(r / w.N);
            const int k = ck * 8 + c8;
            const int grp = k / w.C, c = k - grp * w.C;
            const float* s = w.src + (long long)n * w.sn + (long long)c * w.sc;
            float g[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) g[a][b] = s[(long long)w.kmap[grp * 4 + a * 2 + b] * w.st];
            float t[4][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                t[0][b] = g[0][b];
                t[1][b] = 0.5f * (g[0][b] + g[1][b]);
                t[2][b] = 0.5f * (g[0][b] - g[1][b]);
                t[3][b] = g[1][b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float* o = w.dst + (((long long)(a * 4) * (K / 8) + ck) * w.N + n) * 8 + c8;
                o[0] = t[a][0];
                o[xs] = 0.5f * (t[a][0] + t[a][1]);
                o[2 * xs] = 0.5f * (t[a][0] - t[a][1]);
                o[3 * xs] = t[a][1];
            }
        }
    }
}

// LEAN: at most 128 registers (one weight-fragment set), two workgroups per CU -- one's prologue / exchange epilogue under the
// other's MFMAs, for the launches with few K steps (the data-gradient classes: K = the layer's output channels)
template <int NB, bool LEAN = false>
__global__ __launch_bounds__(512, LEAN ? 4 : 1) void wino32_conv_kernel(const W32Params wp) {
    constexpr int PX = 4, NP = 16, PW = 2;
    constexpr int AS_BUF = NP * WT * WALD;
    constexpr int X_SIZE = NP * WT * WXLD;
    __shared__ __attribute__((aligned(16))) float Ls[(2 * AS_BUF > X_SIZE) ? 2 * AS_BUF : X_SIZE];
    const IgemmParams& p = wp.p;
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (wp.xcd_order) {          // as wino_conv_kernel: 1 = weights dominate (dispatch order), 2 = input dominates (tile block slowest)
        const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
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
    const int si = bz / p.splitk;                  // set (parity class)
    const int zk = bz - si * p.splitk;             // K slice
    // (field by field: a run-time index into the by-value kernel argument may be lowered through a scratch copy)
    const W32Set S = si == 0 ? wp.set[0] : (si == 1 ? wp.set[1] : (si == 2 ? wp.set[2] : wp.set[3]));
    const int nsteps = wp.groups * wp.gsteps;
    const int st_beg = zk * (p.c_per_split >> 4);
    const int st_end = min(nsteps, st_beg + (p.c_per_split >> 4));
    const int nst = st_end - st_beg;
    const int st_last = st_end - 1;

    // ---- transform role: thread (tile tt, channel quad tq, patch row ti)
    const int ti = tid & 3, tq = (tid >> 2) & 3, tt = tid >> 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    unsigned pbase;
    unsigned pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0;      // validity of the row's four pixels for phase (py, px) = 2 py + px
    {
        const int tg = tile0 + tt;
        const bool tv = tg < wp.ntiles;
        int b, ty, tx;
        pix_decompose(tg, wp.tiles_x, wp.tiles_y, b, ty, tx);
        const int row0 = wp.ps * (3 * ty + ti) + S.base_y;
        const int col0 = wp.ps * (3 * tx) + S.base_x;
        // (formed modulo 2^32; every VALID pixel's offset, phase displacement included, is in range)
        pbase = (unsigned)(((((long long)b * g.IH + row0) * g.IW + col0) * a.in_ld + 4 * tq) * 4);
        unsigned cm0 = 0, cm1 = 0;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            if ((unsigned)(col0 + wp.ps * j) < (unsigned)g.IW) cm0 |= 1u << j;
            if ((unsigned)(col0 + wp.ps * j + 1) < (unsigned)g.IW) cm1 |= 1u << j;
        }
        const bool r0 = tv & ((unsigned)row0 < (unsigned)g.IH), r1 = tv & ((unsigned)(row0 + 1) < (unsigned)g.IH);
        pv0 = r0 ? cm0 : 0u;
        pv1 = r0 ? cm1 : 0u;
        pv2 = r1 ? cm0 : 0u;
        pv3 = r1 ? cm1 : 0u;
    }
    const int px_b = wp.ps * a.in_ld * 4;          // byte displacement of one patch pixel
    const int row_b = g.IW * a.in_ld * 4;          // ... of phase py = 1
    const int col_b = a.in_ld * 4;                 // ... of phase px = 1
    f32x4 d[PX];
    auto load_patch = [&](int st) {                // step st of the K sum: phase grp = st / gsteps, channels 16 (st % gsteps) + 4 tq .. + 3
        const int grp = st / wp.gsteps;            // (wave-uniform: scalar unit)
        const int sg = st - grp * wp.gsteps;
        const unsigned pvalid = grp == 0 ? pv0 : (grp == 1 ? pv1 : (grp == 2 ? pv2 : pv3));
        const unsigned pb = pbase + (unsigned)((grp >> 1) * row_b + (grp & 1) * col_b);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const unsigned vo = ((pvalid >> j) & 1u) ? pb + (unsigned)(j * px_b) : 0x80000000u;
            d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, sg * 64, 0));
        }
    };
    const float qsign = ti == 1 ? 1.f : -1.f;
    auto row_value = [&](int j) -> f32x4 {
        return j == 0 ? d[0] - d[2] : (j == 1 ? d[1] + d[2] : (j == 2 ? d[2] - d[1] : d[1] - d[3]));
    };
    auto transform_store = [&](float* As) {
        float* o = As + tt * WALD + 4 * tq;
#pragma unroll
        for (int j = 0; j < PX; ++j) *reinterpret_cast<f32x4*>(o + (PX * ti + j) * (WT * WALD)) = wino_quad_rows(row_value(j), qsign);
    };

    // ---- MFMA role: positions 2 wave, 2 wave + 1
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(S.w), (short)0, (int)wp.w_bytes, 0x00020000);
    const unsigned w_lane = (unsigned)(((n0 + l31) * 8 + kh * 4) * 4);
    const int xi_stride_b = wp.nchunk * a.N * 32;
    const int w_pos0 = PW * wave * xi_stride_b;
    auto load_b = [&](int ck, f32x4 (&bf)[PW][NB]) {
#pragma unroll
        for (int x = 0; x < PW; ++x)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bf[x][nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, w_lane + (unsigned)(nb * 1024), w_pos0 + x * xi_stride_b + ck * a.N * 32, 0));
    };
    f32x16 acc[PW][NB];
#pragma unroll
    for (int x = 0; x < PW; ++x)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][nb][e] = 0.f;
    f32x4 af[PW];
    auto load_af = [&](const float* Ac, int u) {
#pragma unroll
        for (int x = 0; x < PW; ++x) af[x] = *reinterpret_cast<const f32x4*>(Ac + ((PW * wave + x) * WT + l31) * WALD + u * 8 + kh * 4);
    };
    auto mfma_group = [&](int s, const f32x4 (&bf)[PW][NB]) {
#pragma unroll
        for (int x = 0; x < PW; ++x)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[x][nb] = mfma32(bf[x][nb][s], af[x][s], acc[x][nb]);      // (transposed blocks: rows = channels)
    };

    f32x4 b0[PW][NB], b1[PW][NB];
    if (nst > 0) {
        load_patch(st_beg);
        load_b(2 * st_beg, b0);
        transform_store(Ls);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    // the K step of wino_conv_kernel<2, false, 4>: loads spread over the MFMAs, the next step's transform between the second half's
#pragma unroll 1
    for (int j = 0; j < nst; ++j) {
        const int st = st_beg + j;
        const float* Ac = Ls + (j & 1) * AS_BUF;
        float* An = Ls + ((j + 1) & 1) * AS_BUF;
        load_af(Ac, 0);
        load_patch(min(st + 1, st_last));
        if constexpr (LEAN) {
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, b0); mfma_group(1, b0); mfma_group(2, b0); mfma_group(3, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(2 * st + 1, b0);
            load_af(Ac, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, b0); mfma_group(1, b0); mfma_group(2, b0); mfma_group(3, b0);
            transform_store(An);
#pragma unroll
            for (int i = 0; i < 8 * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_b(2 * min(st + 1, st_last), b0);
        } else {
            load_b(2 * st + 1, b1);
            mfma_group(0, b0); mfma_group(1, b0); mfma_group(2, b0); mfma_group(3, b0);
            __builtin_amdgcn_sched_group_barrier(0x100, PW, 0);
#pragma unroll
            for (int i = 0; i < PX + PW * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * PW * NB - 2 * (PX + PW * NB), 0);
            __builtin_amdgcn_sched_barrier(0);
            load_af(Ac, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_b(2 * min(st + 1, st_last), b0);
            mfma_group(0, b1); mfma_group(1, b1); mfma_group(2, b1); mfma_group(3, b1);
            transform_store(An);
#pragma unroll
            for (int i = 0; i < 8 * NB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NB == 4 ? 3 : 6, 0);
                if ((i & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if ((i & (NB == 4 ? 7 : 3)) == (NB == 4 ? 7 : 3)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    // ---- epilogue: thread (tile etl, channel quad enq, ei): ei = 0 finishes output rows 0 and 2 of the 3 x 3 tile, ei = 1 row 1
    const ScalePair sp = load_scale(a);
    const int ei = tid & 1, enq = (tid >> 1) & 7, etl = tid >> 4;
    int eb, ety, etx;
    const int etg = tile0 + etl;
    pix_decompose(etg, wp.tiles_x, wp.tiles_y, eb, ety, etx);
    const bool etv = etg < wp.ntiles;
    const bool vec = (p.wide & 1) != 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + nb * 32 + 4 * enq;
#pragma unroll
        for (int x = 0; x < PW; ++x) {
            float* X = Ls + ((PW * wave + x) * WT + l31) * WXLD + 4 * kh;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<f32x4*>(X + 8 * gq) = f32x4{acc[x][nb][4 * gq], acc[x][nb][4 * gq + 1], acc[x][nb][4 * gq + 2], acc[x][nb][4 * gq + 3]};
        }
        __syncthreads();
        // the epilogue operands of the thread's (up to) six pixels are requested in ONE batch before the exchange image is read
        // (requested pixel by pixel behind the stores, every pixel stood for a memory round trip of its own: the output may
        // alias them as far as the compiler knows)
        constexpr int RB = LEAN ? 1 : 2;          // rows per batch (LEAN: 128 registers)
        int opix[2][3];          // (inside 31 bits with the channel stride applied: wino32_eligible)
        bool pok[2][3];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int oy = 3 * ety + (rr == 0 ? ei : 2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ox = 3 * etx + q;
                pok[rr][q] = etv && (rr == 0 || ei == 0) && oy < g.OH && ox < g.OW;
                opix[rr][q] = (eb * g.OHF + oy * g.out_sy + S.out_oy) * g.OWF + ox * g.out_sx + S.out_ox;
            }
        }
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 e1[RB][3], e2[RB][3], em[RB][3];
        auto load_ops = [&](int rr, int slot) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                e1[slot][q] = f32x4{-0.0f, -0.0f, -0.0f, -0.0f};
                e2[slot][q] = e1[slot][q];
                em[slot][q] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (p.splitk > 1 || !pok[rr][q]) continue;
                const long long op = opix[rr][q];
                if (vec) {
                    if (a.add1) e1[slot][q] = *reinterpret_cast<const f32x4*>(a.add1 + op * a.add1_ld + n);
                    if (!LEAN && a.add2) e2[slot][q] = *reinterpret_cast<const f32x4*>(a.add2 + op * a.add2_ld + n);
                    if (a.mask) em[slot][q] = *reinterpret_cast<const f32x4*>(a.mask + op * a.mask_ld + n);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (a.add1) e1[slot][q][c] = a.add1[op * a.add1_ld + n + c];
                        if (!LEAN && a.add2) e2[slot][q][c] = a.add2[op * a.add2_ld + n + c];
                        if (a.mask) em[slot][q][c] = a.mask[op * a.mask_ld + n + c];
                    }
                }
            }
        };
        if (p.splitk == 1 && a.bias) {
#pragma unroll
            for (int c = 0; c < 4; ++c) bias4[c] = a.bias[n + c];
        }
        if constexpr (!LEAN) { load_ops(0, 0); load_ops(1, 1); }
        f32x4 ta[PX], tb[PX];      // rows ei (ta) and, for ei = 0, 2 (tb) of A^T m, per patch column
#pragma unroll
        for (int b = 0; b < PX; ++b) {
            const float* col = Ls + (b * WT + etl) * WXLD + 4 * enq;              // position xi = 4 a + b at col + a * 4 * WT * WXLD
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(col + 1 * PX * WT * WXLD);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(col + 2 * PX * WT * WXLD);
            if (ei) {
                ta[b] = m1 - m2;
                tb[b] = ta[b];
            } else {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(col);
                const f32x4 m3 = *reinterpret_cast<const f32x4*>(col + 3 * PX * WT * WXLD);
                const f32x4 s12 = m1 + m2;
                ta[b] = m0 + s12;
                tb[b] = s12 - m3;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int slot = LEAN ? 0 : rr;
            if constexpr (LEAN) {            // (one row's operands at a time, after the exchange image has been read: 128 registers)
                __builtin_amdgcn_sched_barrier(0);
                load_ops(rr, 0);
            }
            const int oy = 3 * ety + (rr == 0 ? ei : 2);
            f32x4 y[3];
            if (rr == 0) {
                y[0] = ta[0] + ta[1] + ta[2];
                y[1] = ta[1] - ta[2];
                y[2] = ta[1] + ta[2] - ta[3];
            } else {
                y[0] = tb[0] + tb[1] + tb[2];
                y[1] = tb[1] - tb[2];
                y[2] = tb[1] + tb[2] - tb[3];
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (!pok[rr][q]) continue;
                const int m = (eb * g.OH + oy) * g.OW + 3 * etx + q;              // launch-grid pixel
                if (p.splitk > 1) {
                    float* slab = S.ws + (long long)zk * ((long long)p.M * a.N) + (long long)m * a.N + n;
                    if (vec && (p.wide & 2)) *reinterpret_cast<f32x4*>(slab) = y[q];
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c) slab[c] = y[q][c];
                    continue;
                }
                const float sc = pick_scale(sp, m);
                f32x4 v = y[q];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float u = v[c] * sc + bias4[c];
                    u += e1[slot][q][c];
                    if constexpr (LEAN) { if (a.add2) u += a.add2[(long long)opix[rr][q] * a.add2_ld + n + c]; }      // (a rare second operand: at its use)
                    else u += e2[slot][q][c];
                    if (a.act == MTD_ACT_RELU) u = u > 0.f ? u : 0.f;
                    else if (a.act == MTD_ACT_LRELU) u = u > 0.f ? u : 0.2f * u;
                    if (a.mask) u *= (em[slot][q][c] > 0.f) ? 1.f : a.mask_slope;
                    v[c] = u;
                }
                float* o = a.out + (long long)opix[rr][q] * a.out_ld + n;
                if (vec) *reinterpret_cast<f32x4*>(o) = v;
                else
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = v[c];
            }
        }
        __syncthreads();
    }
}

// ---- host side
struct W32Form { int groups, ps, base_y, base_x; int kmap[16]; };

// Which form is this geometry?  forward: 4x4 taps at stride 2 (tap = 2 j + phase); class: 2x2 taps at stride 1
bool wino32_form(const mtd_geom& g, W32Form& f) {
    if (g.TH != g.TW || g.in_sy != g.in_sx || g.tap_dy != g.tap_dx) return false;
    if (g.TH == 4 && g.in_sy == 2 && g.tap_dy == 1) {
        f.groups = 4; f.ps = 2; f.base_y = g.off_y; f.base_x = g.off_x;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int jy = 0; jy < 2; ++jy)
                    for (int jx = 0; jx < 2; ++jx)
                        f.kmap[(py * 2 + px) * 4 + jy * 2 + jx] = (g.ky0 + (2 * jy + py) * g.ky_step) * g.KW + (g.kx0 + (2 * jx + px) * g.kx_step);
        return true;
    }
    if (g.TH == 2 && g.in_sy == 1 && (g.tap_dy == 1 || g.tap_dy == -1)) {
        const bool rev = g.tap_dy < 0;
        f.groups = 1; f.ps = 1; f.base_y = g.off_y - (rev ? 1 : 0); f.base_x = g.off_x - (rev ? 1 : 0);
        for (int i = 0; i < 16; ++i) f.kmap[i] = 0;
        for (int jy = 0; jy < 2; ++jy)
            for (int jx = 0; jx < 2; ++jx) {
                const int ty = rev ? 1 - jy : jy, tx = rev ? 1 - jx : jx;
                f.kmap[jy * 2 + jx] = (g.ky0 + ty * g.ky_step) * g.KW + (g.kx0 + tx * g.kx_step);
            }
        return true;
    }
    return false;
}

bool wino32_eligible(const mtd_conv_args* a, int count) {
    if (!a || count < 1 || count > 4) return false;
    W32Form f0;
    if (!wino32_form(a[0].g, f0)) return false;
    for (int i = 0; i < count; ++i) {
        const mtd_conv_args& s = a[i];
        if (!s.in || !s.w || !s.out || s.C <= 0 || s.N <= 0 || s.in_ld < s.C || s.out_ld < s.N) return false;
        if ((s.C % 16) || (s.N % 64) || s.out2 || s.act == MTD_ACT_RELU_ADD || (s.in_ld % 4) || !aligned16(s.in) || !aligned16(s.w)) return false;
        W32Form f;
        if (!wino32_form(s.g, f) || f.groups != f0.groups) return false;
        const mtd_geom &g = s.g, &h = a[0].g;
        if (g.out_sy < 1 || g.out_sx < 1) return false;
        if ((g.OH - 1) * g.out_sy + g.out_oy >= g.OHF || (g.OW - 1) * g.out_sx + g.out_ox >= g.OWF) return false;
        if (i) {      // one shape, one set of operands: the classes differ in offsets, filter entries and where their pixels land
            if (g.B != h.B || g.IH != h.IH || g.IW != h.IW || g.OH != h.OH || g.OW != h.OW || g.OHF != h.OHF || g.OWF != h.OWF ||
                g.out_sy != h.out_sy || g.out_sx != h.out_sx) return false;
            if (s.in != a[0].in || s.in_ld != a[0].in_ld || s.C != a[0].C || s.N != a[0].N || s.out != a[0].out || s.out_ld != a[0].out_ld ||
                s.scale != a[0].scale || s.scale2 != a[0].scale2 || s.scale_split != a[0].scale_split || s.bias != a[0].bias ||
                s.add1 != a[0].add1 || s.add1_ld != a[0].add1_ld || s.add2 != a[0].add2 || s.add2_ld != a[0].add2_ld || s.act != a[0].act ||
                s.mask != a[0].mask || s.mask_ld != a[0].mask_ld || s.mask_slope != a[0].mask_slope) return false;
        }
        const long long npix = (long long)g.B * g.IH * g.IW;
        if (((npix - 1) * s.in_ld + s.C) * 4 >= (1ll << 31)) return false;
        if (geom_pixels(g) * s.N >= (1ll << 31)) return false;
        if ((long long)64 * f.groups * s.N * s.C >= (1ll << 31)) return false;
        if ((long long)g.B * g.OHF * g.OWF * s.out_ld >= (1ll << 31)) return false;
    }
    return true;
}

struct W32Plan { int splitk, c_per_split, tiles_x, tiles_y, ntiles, lean, pays, nb; };

W32Plan wino32_plan_nb(const mtd_conv_args& a, int count, int groups, int nb) {
    W32Plan pl{};
    pl.tiles_x = (a.g.OW + 2) / 3;
    pl.tiles_y = (a.g.OH + 2) / 3;
    pl.ntiles = a.g.B * pl.tiles_x * pl.tiles_y;
    pl.nb = nb;
    const long long blocks = (long long)((pl.ntiles + WT - 1) / WT) * (a.N / (32 * pl.nb)) * count;
    const int chunks = groups * a.C / 16;
    int sk = blocks <= 128 ? (int)(256 / blocks) : 1;
    if (sk > chunks / 4) sk = chunks / 4;
    if (sk > 16) sk = 16;
    if (sk < 1) sk = 1;
    static const int env_sk = [] { const char* e = mtd_lab_env("MTD_WINO_S2_SPLITK"); return e ? atoi(e) : 0; }();
    if (env_sk > 0) sk = env_sk < chunks ? env_sk : chunks;
    const int cps = (chunks + sk - 1) / sk;
    pl.splitk = (chunks + cps - 1) / cps;
    pl.c_per_split = cps * 16;
    static const int env_lean = [] { const char* e = mtd_lab_env("MTD_WINO_S2_LEAN"); return e ? atoi(e) : 1; }();
    const long long grid = blocks * pl.splitk;
    pl.lean = pl.nb == 2 && env_lean && (env_lean == 2 || (cps <= 8 && grid >= 384));
    // Does the form pay against the implicit GEMM (tools/wino_s2_probe.py, profiles/r5_wino_s2_probe.txt)?  A workgroup is 32 tiles x 64
    // (or 128) channels with a fixed cost outside its K loop, so: forward -- where the grid fills the 256 CUs' rounds to 80 % (1.2 .. 1.5x
    // on down1 / down3 at both batch sizes and down2 at 32 images); data gradient -- K is the layer's output channels, only down1's four
    // steps in the two-per-CU form come out ahead (1.36 .. 1.42x; down2: 1.09x at 64 images, 0.72x at 32).
    const double fill = (double)grid / (double)(((grid + 255) / 256) * 256);
    pl.pays = groups == 4 ? (fill >= 0.8) : (pl.lean && chunks <= 4);
    return pl;
}

W32Plan wino32_plan(const mtd_conv_args& a, int count, int groups) {
    static const int env_nb = [] { const char* e = mtd_lab_env("MTD_WINO_S2_NB"); return e ? atoi(e) : 0; }();
    if (env_nb == 4 && a.N % 128 == 0) return wino32_plan_nb(a, count, groups, 4);
    W32Plan pl = wino32_plan_nb(a, count, groups, 2);
    // forward grids the 64-channel workgroups leave short (down2 at 64 images: 144 of them, 0.93x): 128-channel workgroups and the
    // split of K that goes with them (72 x 3: 1.19x)
    if (!pl.pays && groups == 4 && a.N % 128 == 0 && env_nb != 2) {
        const W32Plan p4 = wino32_plan_nb(a, count, groups, 4);
        if (p4.pays) pl = p4;
    }
    return pl;
}

}  // namespace

extern "C" int mtd_winograd_s2_kmap(const mtd_geom* g, int* groups, int* kmap16) {
    W32Form f;
    if (!g || !groups || !kmap16 || !wino32_form(*g, f)) return MTD_EINVAL;
    *groups = f.groups;
    for (int i = 0; i < 16; ++i) kmap16[i] = f.kmap[i];
    return MTD_OK;
}

extern "C" int mtd_winograd_s2_weights(const mtd_wino_s2_weight_desc* table_dev, const mtd_wino_s2_weight_desc* table_host, int count, void* stream) {
    static_assert(sizeof(mtd_wino_s2_weight_desc) == sizeof(Wino32WDesc), "descriptor layouts must agree");
    if (!table_dev || !table_host || count <= 0) return MTD_EINVAL;
    long long most = 0;
    for (int i = 0; i < count; ++i) {
        const mtd_wino_s2_weight_desc& d = table_host[i];
        if (!d.src || !d.dst || d.N <= 0 || d.C <= 0 || (d.C % 8) || !(d.groups == 1 || d.groups == 4)) return MTD_EINVAL;
        for (int k = 0; k < 4 * d.groups; ++k)
            if (d.kmap[k] < 0 || d.kmap[k] > 15) return MTD_EINVAL;
        const long long t = (long long)d.N * d.C * d.groups;
        most = t > most ? t : most;
    }
    int gx = (int)((most + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(wino32_weights_kernel, dim3(gx, count < 64 ? count : 64), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const Wino32WDesc*>(table_dev), count);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// 0: not in the kernel's domain; 1: in the domain; 2: and the plan expects it to beat the implicit GEMM on this shape
extern "C" int mtd_conv_winograd_s2_ok(const mtd_conv_args* a, int count) {
    if (!wino32_eligible(a, count)) return 0;
    W32Form f;
    wino32_form(a[0].g, f);
    return wino32_plan(a[0], count, f.groups).pays ? 2 : 1;
}

// bytes of split-K workspace EACH set needs (0: no split)
extern "C" size_t mtd_conv_winograd_s2_ws_bytes(const mtd_conv_args* a, int count) {
    if (!wino32_eligible(a, count)) return 0;
    W32Form f;
    wino32_form(a[0].g, f);
    const W32Plan pl = wino32_plan(a[0], count, f.groups);
    return pl.splitk > 1 ? (size_t)pl.splitk * (size_t)geom_pixels(a[0].g) * a[0].N * sizeof(float) : 0;
}

// a[0 .. count): as for mtd_conv_igemm_multi, except that a[i].w points to the TRANSFORMED weights of set i
// (mtd_winograd_s2_weights with the kmap of a[i].g; w_sn / w_sc / w_st ignored).
extern "C" int mtd_conv_winograd_s2(const mtd_conv_args* a, int count, void* stream) {
    if (!wino32_eligible(a, count)) return MTD_EINVAL;
    W32Form f0;
    wino32_form(a[0].g, f0);
    const W32Plan pl = wino32_plan(a[0], count, f0.groups);
    W32Params wp;
    IgemmParams& p = wp.p;
    p.a = a[0];
    p.M = (int)geom_pixels(a[0].g);
    p.splitk = pl.splitk;
    p.c_per_split = pl.c_per_split;
    {
        const long long npix = (long long)a[0].g.B * a[0].g.IH * a[0].g.IW;
        p.in_bytes = (unsigned)(((npix - 1) * a[0].in_ld + a[0].C) * 4);
    }
    p.w_bytes = 0;
    for (int t = 0; t < 16; ++t) p.tap_dy[t] = p.tap_dx[t] = p.tap_delta[t] = p.tap_kidx[t] = 0;
    {
        const mtd_geom& g = a[0].g;
        p.out_identity = (g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW);
    }
    p.out_linear = 0;
    p.xcd_map = 0;
    p.nt_store = 0;
    p.fin = 0;
    bool slabs16 = pl.splitk > 1 && (a[0].N % 4) == 0;
    for (int i = 0; i < count; ++i) slabs16 = slabs16 && aligned16(a[i].ws);
    p.wide = (wide_epilogue_ok(a[0]) ? 1 : 0) | (slabs16 ? 2 : 0);
    wp.tiles_x = pl.tiles_x;
    wp.tiles_y = pl.tiles_y;
    wp.ntiles = pl.ntiles;
    wp.groups = f0.groups;
    wp.gsteps = a[0].C / 16;
    wp.ps = f0.ps;
    wp.nchunk = f0.groups * a[0].C / 8;
    wp.nsets = count;
    wp.w_bytes = (unsigned)((long long)64 * f0.groups * a[0].N * a[0].C);
    {
        const double wbytes = 64.0 * f0.groups * a[0].C * a[0].N * count, ibytes = (double)a[0].g.B * a[0].g.IH * a[0].g.IW * a[0].C * 4;
        wp.xcd_order = wbytes >= ibytes ? 1 : 2;
    }
    const size_t need = pl.splitk > 1 ? (size_t)pl.splitk * (size_t)p.M * a[0].N * sizeof(float) : 0;
    for (int i = 0; i < 4; ++i) {
        const mtd_conv_args& s = a[i < count ? i : 0];
        W32Form f;
        wino32_form(s.g, f);
        if (need && (!s.ws || s.ws_bytes < need)) return MTD_EWS;
        wp.set[i] = W32Set{s.w, s.ws, f.base_y, f.base_x, s.g.out_oy, s.g.out_ox};
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((wp.ntiles + WT - 1) / WT, a[0].N / (32 * pl.nb), pl.splitk * count);
    double bytes = 0.0;
    for (int i = 0; i < count; ++i) bytes += algorithmic_bytes(&a[i]);
    const int prof = mtd_prof_begin(0, pl.nb == 4 ? 31 : pl.lean ? 30 : 29, pl.splitk, (long long)p.M * count, a[0].N, a[0].C, a[0].g.TH * a[0].g.TW, s, bytes);
    if (pl.nb == 4) MTD_LAUNCH((wino32_conv_kernel<4>), grid, dim3(512), 0, s, wp);
    else if (pl.lean) MTD_LAUNCH((wino32_conv_kernel<2, true>), grid, dim3(512), 0, s, wp);
    else MTD_LAUNCH((wino32_conv_kernel<2>), grid, dim3(512), 0, s, wp);
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    if (pl.splitk > 1) {
        for (int i = 0; i < count; ++i) {
            IgemmParams q = p;
            q.a = a[i];
            const mtd_geom& g = a[i].g;
            q.out_identity = (g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW);
            const long long total = (long long)p.M * a[i].N;
            const bool vec = splitk_vec_ok(a[i], p.M);
            int blocks = (int)(((vec ? total / 4 : total) + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            if (vec) hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, s, q);
            else hipLaunchKernelGGL(splitk_epilogue_scalar_kernel, dim3(blocks), dim3(256), 0, s, q);
            MTD_LAUNCH_CHECK();
        }
    }
    return MTD_OK;
}
