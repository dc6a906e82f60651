// Winograd F(2x2, 3x3) weight gradient of the generator's 32 -> 32 channel 3x3 layers (arch/Ours/networks.py:95-164: Conv2d /
// ConvTranspose2d(32, 32, 3, 1, 1)) on fp32 MFMA -- wgrad_wino_kernel (conv_wgrad_wino.h) for ONE 32 x 32 block of (n, c):
//   * a chunk is SIXTEEN tiles (two per wave: the halves of a wave are neighbouring tiles of one tile row), so the per-position
//     product over a chunk is 32 (n) x 32 (c) x 16 tiles = eight k-steps of v_mfma_f32_32x32x2_f32, 16 MFMAs per wave and chunk
//     for the same transform work per thread as the 64 x 64 kernel's 32;
//   * LDS planes [xi][tile 0..15][32 channels] (the 64 x 64 kernel's [tile 0..7][64]: same size, same skew);
//   * the wave's tile PAIR walks the slice in scalar registers; the lane adds its half to the column -- tile rows hold an even
//     number of tiles (map width a multiple of 4) and ranges start at even tiles, so a pair never straddles a row or a range;
//   * the transposed-conv layers (tap_d = -1: the decoder's ConvTranspose2d) differ only in which tap of the slab a position
//     pair lands on (flipped).
// Slab layout as everywhere: slab[(tap 32 + n) 32 + c], bias row at tap = 9.  Executed flops 2 M 32 32 4.
constexpr int W32_T = 16;                        // tiles per chunk
constexpr int W32_PL = W32_T * 32 + 4;           // plane stride (floats) of one position
constexpr int W32_BUF = 2 * 16 * W32_PL;         // one chunk buffer: U planes then V planes

__global__ __launch_bounds__(512, 1) void wgrad_wino32_kernel(const WgradWinoParams wp, int flip) {
    __shared__ __attribute__((aligned(16))) float Ls[2 * W32_BUF];
    const WgradParams& p = wp.w;
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int zk = blockIdx.x;
    const int tile_hi = wp.ntiles;
    const int ck_beg = zk * wp.chunks_per_split;
    const int nchunks_all = (tile_hi + W32_T - 1) / W32_T;
    const int ck_end = min(nchunks_all, ck_beg + wp.chunks_per_split);
    const int nck = ck_end - ck_beg;
    const int ck_last = ck_end - 1;

    // ---- transform roles: thread (tile t16 of the chunk = 2 wave + half, channel quad cq, quad position qp)
    const int qp = tid & 3, cq = (tid >> 2) & 7, t16 = tid >> 5, half = kh;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q), (short)0, (int)p.q_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p), (short)0, (int)p.p_bytes, 0x00020000);
    const int qpx_b = a.q_ld * 4;
    struct Pre { f32x4 d[4]; f32x4 y; };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    const int tiles_y = wp.tiles_per_image / wp.tiles_x;
    const int d_tx = W32_T % wp.tiles_x, d_ty = (W32_T / wp.tiles_x) % tiles_y, d_b = W32_T / wp.tiles_per_image;
    int cur_ck = ck_beg, cur_tg = ck_beg * W32_T + 2 * wave;             // the pair's first tile
    int cur_b = cur_tg / wp.tiles_per_image, cur_ty, cur_tx;
    {
        const int rr = cur_tg - cur_b * wp.tiles_per_image;
        cur_ty = rr / wp.tiles_x;
        cur_tx = rr - cur_ty * wp.tiles_x;
    }
    // the lane's part of the offsets: patch row qp, its tile of the pair (two pixels along the row per tile), its channels
    const unsigned u_lane = (unsigned)(((qp * g.IW + 2 * half) * a.q_ld + 4 * cq) * 4);
    const unsigned p_lane = (unsigned)((((qp >> 1) * g.OW + (qp & 1) + 2 * half) * a.p_ld + 4 * cq) * 4);
    unsigned nx_uv = 0, nx_pv = 0x80000000u;
    bool nx_rowok = false;
    int nx_tx = 0;
    auto prep_next = [&]() {
        const bool tv = cur_tg < tile_hi;
        const unsigned u_s = (((unsigned)cur_b * (unsigned)g.IH + (unsigned)(2 * cur_ty - 1)) * (unsigned)g.IW + (unsigned)(2 * cur_tx)) * (unsigned)a.q_ld * 4u;
        nx_uv = u_s + u_lane;
        nx_rowok = tv & ((unsigned)(2 * cur_ty - 1 + qp) < (unsigned)g.IH);
        nx_tx = cur_tx + half;
        const unsigned p_s = tv ? (((unsigned)cur_b * (unsigned)g.OH + (unsigned)(2 * cur_ty)) * (unsigned)g.OW + (unsigned)(2 * cur_tx)) * (unsigned)a.p_ld * 4u : 0x80000000u;
        nx_pv = tv ? p_s + p_lane : 0x80000000u;
        const int adv = cur_ck < ck_last ? 1 : 0;
        cur_ck += adv;
        cur_tg += adv ? W32_T : 0;
        cur_tx += adv ? d_tx : 0;
        const int c1 = cur_tx >= wp.tiles_x ? 1 : 0;
        cur_tx -= c1 ? wp.tiles_x : 0;
        cur_ty += (adv ? d_ty : 0) + c1;
        const int c2 = cur_ty >= tiles_y ? 1 : 0;
        cur_ty -= c2 ? tiles_y : 0;
        cur_b += (adv ? d_b : 0) + c2;
    };
    auto issue_next = [&](Pre& r, int j) {
        if (j < 4) {
            const bool colok = (unsigned)(2 * nx_tx - 1 + j) < (unsigned)g.IW;
            r.d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, (nx_rowok & colok) ? nx_uv + (unsigned)((j - 1) * qpx_b) : 0x80000000u, 0, 0));
        } else {
            r.y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, nx_pv, 0, 0));
        }
    };
    auto load_next = [&](Pre& r) {
        prep_next();
#pragma unroll
        for (int j = 0; j < 5; ++j) issue_next(r, j);
    };
    const float usign = qp == 1 ? 1.f : -1.f;
    const float valpha = qp == 3 ? 0.f : 1.f, vbeta = qp == 0 ? 0.f : (qp == 1 ? 1.f : -1.f);
    f32x4 vs0, vs1;
    auto tr_u = [&](float* Lb, const Pre& r, int j) {
        const f32x4 rj = j == 0 ? r.d[0] - r.d[2] : (j == 1 ? r.d[1] + r.d[2] : (j == 2 ? r.d[2] - r.d[1] : r.d[1] - r.d[3]));
        *reinterpret_cast<f32x4*>(Lb + t16 * 32 + 4 * cq + (4 * qp + j) * W32_PL) = wgw_quad_rows(rj, usign);
    };
    auto tr_v_gather = [&](const Pre& r, float live) {
        dbacc += live * r.y;
        wgw_quad_pair(r.y, valpha, vbeta, vs0, vs1);
    };
    auto tr_v_store = [&](float* Lb, int b0) {
        float* vo = Lb + 16 * W32_PL + t16 * 32 + 4 * cq;
        if (b0 == 0) {
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 0) * W32_PL) = vs0;
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 1) * W32_PL) = vs0 + vs1;
        } else {
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 2) * W32_PL) = vs0 - vs1;
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 3) * W32_PL) = -vs1;
        }
    };
    auto transform_store = [&](float* Lb, const Pre& r, float live) {
        tr_u(Lb, r, 0); tr_u(Lb, r, 1); tr_u(Lb, r, 2); tr_u(Lb, r, 3);
        tr_v_gather(r, live);
        tr_v_store(Lb, 0);
        tr_v_store(Lb, 2);
    };

    f32x16 acc[2];                             // [position]: 32 (n) x 32 (c)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;

    Pre pa, pb;
    if (nck > 0) {
        load_next(pa);
        transform_store(Ls, pa, 1.f);
        load_next(pa);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    auto one_chunk = [&](int k, const Pre& cur, Pre& nxt) {
        const float* Lc = Ls + (k & 1) * W32_BUF;
        float* Ln = Ls + ((k + 1) & 1) * W32_BUF;
        prep_next();                                                   // (chunk k + 2, or the last one again)
        __builtin_amdgcn_sched_barrier(0);
        float fa[2][2], fb[2][2];                                      // [ping-pong][position]
        auto frag = [&](int gi, int pp) {                             // group gi = k-step: tiles 2 gi + kh
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const float* up = Lc + (2 * wave + x) * W32_PL + (2 * gi + kh) * 32 + l31;
                fa[pp][x] = up[16 * W32_PL];
                fb[pp][x] = up[0];
            }
        };
        frag(0, 0);
        const float live = (k + 1 < nck) ? 1.f : 0.f;
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int pp = gi & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 1 < 8) frag(gi + 1, pp ^ 1);
            acc[0] = mfma32(fa[pp][0], fb[pp][0], acc[0]);
            acc[1] = mfma32(fa[pp][1], fb[pp][1], acc[1]);
            if (gi < 5) issue_next(nxt, gi);                           // one request per MFMA group
            __builtin_amdgcn_sched_barrier(0);
            if (gi >= 1 && gi <= 4) tr_u(Ln, cur, gi - 1);
            else if (gi == 5) tr_v_gather(cur, live);
            else if (gi == 6) tr_v_store(Ln, 0);
            else if (gi == 7) tr_v_store(Ln, 2);
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int k = 0; k < nck; k += 2) {
        one_chunk(k, pa, pb);
        if (k + 1 < nck) one_chunk(k + 1, pb, pa);
    }

    // ---- epilogue: G^T dU G per (n, c) through X[xi][n][c]; the bias gradient
    float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
    const int en_c = tid & 31, en_n = tid >> 5;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float* X = Ls + (2 * wave + x) * (32 * WGW_XLD) + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) X[mfma32_row(e, lane) * WGW_XLD] = acc[x][e];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int nl = en_n + 16 * r;
        float m[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi] = Ls[(xi * 32 + nl) * WGW_XLD + en_c];
        float t[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float hs = 0.5f * (m[4 + b] + m[8 + b]), hd = 0.5f * (m[4 + b] - m[8 + b]);
            t[0][b] = m[b] + hs;
            t[1][b] = hd;
            t[2][b] = hs + m[12 + b];
        }
        float* o = slab + (long long)nl * 32 + en_c;
        const int tap_stride = 32 * 32;
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
            const float hs = 0.5f * (t[pr][1] + t[pr][2]), hd = 0.5f * (t[pr][1] - t[pr][2]);
            // correlation position (pr, q) is filter entry (pr, q) of a forward conv, (2 - pr, 2 - q) of a transposed one
            const int t0 = flip ? 8 - (pr * 3 + 0) : pr * 3 + 0, t1 = flip ? 8 - (pr * 3 + 1) : pr * 3 + 1, t2 = flip ? 8 - (pr * 3 + 2) : pr * 3 + 2;
            o[t0 * tap_stride] = t[pr][0] + hs;
            o[t1 * tap_stride] = hd;
            o[t2 * tap_stride] = hs + t[pr][3];
        }
    }
    __syncthreads();
    if (a.db) {
        f32x4 s;
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = ((wgw_quad(dbacc[c], 2) + wgw_quad(dbacc[c], 3)) + wgw_quad(dbacc[c], 4)) + wgw_quad(dbacc[c], 5);
        if (qp == 0) *reinterpret_cast<f32x4*>(Ls + t16 * 32 + 4 * cq) = s;
        __syncthreads();
        if (tid < 32) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) v += Ls[w * 32 + tid];
            slab[(long long)p.T * 32 * 32 + tid] = v;
        }
    }
}

// the kernel's domain: 3x3 / stride 1 / pad 1 in the forward (tap_d = +1, off = -1) or the transposed (tap_d = -1, off = +1) tap order,
// N = C = 32, even height, width a multiple of 4 (tile rows of an even number of tiles)
bool wgrad_wino32_ok(const mtd_wgrad_args& a) {
    const mtd_geom& g = a.g;
    if (g.TH != 3 || g.TW != 3 || g.in_sy != 1 || g.in_sx != 1 || g.tap_dy != g.tap_dx) return false;
    if (!((g.tap_dy == 1 && g.off_y == -1 && g.off_x == -1) || (g.tap_dy == -1 && g.off_y == 1 && g.off_x == 1))) return false;
    if (g.ky0 != 0 || g.kx0 != 0 || g.ky_step != 1 || g.kx_step != 1 || g.KW != 3) return false;
    if (g.IH != g.OH || g.IW != g.OW || (g.OH & 1) || (g.OW & 3)) return false;
    if (a.N != 32 || a.C != 32) return false;
    if (!aligned16(a.p) || !aligned16(a.q) || (a.p_ld % 4) || (a.q_ld % 4)) return false;
    return true;
}
