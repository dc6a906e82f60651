// Winograd F(3x3, 2x2) weight gradient of the 4x4 / stride-2 / padding-1 layers (arch/Ours/networks.py:185-215 down1..3) on fp32
// MFMA -- the transpose of conv_wino_s2.h's forward form (included by conv_wgrad.hip after conv_wgrad_wino.h, whose LDS layout,
// chunk loop, tile cursor and slab contract it shares).
//
// Polyphase view: tap (ky, kx) = (2 jy + py, 2 jx + px) of the 4x4 filter is entry (jy, jx) of a 2x2 stride-1 filter over phase
// (py, px) of the padded input (conv_wino_s2.h), so with the 4 C channels c' = (py, px, c)
//     dW' = G^T [ sum over 3x3 output tiles  (A dY A^T) (.) (B^T d B) ] G        dY: the tile's 3x3 cotangent, d: its 4x4 patch of the phase
//     A = [1 0 0; 1 1 1; 1 -1 1; 0 0 -1],  B^T as F(2,3)'s,  G = [1 0; .5 .5; .5 -.5; 0 1]
// -- 16 multiplications per tile and (n, c') pair for 9 output pixels x 4 taps = 36 of the direct form: 2.25x fewer MFMA flops (less the
// ragged last tiles).  A 512-thread workgroup owns a 64 (n) x 64 (c') block -- one phase, 64 of its channels -- for all 16 positions
// over one slice of the tiles; chunks of 8 tiles, one per wave:
//   * U = B^T d B: thread (tile, channel quad, patch row qp) loads its row of the phase -- image row 2 (3 ty + qp) - 1 + py, pixels
//     2 (3 tx + j) - 1 + px -- as four 16-byte vectors; the transform is wgrad_wino_kernel's;
//   * V = A dY A^T: thread (tile, channel quad, cotangent row qp < 3) loads the row's three pixels, forms R[b] = (dY A^T)[qp][b] in
//     registers and row a = qp of A R across the quad with DPP (lane a takes alpha R0 + beta R1 + gamma R2); sums the bias gradient;
//   * epilogue: G^T . G per (n, c'): four values, the taps (2 jy + py) 4 + 2 jx + px of the slab (layout of the other weight-gradient
//     kernels: slab[(tap N + n) C + c], bias row at tap = 16).
// Domain: forward geometry of Conv2d(k4, s2, p1), N and C multiples of 64.  Executed flops 2 tiles 16 N 4 C.

//   V[a][b] of the lane = alpha r[lane 0] + beta r[lane 1] + gamma r[lane 2]  per channel
__device__ __forceinline__ f32x4 wgs_quad3(f32x4 r, float alpha, float beta, float gamma) {
    float v0, v1, v2, v3;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %4, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %1, %5, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %2, %6, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %3, %7, %8 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %4, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %5, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %6, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %7, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %4, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %5, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %6, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %7, %10 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(alpha), "v"(beta), "v"(gamma));
    return f32x4{v0, v1, v2, v3};
}

__global__ __launch_bounds__(512, 1) void wgrad_wino_s2_kernel(const WgradWinoParams wp) {
    __shared__ __attribute__((aligned(16))) float Ls[2 * WGW_BUF];
    const WgradParams& p = wp.w;
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int ncb = 4 * a.C / 64;                                          // c' blocks: phase-major
    const int nblk = blockIdx.y / ncb, cblk = blockIdx.y - nblk * ncb;
    const int n0 = nblk * 64;
    const int phase = (cblk * 64) / a.C, c0 = cblk * 64 - phase * a.C;
    const int py = phase >> 1, px = phase & 1;
    const int second = (wp.ns_first > 0 && (int)blockIdx.x >= wp.ns_first) ? 1 : 0;
    const int zk = blockIdx.x - second * wp.ns_first;
    const int tile_lo = second ? wp.first_tiles : 0;
    const int tile_hi = (wp.ns_first > 0 && !second) ? wp.first_tiles : wp.ntiles;
    const int ck_beg = zk * wp.chunks_per_split;
    const int nchunks_all = (tile_hi - tile_lo + WGW_T - 1) / WGW_T;
    const int ck_end = min(nchunks_all, ck_beg + wp.chunks_per_split);
    const int nck = ck_end - ck_beg;
    const int ck_last = ck_end - 1;

    const int qp = tid & 3, cq = (tid >> 2) & 15, t8 = tid >> 6;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q), (short)0, (int)p.q_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p), (short)0, (int)p.p_bytes, 0x00020000);
    const int qpx_b = 2 * a.q_ld * 4;          // one patch pixel = two image pixels
    const int ppx_b = a.p_ld * 4;
    struct Pre { f32x4 d[4]; f32x4 y[3]; };
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    const int tiles_y = wp.tiles_per_image / wp.tiles_x;
    const int d_tx = WGW_T % wp.tiles_x, d_ty = (WGW_T / wp.tiles_x) % tiles_y, d_b = WGW_T / wp.tiles_per_image;
    int cur_ck = ck_beg, cur_tg = tile_lo + ck_beg * WGW_T + wave;
    int cur_b = cur_tg / wp.tiles_per_image, cur_ty, cur_tx;
    {
        const int rr = cur_tg - cur_b * wp.tiles_per_image;
        cur_ty = rr / wp.tiles_x;
        cur_tx = rr - cur_ty * wp.tiles_x;
    }
    // the lane's part of the offsets: patch row qp = two image rows down per row, the phase's origin; cotangent row qp
    const unsigned u_lane = (unsigned)((((2 * qp + py) * g.IW + px) * a.q_ld + c0 + 4 * cq) * 4);
    const unsigned p_lane = (unsigned)((qp * g.OW * a.p_ld + n0 + 4 * cq) * 4);
    unsigned nx_uv = 0, nx_pv = 0;
    bool nx_rowok = false, nx_prow = false;
    int nx_tx = 0;
    auto prep_next = [&]() {
        const bool tv = cur_tg < tile_hi;
        // U: image row 6 ty - 1 + py + 2 qp, pixels 6 tx - 1 + px + 2 j  (offsets formed modulo 2^32: every VALID pixel's is in range)
        const unsigned u_s = (((unsigned)cur_b * (unsigned)g.IH + (unsigned)(6 * cur_ty - 1)) * (unsigned)g.IW + (unsigned)(6 * cur_tx - 1)) * (unsigned)a.q_ld * 4u;
        nx_uv = u_s + u_lane;
        nx_rowok = tv & ((unsigned)(6 * cur_ty - 1 + py + 2 * qp) < (unsigned)g.IH);
        nx_tx = cur_tx;
        // V: cotangent row 3 ty + qp (qp < 3), pixels 3 tx + k
        const unsigned p_s = (((unsigned)cur_b * (unsigned)g.OH + (unsigned)(3 * cur_ty)) * (unsigned)g.OW + (unsigned)(3 * cur_tx)) * (unsigned)a.p_ld * 4u;
        nx_pv = p_s + p_lane;
        nx_prow = tv & (qp < 3) & (3 * cur_ty + qp < g.OH);
        const int adv = cur_ck < ck_last ? 1 : 0;
        cur_ck += adv;
        cur_tg += adv ? WGW_T : 0;
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
            const bool colok = (unsigned)(6 * nx_tx - 1 + px + 2 * j) < (unsigned)g.IW;
            r.d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, (nx_rowok & colok) ? nx_uv + (unsigned)(j * qpx_b) : 0x80000000u, 0, 0));
        } else {
            const int k = j - 4;
            const bool colok = 3 * nx_tx + k < g.OW;
            r.y[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, (nx_prow & colok) ? nx_pv + (unsigned)(k * ppx_b) : 0x80000000u, 0, 0));
        }
    };
    auto load_next = [&](Pre& r) {
        prep_next();
#pragma unroll
        for (int j = 0; j < 7; ++j) issue_next(r, j);
    };
    const float usign = qp == 1 ? 1.f : -1.f;
    // row a = qp of A R:  A = [1 0 0; 1 1 1; 1 -1 1; 0 0 -1]
    const float valpha = qp == 3 ? 0.f : 1.f, vbeta = qp == 1 ? 1.f : (qp == 2 ? -1.f : 0.f), vgamma = qp == 0 ? 0.f : (qp == 3 ? -1.f : 1.f);
    auto tr_u = [&](float* Lb, const Pre& r, int j) {
        const f32x4 rj = j == 0 ? r.d[0] - r.d[2] : (j == 1 ? r.d[1] + r.d[2] : (j == 2 ? r.d[2] - r.d[1] : r.d[1] - r.d[3]));
        *reinterpret_cast<f32x4*>(Lb + t8 * 64 + 4 * cq + (4 * qp + j) * WGW_PL) = wgw_quad_rows(rj, usign);
    };
    // column b of V: R[b] = (dY A^T)[qp][b] = y0 | y0 + y1 + y2 | y0 - y1 + y2 | -y2, then down the quad
    auto tr_v = [&](float* Lb, const Pre& r, int b) {
        const f32x4 rb = b == 0 ? r.y[0] : (b == 1 ? (r.y[0] + r.y[2]) + r.y[1] : (b == 2 ? (r.y[0] + r.y[2]) - r.y[1] : -r.y[2]));
        *reinterpret_cast<f32x4*>(Lb + 16 * WGW_PL + t8 * 64 + 4 * cq + (4 * qp + b) * WGW_PL) = wgs_quad3(rb, valpha, vbeta, vgamma);
    };
    auto transform_store = [&](float* Lb, const Pre& r, float live) {
        tr_u(Lb, r, 0); tr_u(Lb, r, 1); tr_u(Lb, r, 2); tr_u(Lb, r, 3);
        dbacc += live * ((r.y[0] + r.y[1]) + r.y[2]);
        tr_v(Lb, r, 0); tr_v(Lb, r, 1); tr_v(Lb, r, 2); tr_v(Lb, r, 3);
    };

    f32x16 acc[2][2][2];                       // [position][n half][c half]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[x][h][h2][e] = 0.f;

    Pre pa, pb;
    if (nck > 0) {
        load_next(pa);
        transform_store(Ls, pa, 1.f);
        load_next(pa);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    auto one_chunk = [&](int k, const Pre& cur, Pre& nxt) {
        const float* Lc = Ls + (k & 1) * WGW_BUF;
        float* Ln = Ls + ((k + 1) & 1) * WGW_BUF;
        prep_next();                                                   // (chunk k + 2, or the last one again)
        __builtin_amdgcn_sched_barrier(0);
        float fa[2][2], fb[2][2];
        auto frag = [&](int gi, int pp) {
            const int sst = gi >> 1, x = gi & 1;
            const float* up = Lc + (2 * wave + x) * WGW_PL + (2 * sst + kh) * 64 + l31;
            const float* vp = up + 16 * WGW_PL;
            fa[pp][0] = vp[0]; fa[pp][1] = vp[32];
            fb[pp][0] = up[0]; fb[pp][1] = up[32];
        };
        frag(0, 0);
        const float live = (k + 1 < nck) ? 1.f : 0.f;
#pragma unroll
        for (int gi = 0; gi < 8; ++gi) {
            const int pp = gi & 1, x = gi & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 1 < 8) frag(gi + 1, pp ^ 1);
            acc[x][0][0] = mfma32(fa[pp][0], fb[pp][0], acc[x][0][0]);
            acc[x][0][1] = mfma32(fa[pp][0], fb[pp][1], acc[x][0][1]);
            acc[x][1][0] = mfma32(fa[pp][1], fb[pp][0], acc[x][1][0]);
            acc[x][1][1] = mfma32(fa[pp][1], fb[pp][1], acc[x][1][1]);
            if (gi < 7) issue_next(nxt, gi);                           // one request per MFMA group
            __builtin_amdgcn_sched_barrier(0);
            // the transform of chunk k + 1, a piece per group
            if (gi == 0) { tr_u(Ln, cur, 0); }
            else if (gi == 1) { tr_u(Ln, cur, 1); }
            else if (gi == 2) { tr_u(Ln, cur, 2); }
            else if (gi == 3) { tr_u(Ln, cur, 3); dbacc += live * ((cur.y[0] + cur.y[1]) + cur.y[2]); }
            else tr_v(Ln, cur, gi - 4);
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int k = 0; k < nck; k += 2) {
        one_chunk(k, pa, pb);
        if (k + 1 < nck) one_chunk(k + 1, pb, pa);
    }

    // ---- epilogue: G^T dU G per (n, c'), one 32 x 32 sub-block at a time; four taps of the phase
    float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
    const int en_c = tid & 31, en_n = tid >> 5;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                float* X = Ls + (2 * wave + x) * (32 * WGW_XLD) + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) X[mfma32_row(e, lane) * WGW_XLD] = acc[x][h][h2][e];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int nl = en_n + 16 * r;
                float m[16];
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) m[xi] = Ls[(xi * 32 + nl) * WGW_XLD + en_c];
                // t = G^T m (2 x 4), out = t G (2 x 2);  G^T = [1 .5 .5 0; 0 .5 -.5 1]
                float t[2][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    t[0][b] = m[b] + 0.5f * (m[4 + b] + m[8 + b]);
                    t[1][b] = 0.5f * (m[4 + b] - m[8 + b]) + m[12 + b];
                }
                float* o = slab + ((long long)(n0 + 32 * h + nl)) * a.C + c0 + 32 * h2 + en_c;
                const long long tap_stride = (long long)a.N * a.C;
#pragma unroll
                for (int jy = 0; jy < 2; ++jy) {
                    o[((2 * jy + py) * 4 + px) * tap_stride] = t[jy][0] + 0.5f * (t[jy][1] + t[jy][2]);
                    o[((2 * jy + py) * 4 + 2 + px) * tap_stride] = 0.5f * (t[jy][1] - t[jy][2]) + t[jy][3];
                }
            }
            __syncthreads();
        }
    if (a.db && cblk == 0) {
        f32x4 s;
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = ((wgw_quad(dbacc[c], 2) + wgw_quad(dbacc[c], 3)) + wgw_quad(dbacc[c], 4)) + wgw_quad(dbacc[c], 5);
        if (qp == 0) *reinterpret_cast<f32x4*>(Ls + t8 * 64 + 4 * cq) = s;
        __syncthreads();
        if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += Ls[w * 64 + tid];
            slab[(long long)p.T * a.N * a.C + n0 + tid] = v;
        }
    }
}

// the kernel's domain: the forward geometry of Conv2d(k4, s2, p1), N and C multiples of 64
bool wgrad_wino_s2_ok(const mtd_wgrad_args& a) {
    const mtd_geom& g = a.g;
    if (g.TH != 4 || g.TW != 4 || g.in_sy != 2 || g.in_sx != 2 || g.tap_dy != 1 || g.tap_dx != 1 || g.off_y != -1 || g.off_x != -1) return false;
    if (g.ky0 != 0 || g.kx0 != 0 || g.ky_step != 1 || g.kx_step != 1 || g.KW != 4) return false;
    if (g.IH != 2 * g.OH || g.IW != 2 * g.OW) return false;
    if ((a.N % 64) || (a.C % 64)) return false;
    if (!aligned16(a.p) || !aligned16(a.q) || (a.p_ld % 4) || (a.q_ld % 4)) return false;
    return true;
}
inline long long wgrad_wino_s2_blocks(const mtd_wgrad_args& a) { return (long long)(a.N / 64) * (4 * a.C / 64); }
inline long long wgrad_wino_s2_tiles(const mtd_wgrad_args& a, long long images) { return images * ((a.g.OH + 2) / 3) * ((a.g.OW + 2) / 3); }
