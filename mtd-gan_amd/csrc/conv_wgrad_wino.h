// Winograd F(2x2, 3x3) weight gradient on fp32 MFMA (included by conv_wgrad.hip inside its anonymous namespace).
//
//   dW = G^T [ sum over 2x2 output tiles  (A dY A^T) (.) (B^T d B) ] G        dY: the tile's cotangent, d: its 4x4 input patch
//
// -- the transpose of conv_winograd.hip's forward form: 16 multiplications per tile and (n, c) pair instead of 36.  For each
// of the 16 transform positions xi an independent product over the TILES:
//     dU_xi[n][c] += V_xi[tile][n] * U_xi[tile][c],     V = A dY A^T (4x4 from 2x2),  U = B^T d B.
// A 512-thread workgroup owns a 64 (n) x 64 (c) block for all 16 positions -- wave w holds positions 2w, 2w+1 as 2 x 2 x 2
// accumulator blocks of 32 x 32 (128 registers) -- over one slice of the tiles (pixel split: the slices' results go to slabs
// that the existing fixed-order reduce sums, conv_wgrad.hip).  K runs in chunks of 8 tiles:
//   * both operands need a transform, so both go through LDS, [xi][tile][64 channels]: a fragment of a k-step is one
//     ds_read_b32 per lane (lane = channel, k half = tile parity), conflict-free;
//   * thread (tile, channel quad, patch row) loads its row of the input patch as four 16-byte vectors, applies B along the row
//     in registers and B^T across the quad with DPP (conv_winograd.hip's transform), stores one row of U as four 16-byte
//     vectors; thread (tile, channel quad, pixel) loads its pixel of the cotangent tile as one 16-byte vector, gathers the
//     quad's four pixels with DPP, forms row a = its quad position of A dY A^T, stores four 16-byte vectors -- and sums the
//     cotangent for the bias gradient on the way;
//   * the next chunk's loads are requested at the top of a chunk and transformed at its end, between the MFMAs; two LDS
//     buffers, one workgroup barrier per chunk;
//   * at the end the 16 positions of an (n, c) pair sit in 8 waves: they meet in LDS one 32 x 32 sub-block at a time, thread
//     (n, c) applies G^T . G (16 -> 9 values) and writes the nine taps of the slab.
// Slab layout = the other weight-gradient kernels': slab[(tap * N + n) * C + c], bias row at tap = 9.
// Domain: forward-oriented 3x3 / stride 1 / pad 1 geometry, even map sides, N and C multiples of 64.
// Roofline: fp32 MFMA; executed flops 2 M N C 4 of the algorithmic 2 M N C 9.

constexpr int WGW_T = 8;                       // tiles per chunk
constexpr int WGW_PL = WGW_T * 64 + 4;         // plane stride (floats) of one position: 16 bytes of skew per position keeps the four
                                               // quad lanes' 16-byte stores (positions 4 i' + j) on different banks
constexpr int WGW_BUF = 2 * 16 * WGW_PL;       // one chunk buffer: U planes then V planes
constexpr int WGW_XLD = 33;

struct WgradWinoParams {
    WgradParams w;
    int ntiles, tiles_x, tiles_per_image;
    int chunks_per_split;
    // pair form (mtd_conv_wgrad_pair): the batch is two image ranges with a weight gradient each -- slabs 0 .. ns_first - 1 are
    // sums over tiles [0, first_tiles), the others over [first_tiles, ntiles).  ns_first = 0: one range.
    int ns_first, first_tiles;
    // second cotangent (mtd_conv_wgrad_pair_sum): the gradient is taken from p + p_add, summed as the operands arrive (a decoder
    // layer's cotangents of two task passes: a weight gradient is linear in its cotangent).  Same layout and extent as a.p; NULL: none.
    const float* p_add;
};

__device__ __forceinline__ float wgw_quad(float v, int ctrl) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl) {
        case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x64, 0xf, 0xf, true); break;      // lanes [0, 1, 2, 1]
        case 1: r = __builtin_amdgcn_update_dpp(0, x, 0xDA, 0xf, 0xf, true); break;      // lanes [2, 2, 1, 3]
        case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, true); break;      // lane 0
        case 3: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xf, 0xf, true); break;      // lane 1
        case 4: r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xf, 0xf, true); break;      // lane 2
        default: r = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xf, 0xf, true); break;     // lane 3
    }
    return __builtin_bit_cast(float, r);
}

// The quad arithmetic written out with the permuted operand taken directly by the arithmetic instruction (v_mul_f32_dpp,
// v_add_f32_dpp, v_fmac_f32_dpp) instead of a v_mov_b32_dpp per operand: beside fp32 MFMAs every vector instruction is paid in
// full (tools/overlap_probe.hip).  The s_nop covers the two wait states between a vector write of the source and
// its first DPP read.
//   wgw_quad_rows:  r[perm0] + sign r[perm1]  per channel  (F(2,3) across the four lanes that hold a patch's rows)
__device__ __forceinline__ f32x4 wgw_quad_rows(f32x4 r, float sign) {
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
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(sign));
    return f32x4{u0, u1, u2, u3};
}
//   wgw_quad_pair:  s0 = alpha y[lane 0] + beta y[lane 2],  s1 = alpha y[lane 1] + beta y[lane 3]  per channel (the 2 x 2
//   cotangent tile of a quad, one pixel per lane, combined down its rows)
__device__ __forceinline__ void wgw_quad_pair(f32x4 y, float alpha, float beta, f32x4& s0, f32x4& s1) {
    float a0, a1, a2, a3, b0, b1, b2, b3;
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %8, %12 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %1, %9, %12 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %2, %10, %12 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %3, %11, %12 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %4, %8, %12 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %5, %9, %12 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %6, %10, %12 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %7, %11, %12 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %0, %8, %13 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %9, %13 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %10, %13 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %11, %13 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %4, %8, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %5, %9, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %6, %10, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %7, %11, %13 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(alpha), "v"(beta));
    s0 = f32x4{a0, a1, a2, a3};
    s1 = f32x4{b0, b1, b2, b3};
}

#ifndef WGW_SPREAD
#define WGW_SPREAD 1      // 1 (round 5): a chunk's six requests one per MFMA group; 0: all at the top of the chunk
#endif
__global__ __launch_bounds__(512, 1) void wgrad_wino_kernel(const WgradWinoParams wp) {
    __shared__ __attribute__((aligned(16))) float Ls[2 * WGW_BUF];         // 2 x 66 KB; the exchange image of the epilogue reuses it
    const WgradParams& p = wp.w;
    const mtd_wgrad_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int ncb = a.C / 64;
    const int nblk = blockIdx.y / ncb, cblk = blockIdx.y - nblk * ncb;
    const int n0 = nblk * 64, c0 = cblk * 64;
    const int second = (wp.ns_first > 0 && (int)blockIdx.x >= wp.ns_first) ? 1 : 0;
    const int zk = blockIdx.x - second * wp.ns_first;                       // slice within its image range
    const int tile_lo = second ? wp.first_tiles : 0;
    const int tile_hi = (wp.ns_first > 0 && !second) ? wp.first_tiles : wp.ntiles;
    const int ck_beg = zk * wp.chunks_per_split;
    const int nchunks_all = (tile_hi - tile_lo + WGW_T - 1) / WGW_T;
    const int ck_end = min(nchunks_all, ck_beg + wp.chunks_per_split);
    const int nck = ck_end - ck_beg;
    const int ck_last = ck_end - 1;

    // ---- transform roles: thread (tile t8 of the chunk, channel quad cq, quad position qp)
    const int qp = tid & 3, cq = (tid >> 2) & 15, t8 = tid >> 6;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q), (short)0, (int)p.q_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p), (short)0, (int)p.p_bytes, 0x00020000);
    const int qpx_b = a.q_ld * 4;
    struct Pre { f32x4 d[4]; f32x4 y, y2; };  // one chunk's operands of this thread, on their way from memory
    // (no second cotangent: the same loads with an out-of-range offset -- zeros, no memory traffic, one path through the loop)
    const __amdgpu_buffer_rsrc_t p2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp.p_add ? wp.p_add : a.p), (short)0, (int)p.p_bytes, 0x00020000);
    const bool has_p2 = wp.p_add != nullptr;
    f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
    // The wave's tile (t8 == wave: a chunk is eight tiles, one per wave) walks the slice in steps of WGW_T tiles; its (image, tile
    // row, tile column) live in scalar registers and a step adds (d_b, d_ty, d_tx) with carries -- no per-chunk divisions, and
    // of the address arithmetic only the lane's own part is vector work (every vector instruction is paid in full beside the
    // MFMAs: DESIGN 3.8).  load_next requests the operands of the cursor's chunk and moves the cursor on, until the slice's last
    // chunk, which it then repeats (the clamped loads of the loop's tail).
    const int tiles_y = wp.tiles_per_image / wp.tiles_x;
    const int d_tx = WGW_T % wp.tiles_x, d_ty = (WGW_T / wp.tiles_x) % tiles_y, d_b = WGW_T / wp.tiles_per_image;
    int cur_ck = ck_beg, cur_tg = tile_lo + ck_beg * WGW_T + wave;
    int cur_b = cur_tg / wp.tiles_per_image, cur_ty, cur_tx;
    {
        const int rr = cur_tg - cur_b * wp.tiles_per_image;
        cur_ty = rr / wp.tiles_x;
        cur_tx = rr - cur_ty * wp.tiles_x;
    }
    const unsigned u_lane = (unsigned)((qp * g.IW * a.q_ld + c0 + 4 * cq) * 4);
    const unsigned p_lane = (unsigned)((((qp >> 1) * g.OW + (qp & 1)) * a.p_ld + n0 + 4 * cq) * 4);
    // (round 5) the six requests of a chunk are issued ONE PER MFMA GROUP instead of as a burst at the top of the chunk: an in-order
    // wave whose load the address unit cannot accept yet issues no MFMA either, and all eight waves burst together after the
    // barrier (profiles/r5_load_spreading.txt).  prep_next computes the chunk's addresses and moves the cursor on; issue_next(j)
    // sends request j (0 .. 3: the patch row's pixels, 4: the cotangent, 5: the second cotangent).
    unsigned nx_uv = 0, nx_pv = 0x80000000u;
    bool nx_rowok = false;
    int nx_tx = 0;
    auto prep_next = [&]() {
        const bool tv = cur_tg < tile_hi;
        // U: patch row qp = image row 2 ty - 1 + qp, pixels 2 tx - 1 .. 2 tx + 2, channels c0 + 4 cq ..  (formed modulo 2^32 around
        // pixel 2 tx, which is inside the image whenever the row is: every VALID pixel's offset is in range AS the vector offset --
        // the range check does not see a scalar offset, so the column cannot travel there)
        const unsigned u_s = (((unsigned)cur_b * (unsigned)g.IH + (unsigned)(2 * cur_ty - 1)) * (unsigned)g.IW + (unsigned)(2 * cur_tx)) * (unsigned)a.q_ld * 4u;
        nx_uv = u_s + u_lane;
        nx_rowok = tv & ((unsigned)(2 * cur_ty - 1 + qp) < (unsigned)g.IH);
        nx_tx = cur_tx;
        // V: cotangent pixel (2 ty + (qp >> 1), 2 tx + (qp & 1)), channels n0 + 4 cq ..
        const unsigned p_s = tv ? (((unsigned)cur_b * (unsigned)g.OH + (unsigned)(2 * cur_ty)) * (unsigned)g.OW + (unsigned)(2 * cur_tx)) * (unsigned)a.p_ld * 4u : 0x80000000u;
        nx_pv = p_s + p_lane;
        // (scalar selects, no branch: one path through the loop)
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
            const bool colok = (unsigned)(2 * nx_tx - 1 + j) < (unsigned)g.IW;
            r.d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, (nx_rowok & colok) ? nx_uv + (unsigned)((j - 1) * qpx_b) : 0x80000000u, 0, 0));
        } else if (j == 4) {
            r.y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, nx_pv, 0, 0));
        } else {
            r.y2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p2rs, has_p2 ? nx_pv : 0x80000000u, 0, 0));
        }
    };
    auto load_next = [&](Pre& r) {
        prep_next();
#pragma unroll
        for (int j = 0; j < 6; ++j) issue_next(r, j);
    };
    const float usign = qp == 1 ? 1.f : -1.f;
    // A = [1 0; 1 1; 1 -1; 0 -1]: row a = qp of A dY A^T is  alpha * R0 + beta * R1,  R_i[b] = (dY A^T)[i][b]
    const float valpha = qp == 3 ? 0.f : 1.f, vbeta = qp == 0 ? 0.f : (qp == 1 ? 1.f : -1.f);
    // The transform of one chunk in six pieces (the K loop places them between groups of MFMAs): U columns 0 .. 3, then the
    // cotangent's quad gather, then its two store pairs.  `live` = 1 for a chunk of the slice, 0 for the clamped repeat past
    // its end (stored into a buffer nobody reads, not summed into the bias gradient).
    f32x4 vs0, vs1;
    auto tr_u = [&](float* Lb, const Pre& r, int j) {
        const f32x4 rj = j == 0 ? r.d[0] - r.d[2] : (j == 1 ? r.d[1] + r.d[2] : (j == 2 ? r.d[2] - r.d[1] : r.d[1] - r.d[3]));
        *reinterpret_cast<f32x4*>(Lb + t8 * 64 + 4 * cq + (4 * qp + j) * WGW_PL) = wgw_quad_rows(rj, usign);
    };
    // row a = qp of A dY A^T = [s0, s0 + s1, s0 - s1, -s1]  with  s_b = alpha dY[0][b] + beta dY[1][b]  (A combined down the tile's
    // rows first -- two DPP multiply-adds per value, no gather of the four pixels --, then along them)
    auto tr_v_gather = [&](const Pre& r, float live) {
        const f32x4 ys = r.y + r.y2;
        dbacc += live * ys;
        wgw_quad_pair(ys, valpha, vbeta, vs0, vs1);
    };
    auto tr_v_store = [&](float* Lb, int b0) {
        float* vo = Lb + 16 * WGW_PL + t8 * 64 + 4 * cq;
        if (b0 == 0) {
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 0) * WGW_PL) = vs0;
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 1) * WGW_PL) = vs0 + vs1;
        } else {
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 2) * WGW_PL) = vs0 - vs1;
            *reinterpret_cast<f32x4*>(vo + (4 * qp + 3) * WGW_PL) = -vs1;
        }
    };
    auto transform_store = [&](float* Lb, const Pre& r, float live) {
        tr_u(Lb, r, 0); tr_u(Lb, r, 1); tr_u(Lb, r, 2); tr_u(Lb, r, 3);
        tr_v_gather(r, live);
        tr_v_store(Lb, 0);
        tr_v_store(Lb, 2);
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
    // One chunk: the operands of chunk k + 2 are requested at the top (TWO chunks of MFMAs cover their way from HBM: two
    // register sets alternate), the 32 MFMAs of chunk k run with their fragments read from LDS one group ahead, and the
    // transform of chunk k + 1 (requested a chunk ago) is interleaved with the second half of the MFMAs.  One path through
    // the loop (indices clamped), so the compiler's wait counts are exact.
    auto one_chunk = [&](int k, const Pre& cur, Pre& nxt) {
        const float* Lc = Ls + (k & 1) * WGW_BUF;
        float* Ln = Ls + ((k + 1) & 1) * WGW_BUF;
        if (WGW_SPREAD) prep_next();                                       // (chunk k + 2, or the last one again)
        else load_next(nxt);
        __builtin_amdgcn_sched_barrier(0);
        float fa[2][2], fb[2][2];                                      // [ping-pong][half]
        auto frag = [&](int gi, int pp) {                             // group gi = 2 s + x
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
            if (gi + 1 < 8) frag(gi + 1, pp ^ 1);                      // the next group's fragments, under this group's MFMAs
            acc[x][0][0] = mfma32(fa[pp][0], fb[pp][0], acc[x][0][0]);
            acc[x][0][1] = mfma32(fa[pp][0], fb[pp][1], acc[x][0][1]);
            acc[x][1][0] = mfma32(fa[pp][1], fb[pp][0], acc[x][1][0]);
            acc[x][1][1] = mfma32(fa[pp][1], fb[pp][1], acc[x][1][1]);
            if (WGW_SPREAD && gi < 6) issue_next(nxt, gi);               // one request per MFMA group
            __builtin_amdgcn_sched_barrier(0);
            // the transform of chunk k + 1, a piece per group from the second group on (its operands were requested a chunk ago)
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

    // ---- epilogue: G^T dU G per (n, c), one 32 x 32 sub-block at a time through X[xi][n][c]; the bias gradient
    float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
    const int en_c = tid & 31, en_n = tid >> 5;                        // this thread's c and n (+16) inside the sub-block
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
                // t = G^T m (3 x 4), out = t G (3 x 3);  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
                float t[3][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float hs = 0.5f * (m[4 + b] + m[8 + b]), hd = 0.5f * (m[4 + b] - m[8 + b]);
                    t[0][b] = m[b] + hs;
                    t[1][b] = hd;
                    t[2][b] = hs + m[12 + b];
                }
                float* o = slab + ((long long)(n0 + 32 * h + nl)) * a.C + c0 + 32 * h2 + en_c;
                const long long tap_stride = (long long)a.N * a.C;
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
                    const float hs = 0.5f * (t[pr][1] + t[pr][2]), hd = 0.5f * (t[pr][1] - t[pr][2]);
                    o[(pr * 3 + 0) * tap_stride] = t[pr][0] + hs;
                    o[(pr * 3 + 1) * tap_stride] = hd;
                    o[(pr * 3 + 2) * tap_stride] = hs + t[pr][3];
                }
            }
            __syncthreads();
        }
    if (a.db && cblk == 0) {
        // bias gradient of the slice: the quad's four pixels, then the chunk's eight tiles (= the eight waves: t8 == wave)
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

// the kernel's domain
bool wgrad_wino_ok(const mtd_wgrad_args& a) {
    const mtd_geom& g = a.g;
    if (g.TH != 3 || g.TW != 3 || g.in_sy != 1 || g.in_sx != 1 || g.tap_dy != 1 || g.tap_dx != 1 || g.off_y != -1 || g.off_x != -1) return false;
    if (g.ky0 != 0 || g.kx0 != 0 || g.ky_step != 1 || g.kx_step != 1 || g.KW != 3) return false;
    if (g.IH != g.OH || g.IW != g.OW || (g.OH & 1) || (g.OW & 1)) return false;
    if ((a.N % 64) || (a.C % 64)) return false;
    if (!aligned16(a.p) || !aligned16(a.q) || (a.p_ld % 4) || (a.q_ld % 4)) return false;
    return true;
}

// (A Winograd F(2x4, 3x3) form of this kernel -- the transpose of wino_conv_kernel<., ., 6>, 24 positions on a 64 x 32 block --
// was built in round 4, measured slower twice (132-138 against 102 us per launch on the large layers, step 31.08 against 29.94 ms:
// 24 MFMAs per chunk and 32 channels to this kernel's 32 and 64, a heavier cotangent transform per MFMA) and removed in round 6;
// docs/LAB_NOTES_r3_r5.md has the numbers, the history has the kernel.)
constexpr int wgrad_wino_px(const mtd_wgrad_args&) { return 4; }
// (n, c) blocks and tiles of a layer in the form it takes
inline long long wgrad_wino_blocks(const mtd_wgrad_args& a) { return (long long)(a.N / 64) * (a.C / 64); }
inline long long wgrad_wino_tiles(const mtd_wgrad_args& a, long long images) {
    return images * (a.g.OH / 2) * (a.g.OW / 2);
}
