// F(2x4, 3x3) for the 32 -> 32 channel layers (arch/Ours/networks.py:95-164: the generator's encoder / decoder / block convs):
// persistent workgroups, the transformed weights in registers, every memory round trip a block ahead of its use.
// Included by conv_winograd.hip (inside its anonymous namespace, after wino_conv_kernel).
//
// Why a kernel of its own: with C = 32 a tile block (32 tiles = 256 output pixels) has TWO K steps of 16 channels.  In the
// general kernel a workgroup then spends three exposed memory round trips (first patch, second patch, the epilogue's operands)
// around 2 x 24 MFMAs per wave: 11 us per block on a CU that needs 2.6 us for its MFMAs (whole-slice inference, 8 x 512 x 512:
// 365 us per layer against an HBM floor of 100 - 170 us).  Here:
//   * one workgroup per CU walks a run of consecutive blocks of ITS XCD's share of the map (the halo rows of neighbouring
//     blocks meet in that XCD's L2);
//   * wave w holds positions 3 w .. 3 w + 2: their weights -- 3 x 32 x 32 floats = 48 registers -- are loaded ONCE, the
//     accumulators are another 48;
//   * half-steps alternate without a gap: while the MFMAs of half h read As[h] the transform of the NEXT half (h = 1 of this
//     block, then h = 0 of the next block) is written to the other buffer by the same waves, one MFMA : a few VALU operations,
//     as in the general kernel's K loop.  Patch registers are re-requested as soon as their transform has consumed them (two
//     phases = more than 5000 clocks ahead of their next use), the residual operand at the top of the block;
//   * every load and store is a buffer instruction issued unconditionally (tiles past the end: out-of-range offsets), so the
//     loop has one path and the compiler's vmcnt waits count exactly the younger requests (conv_winograd.hip, K loop comment);
//   * after the second half the 24 positions of a tile meet in LDS (X[xi][tile][32 n], 96 KB, aliasing As[1] and the spare
//     space behind it; As[0] already holds the next block's first half), thread (tile, channel quad, row) applies A^T . A and
//     the epilogue -- bias, optional residual before or after the ReLU (MTD_ACT_RELU_ADD), NONE / ReLU / LeakyReLU -- on 4 x 4
//     values and stores 16-byte vectors.
// LDS: As[0] 60 KB + max(As[1], X) 96 KB = 156 KB.  Roofline: fp32 MFMA at 2 M 32 32 3 executed flops; HBM at
// (in + out + residual) x 128 B per pixel -- the larger of the two on these layers.
#ifndef MTD_C32_SKIP
#define MTD_C32_SKIP 0      // lab (tools/c32_variants.sh): 1 no MFMAs, 2 no transforms, 4 no exchange / inverse transform, 8 no patch loads
#endif
constexpr int C32_AS = 24 * WT * WALD;          // floats per half-step image As[xi][tile][16 c (+4 pad)]
constexpr int C32_X = 24 * WT * 32;             // floats of the exchange image X[xi][tile][32 n]

typedef unsigned c32_u32x4 __attribute__((__vector_size__(16)));

struct C32Params {
    WinoParams wp;
    int nblocks;              // tile blocks of WT tiles
    float inv_tpi, inv_tx;    // 1 / tiles_per_image, 1 / tiles_x (tile index -> image, tile row, tile column without integer division)
    unsigned out_bytes, add_bytes;
};

// q = t / d, r = t % d for 0 <= t < 2^23 with inv = 1.0f / d (one correction step either way)
__device__ __forceinline__ void c32_divmod(int t, int d, float inv, int& q, int& r) {
    q = (int)((float)t * inv);
    r = t - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

template <bool HAS_ADD>
__global__ __launch_bounds__(512) void wino_c32_kernel(const C32Params cp) {
    constexpr int PX = 6, TWX = 4;
    __shared__ __attribute__((aligned(16))) float Ls[C32_AS + C32_X];
    float* const As0 = Ls;
    float* const As1 = Ls + C32_AS;
    float* const Xs = Ls + C32_AS;
    const WinoParams& wp = cp.wp;
    const mtd_conv_args& a = wp.p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;

    // this workgroup's blocks: XCD x (= workgroup id mod 8: consecutive workgroups go round-robin over the XCDs) owns the
    // contiguous run [x per, (x + 1) per) of tile blocks; its workgroups take every `slots`-th block of the run, so at any time
    // the XCD works on `slots` consecutive blocks
    const int slots = (int)gridDim.x >> 3;
    const int per = (cp.nblocks + 7) >> 3;
    const int b_end = min(cp.nblocks, ((int)(blockIdx.x & 7) + 1) * per);
    int blk = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (blk >= b_end) return;

    // ---- roles.  Transform: thread (tile tt, channel quad tq, patch row ti); epilogue: thread (tile tt, channel quad enq, output row ei)
    const int ti = tid & 3, tq = (tid >> 2) & 3, tt = tid >> 4;
    const int ei = tid & 1, enq = (tid >> 1) & 7;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)wp.p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)cp.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t e1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_ADD ? a.add1 : a.in), (short)0, (int)(HAS_ADD ? cp.add_bytes : 0u), 0x00020000);
    const int px_b = a.in_ld * 4;
    struct Loc { unsigned pbase, pvalid; int epix; };
    auto locate = [&](int b_) -> Loc {
        Loc L;
        const int tg = b_ * WT + tt;
        const bool tv = (b_ < b_end) & (tg < wp.ntiles);
        int img, r, ty, tx;
        c32_divmod(tg, wp.tiles_per_image, cp.inv_tpi, img, r);
        c32_divmod(r, wp.tiles_x, cp.inv_tx, ty, tx);
        const int iy = 2 * ty - 1 + ti;
        // (pixel (iy, 4 tx - 1) may lie outside the image: the offset is formed modulo 2^32, every VALID pixel's is in range)
        L.pbase = (unsigned)(((((long long)img * g.IH + iy) * g.IW + (TWX * tx - 1)) * a.in_ld + 4 * tq) * 4);
        L.pvalid = 0;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int ix = TWX * tx - 1 + j;
            if (tv & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) L.pvalid |= 1u << j;
        }
        L.epix = tv ? (img * g.OH + 2 * ty + ei) * g.OW + TWX * tx : -1;
        return L;
    };
    auto load_patch = [&](f32x4 (&d)[PX], const Loc& L, int h) {        // half h: channels 16 h + 4 tq .. + 3 of the row's six pixels
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const unsigned vo = ((L.pvalid >> j) & 1u) ? L.pbase + (unsigned)(j * px_b) : 0x80000000u;
            if constexpr (!(MTD_C32_SKIP & 8)) d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, h * 64, 0));
            else asm volatile("" :: "v"(vo));
        }
    };
    const float qsign = ti == 1 ? 1.f : -1.f;
    auto quad_get = [&](float v, int ctrl) {
        const int x = __builtin_bit_cast(int, v);
        return __builtin_bit_cast(float, ctrl == 0 ? __builtin_amdgcn_update_dpp(0, x, 0x64, 0xf, 0xf, true)      // lanes [0, 1, 2, 1]
                                                  : __builtin_amdgcn_update_dpp(0, x, 0xDA, 0xf, 0xf, true));    // lanes [2, 2, 1, 3]
    };
    // B^T d B of the thread's row -> As[xi = 6 ti + j][tt][4 tq ..], columns j0, j0 + 1 (F(4,3) along the row, F(2,3) across the quad)
    auto transform_cols = [&](float* As, const f32x4 (&d)[PX], int j0) {
        float* o = As + tt * WALD + 4 * tq;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
            f32x4 rj;
            if (j == 0) rj = 4.f * d[0] - 5.f * d[2] + d[4];
            else if (j == 1) rj = (d[4] - 4.f * d[2]) + (d[3] - 4.f * d[1]);
            else if (j == 2) rj = (d[4] - 4.f * d[2]) - (d[3] - 4.f * d[1]);
            else if (j == 3) rj = (d[4] - d[2]) + 2.f * (d[3] - d[1]);
            else if (j == 4) rj = (d[4] - d[2]) - 2.f * (d[3] - d[1]);
            else rj = 4.f * d[1] - 5.f * d[3] + d[5];
            f32x4 u;
#pragma unroll
            for (int c = 0; c < 4; ++c) u[c] = fmaf(qsign, quad_get(rj[c], 1), quad_get(rj[c], 0));
            *reinterpret_cast<f32x4*>(o + (PX * ti + j) * (WT * WALD)) = u;
        }
    };

    // ---- MFMA role: positions 3 wave .. 3 wave + 2; all their weights in registers: wr[x][ck] = channels 8 ck + 4 kh .. + 3 of n = l31
    f32x4 wr[3][4];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int ck = 0; ck < 4; ++ck)
            wr[x][ck] = *reinterpret_cast<const f32x4*>(a.w + ((long long)((3 * wave + x) * 4 + ck) * 32 + l31) * 8 + kh * 4);
    f32x16 acc[3];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * enq);
    // the activation as two slopes for the non-positive side (no branches in the epilogue): before the residual (0: ReLU of
    // MTD_ACT_RELU_ADD, else 1) and after it (ReLU 0, LeakyReLU 0.2, else 1)
    const float slope_pre = a.act == MTD_ACT_RELU_ADD ? 0.f : 1.f;
    const float slope_post = a.act == MTD_ACT_RELU ? 0.f : (a.act == MTD_ACT_LRELU ? 0.2f : 1.f);
    const float esign = ei ? -1.f : 1.f;
    const int m03_off = (ei ? 3 : 0) * PX * WT * 32;

    // One half-step: the MFMAs of half H of the current block (A fragments from Ac) with the transform of the patch rows in d
    // (the next half) into An in the same scheduling region, then the workgroup barrier.
    auto half_step = [&](auto hc, const float* Ac, float* An, const f32x4 (&d)[PX]) {
        constexpr int H = decltype(hc)::value;
        f32x4 af0[3], af1[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            af0[x] = *reinterpret_cast<const f32x4*>(Ac + ((3 * wave + x) * WT + l31) * WALD + kh * 4);
            af1[x] = *reinterpret_cast<const f32x4*>(Ac + ((3 * wave + x) * WT + l31) * WALD + 8 + kh * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(MTD_C32_SKIP & 1)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int x = 0; x < 3; ++x) acc[x] = mfma32(af0[x][s], wr[x][2 * H][s], acc[x]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int x = 0; x < 3; ++x) acc[x] = mfma32(af1[x][s], wr[x][2 * H + 1][s], acc[x]);
        } else {
#pragma unroll
            for (int x = 0; x < 3; ++x) asm volatile("" :: "v"(af0[x]), "v"(af1[x]));
        }
        if constexpr (!(MTD_C32_SKIP & 2)) {
            transform_cols(An, d, 0);
            transform_cols(An, d, 2);
            transform_cols(An, d, 4);
        } else {
#pragma unroll
            for (int j = 0; j < PX; ++j) asm volatile("" :: "v"(d[j]));
        }
        // 24 MFMAs over ~150 vector-ALU / DPP operations and six 16-byte LDS stores
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            if ((i & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };

    // ---- prologue: block 0's first half transformed into As0, its second half and the next block's first half in flight
    Loc L0 = locate(blk), L1 = locate(blk + slots);
    f32x4 dA[PX], dB[PX];
    load_patch(dA, L0, 0);
    load_patch(dB, L0, 1);
    transform_cols(As0, dA, 0);
    transform_cols(As0, dA, 2);
    transform_cols(As0, dA, 4);
    __builtin_amdgcn_sched_barrier(0);
    load_patch(dA, L1, 0);
    __syncthreads();

#pragma unroll 1
    for (; blk < b_end; blk += slots) {
        // residual operand of THIS block's outputs: two phases ahead of the epilogue
        f32x4 e1[TWX];
#pragma unroll
        for (int q = 0; q < TWX; ++q) {
            if constexpr (HAS_ADD) {
                const unsigned vo = L0.epix >= 0 ? (unsigned)(((long long)(L0.epix + q) * a.add1_ld + 4 * enq) * 4) : 0x80000000u;
                e1[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(e1rs, vo, 0, 0));
            } else {
                e1[q] = f32x4{-0.0f, -0.0f, -0.0f, -0.0f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        half_step(std::integral_constant<int, 0>{}, As0, As1, dB);           // MFMAs of half 0; half 1 of this block -> As1
        load_patch(dB, L1, 1);                                                // (dB consumed: the next block's second half)
        __builtin_amdgcn_sched_barrier(0);
        half_step(std::integral_constant<int, 1>{}, As1, As0, dA);           // MFMAs of half 1; half 0 of the next block -> As0
        const Loc L2 = locate(blk + 2 * slots);
        load_patch(dA, L2, 0);                                                // (dA consumed: first half of the block after the next)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 y[TWX];
        if constexpr (!(MTD_C32_SKIP & 4)) {
        // ---- the 24 positions of a tile meet in X (aliases As1: free since the barrier that closed half 1)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float* X = Xs + (3 * wave + x) * (WT * 32) + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                X[mfma32_row(e, lane) * 32] = acc[x][e];
                acc[x][e] = 0.f;
            }
        }
        __syncthreads();
        {
            f32x4 t[PX];
#pragma unroll
            for (int b = 0; b < PX; ++b) {
                const float* col = Xs + (b * WT + tt) * 32 + 4 * enq;                  // position xi = 6 a + b at col + a * 6 * WT * 32
                const f32x4 m1 = *reinterpret_cast<const f32x4*>(col + 1 * PX * WT * 32);
                const f32x4 m2 = *reinterpret_cast<const f32x4*>(col + 2 * PX * WT * 32);
                const f32x4 m03 = *reinterpret_cast<const f32x4*>(col + m03_off);
                t[b] = m1 + esign * (m2 + m03);                                        // row ei of A^T m: m0 + m1 + m2 | m1 - m2 - m3
            }
            // A^T of F(4,3) = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
            const f32x4 s12 = t[1] + t[2], d12 = t[1] - t[2], s34 = t[3] + t[4], d34 = t[3] - t[4];
            y[0] = t[0] + s12 + s34;
            y[1] = d12 + 2.f * d34;
            y[2] = s12 + 4.f * s34;
            y[3] = d12 + 8.f * d34 + t[5];
        }
        } else {
#pragma unroll
            for (int q = 0; q < TWX; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) { y[q][c] = acc[0][4 * q + c] + acc[1][4 * q + c] + acc[2][4 * q + c]; acc[0][4 * q + c] = acc[1][4 * q + c] = acc[2][4 * q + c] = 0.f; }
        }
#pragma unroll
        for (int q = 0; q < TWX; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = y[q][c] + bias4[c];
                v = v > 0.f ? v : v * slope_pre;                                       // MTD_ACT_RELU_ADD: the residual AFTER the activation
                v += e1[q][c];
                v = v > 0.f ? v : v * slope_post;
                y[q][c] = v;
            }
            const unsigned vo = L0.epix >= 0 ? (unsigned)(((long long)(L0.epix + q) * a.out_ld + 4 * enq) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c32_u32x4, y[q]), ors, vo, 0, 0);
        }
        __syncthreads();                                                               // X read: the next half-step writes As1
        L0 = L1;
        L1 = L2;
    }
}
