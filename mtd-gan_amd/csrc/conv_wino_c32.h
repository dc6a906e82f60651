// F(2x4, 3x3) for the 32 -> 32 channel layers (arch/Ours/networks.py:95-164: the generator's encoder / decoder / block convs):
// persistent workgroups, the transformed weights in registers, every memory round trip two blocks ahead of its use, and the
// three kinds of work of a block -- MFMAs, input transform, output transform + epilogue -- of three DIFFERENT blocks in one
// scheduling region.  Included by conv_winograd.hip (inside its anonymous namespace, after wino_conv_kernel).
//
// Why a kernel of its own: with C = 32 a tile block has two K steps of 16 channels.  In the general kernel a workgroup then
// spends three exposed memory round trips (first patch, second patch, the epilogue's operands) around 2 x 24 MFMAs per wave:
// 11 us per 256-pixel block on a CU that needs 2.6 us for its MFMAs (whole-slice inference, 8 x 512 x 512: 365 us per layer
// against an HBM floor of 90 - 140 us).  A first persistent form with 32-tile blocks (round 4, 260 us) still ran its phases
// one after the other -- MFMA pipe busy 36 % of the time, 50 % of the LDS cycles bank conflicts of the transform's stores,
// the exchange image aliasing the input image (profiles/r4_c32_winograd_parts.txt).  This form:
//   * a block is 16 tiles (128 output pixels, v_mfma_f32_16x16x4_f32): the input image of a block (both halves of K, 24
//     positions) is 50 KB, so TWO of them and the exchange image X (49 KB) fit in LDS side by side and nothing aliases;
//   * one workgroup per CU walks every `slots`-th block of ITS XCD's contiguous run of blocks (the halo rows of neighbouring
//     blocks meet in that XCD's L2);
//   * wave w holds positions 3 w .. 3 w + 2 for both 16-channel output blocks: their weights are 48 registers, loaded ONCE; the
//     MFMAs are issued weights-as-A, so a lane ends up with FOUR CONSECUTIVE output channels of its own tile and the exchange
//     stores are 16-byte vectors;
//   * iteration i of the walk:  the 48 MFMAs of block i (A fragments from As[i & 1])  |  the input transform of block i + 1 into
//     As[(i + 1) & 1] (thread = (tile, half of K, channel quad, patch row): six 16-byte loads requested two iterations earlier --
//     the eight lanes of a pixel cover one whole 128-byte line --, F(4,3) along the row in registers (row_stage; the patch
//     registers are dead after it and their next request, for block i + 3, goes out at once), F(2,3) across the quad by DPP and
//     six 16-byte LDS stores (quad_stage))  |  the output transform + epilogue of block i - 1 from X (thread = (tile, channel pair,
//     output row): 18 8-byte LDS loads, A^T . A, bias, residual before or after the ReLU (MTD_ACT_RELU_ADD), NONE / ReLU /
//     LeakyReLU, four 8-byte stores; the residual operand was requested an iteration earlier) -- in four scheduling regions of
//     12 MFMAs, each with its share of the vector-ALU work, LDS accesses and memory instructions.  Then a barrier, the
//     accumulators of block i go to X (six 16-byte stores per lane), another barrier;
//   * every global load and store is a buffer instruction issued unconditionally (tiles past the end: out-of-range offsets), so
//     the loop has one path and the compiler's vmcnt waits count exactly the younger requests (conv_winograd.hip, K loop);
//   * LDS layouts without padding rows: As[h][xi][tile][16 c] with 268 floats between positions (6 x 268 = 8 mod 32: the four
//     patch rows of a quad store to four different bank groups) and the channel quad XOR-swizzled by the tile group (the
//     16-lane groups of a 16-byte LDS load then hit 16 different 4-bank slots); X[a][b][tile][32 n] with the quad swizzled by
//     the tile and 32 extra floats per patch row a (rows 0 and 3, read by the two output rows of a tile, on opposite halves
//     of the 64 banks).
// Roofline: HBM at (in + out + residual) x 128 B per pixel; fp32 MFMA at 2 M 32 32 3 executed flops is the smaller bound on
// these layers.  Measured on 8 x 512 x 512 (profiles/r4_c32_winograd_parts.txt): 235 - 260 us per layer (223 - 240 after the vector-ALU
// trimming of DESIGN 9.2; 201 inside the inference step) against 360
// (general kernel) and 400 - 440 (implicit GEMM); fabric traffic exactly the algorithmic bytes (FETCH_SIZE x 2 = in + residual),
// no LDS bank conflicts, matrix pipe busy 41 %.  With parts switched off (MTD_C32_SKIP): everything but the patch loads 165 us,
// loads + stores alone 135 - 160 us, MFMAs + barriers alone 140 us -- the vector-memory path (about 19 clocks per instruction
// plus 4 per 128-byte line: 4700 clocks per block for 48 + 32 + 32 instructions) and the matrix pipe (3072 clocks per block) are
// both near their limits and overlap only in part; neither the block shape (4 x 4 tiles instead of 16 x 1: a third fewer L2
// requests) nor L2-resident inputs change the time.
#ifndef MTD_C32_SKIP
#define MTD_C32_SKIP 0      // lab (tools/c32_variants.sh): 1 no MFMAs, 2 no input transform, 4 no output transform, 8 no patch loads
#endif
constexpr int C32_T = 16;                         // tiles per block
constexpr int C32_PS = 268;                       // floats between positions of As
constexpr int C32_AS = 2 * 24 * C32_PS;           // floats of one block's input image [h][xi][tile][16 c]
constexpr int C32_XA = 6 * C32_T * 32 + 32;       // floats between the patch rows of X
constexpr int C32_X = 4 * C32_XA;

typedef unsigned c32_u32x4 __attribute__((__vector_size__(16)));
typedef unsigned c32_u32x2 __attribute__((__vector_size__(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct C32Params {
    WinoParams wp;
    int nblocks;              // blocks of C32_T tiles
    int tiles_y;              // tile rows per image
    int d_img, d_ty, d_tx;    // one step of a workgroup's walk (C32_T * slots tiles) as (images, tile rows, tile columns)
    unsigned out_bytes, add_bytes;
    unsigned out2_bytes, mask_bytes;      // MASKED2 form: extents of the second output (0: none) and of the mask operand
};

// MASKED2 (round 6: the data-gradient role of the generator's plain 32 -> 32 layers, generator_path.generator_backward): the epilogue's value
// v goes to a.out2 (where given) and v * (mask > 0 ? 1 : mask_slope) to a.out -- the cotangent and its masked form for the block
// that consumes it, as the halo-tile kernel writes them (conv_igemm.hip).  Four more 8-byte loads and stores per thread and block.
template <bool HAS_ADD, bool MASKED2 = false>
__global__ __launch_bounds__(512) void wino_c32_kernel(const C32Params cp) {
    constexpr int PX = 6, TWX = 4;
    __shared__ __attribute__((aligned(16))) float Ls[2 * C32_AS + C32_X];
    float* const Xs = Ls + 2 * C32_AS;
    const WinoParams& wp = cp.wp;
    const mtd_conv_args& a = wp.p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // this workgroup's blocks: XCD x (= workgroup id mod 8: consecutive workgroups go round-robin over the XCDs) owns the
    // contiguous run [x per, (x + 1) per) of blocks; its workgroups take every `slots`-th block of the run, so at any time the
    // XCD works on `slots` consecutive blocks
    const int slots = (int)gridDim.x >> 3;
    const int per = (cp.nblocks + 7) >> 3;
    const int b_end = min(cp.nblocks, ((int)(blockIdx.x & 7) + 1) * per);
    const int blk0 = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (blk0 >= b_end) return;
    const int nit = (b_end - blk0 + slots - 1) / slots;          // blocks of this workgroup

    // ---- roles
    // transform: patch row, channel quad, half of K, tile.  (Quad and half are the low bits after the row: the eight lanes of a
    // patch-row pixel cover one whole 128-byte line, a load instruction touches 8 lines -- with the half in the wave index it
    // touched 16 half lines and the L1 / address path, at ~2.7 clocks per line, was as busy as the matrix pipe.)
    const int ti = tid & 3, tq = (tid >> 2) & 3, th = (tid >> 4) & 1, tt = tid >> 5;
    const int ei = tid & 1, ecp = (tid >> 1) & 15, et = tid >> 5;                            // epilogue: output row, channel pair, tile
    const int lt = lane & 15, kq = lane >> 4;                                                // MFMA: tile / channel of the fragment, k index
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)wp.p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)cp.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t e1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_ADD ? a.add1 : a.in), (short)0, (int)(HAS_ADD ? cp.add_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASKED2 ? a.mask : a.in), (short)0, (int)(MASKED2 ? cp.mask_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t o2rs = __builtin_amdgcn_make_buffer_rsrc((MASKED2 && a.out2) ? a.out2 : a.out, (short)0, (int)((MASKED2 && a.out2) ? cp.out2_bytes : 0u), 0x00020000);
    const float mslope = a.mask_slope;
    const unsigned px_b = (unsigned)a.in_ld * 4u;

    // a thread's tile as (image, tile row, tile column); one step of the walk adds (d_img, d_ty, d_tx) with carries
    struct Cur { int img, ty, tx; };
    auto cur_init = [&](int tile) {
        Cur c;
        c.img = tile / wp.tiles_per_image;
        const int r = tile - c.img * wp.tiles_per_image;
        c.ty = r / wp.tiles_x;
        c.tx = r - c.ty * wp.tiles_x;
        return c;
    };
    auto cur_step = [&](Cur& c) {
        c.tx += cp.d_tx;
        const bool c1 = c.tx >= wp.tiles_x;
        c.tx -= c1 ? wp.tiles_x : 0;
        c.ty += cp.d_ty + (c1 ? 1 : 0);
        const int c2 = c.ty >= cp.tiles_y ? 1 : 0;
        c.ty -= c2 ? cp.tiles_y : 0;
        c.img += cp.d_img + c2;
    };
    // patch row ti of the tile at c, channels 16 th + 4 tq ..: byte offset of its pixel 0 (formed modulo 2^32: every VALID
    // pixel's offset is in range) and which of its six pixels lie inside the image
    auto load_patch = [&](f32x4 (&d)[PX], const Cur& c, bool live, int j0, int j1) {
        const int iy = 2 * c.ty - 1 + ti;
        const unsigned p_img = (unsigned)c.img, p_iy = (unsigned)iy;
        const unsigned pbase = (((p_img * (unsigned)g.IH + p_iy) * (unsigned)g.IW + (unsigned)(TWX * c.tx - 1)) * (unsigned)a.in_ld + (unsigned)(16 * th + 4 * tq)) * 4u;
        const bool rv = live & (c.img < g.B) & ((unsigned)iy < (unsigned)g.IH);
        const unsigned pvalid = rv ? (0x3Fu & ~(c.tx == 0 ? 1u : 0u) & ~(c.tx == wp.tiles_x - 1 ? 0x20u : 0u)) : 0u;
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const unsigned vo = (pbase + (unsigned)j * px_b) | (((pvalid >> j) & 1u) ? 0u : 0x80000000u);      // (invalid: out of range)
            if constexpr (!(MTD_C32_SKIP & 8)) d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, 0, 0));
            else asm volatile("" :: "v"(vo));
        }
    };
    // left pixel of output row ei of the tile at c (-1: no such tile)
    auto out_pixel_of = [&](const Cur& c, bool live) {
        return ((c.img * g.OH + 2 * c.ty + ei) * g.OW + TWX * c.tx) | ((live & (c.img < g.B)) ? 0 : -1);
    };

    // ---- input transform: B^T d B of the thread's row -> As[th][xi = 6 ti + j][tt][4 (tq ^ swizzle) ..]
    const float qsign = ti == 1 ? 1.f : -1.f;
    const int t_off = (th * 24 + 6 * ti) * C32_PS + tt * 16 + 4 * (tq ^ ((0x78 >> (2 * (tt >> 2))) & 3));
    // Two stages.  row_stage: F(4,3) along the thread's row, r = d B (after it the patch registers are dead: their next request
    // goes out right away and the whole iteration covers its way from memory).  quad_stage: F(2,3) across the quad of lanes that
    // hold the four rows, columns j0 .. j1 - 1, and the stores.
    auto row_stage = [&](f32x4 (&r)[PX], const f32x4 (&d)[PX]) {
        if constexpr ((MTD_C32_SKIP & 2) != 0) {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                asm volatile("" :: "v"(d[j]));
                r[j] = f32x4{1.f, 2.f, 3.f, 4.f};
            }
        } else {
            // F(4,3):  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
            const f32x4 p = d[4] - 4.f * d[2], q = d[3] - 4.f * d[1], u = d[4] - d[2], v = d[3] - d[1];
            r[0] = 4.f * d[0] - 5.f * d[2] + d[4];
            r[1] = p + q;
            r[2] = p - q;
            r[3] = u + 2.f * v;
            r[4] = u - 2.f * v;
            r[5] = 4.f * d[1] - 5.f * d[3] + d[5];
        }
    };
    auto quad_stage = [&](float* As, const f32x4 (&r)[PX], int j0, int j1) {
        if constexpr ((MTD_C32_SKIP & 2) == 0) {
#pragma unroll
            for (int j = j0; j < j1; ++j) {
                *reinterpret_cast<f32x4*>(As + t_off + j * C32_PS) = wino_quad_rows(r[j], qsign);
            }
        } else {
#pragma unroll
            for (int j = j0; j < j1; ++j) asm volatile("" :: "v"(r[j]));
        }
    };

    // ---- MFMA role: positions 3 wave .. 3 wave + 2, both output-channel blocks; weights as the A operand:
    // wr[x][h][nb] = W(n = 16 nb + lt, channels 16 h + 4 kq .. + 3) at position 3 wave + x  (layout [xi][C/8][N][8])
    f32x4 wr[3][2][2];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                wr[x][h][nb] = *reinterpret_cast<const f32x4*>(a.w + ((long long)(((3 * wave + x) * 4 + 2 * h + (kq >> 1)) * 32 + 16 * nb + lt)) * 8 + (kq & 1) * 4);
    f32x4 acc[3][2];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[x][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_off = (3 * wave) * C32_PS + lt * 16 + 4 * (kq ^ ((0x78 >> (2 * (lt >> 2))) & 3));
    // the accumulators' place in X: position xi = 6 a + b at a C32_XA + b 512, tile row of 32 floats, quad (4 nb + kq) ^ (lt & 7)
    int x_off[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        const int pos = 3 * wave + x;
        x_off[x] = (pos / 6) * C32_XA + (pos % 6) * (C32_T * 32) + lt * 32;
    }

    // ---- epilogue role
    f32x2 bias2 = {0.f, 0.f};
    if (a.bias) bias2 = *reinterpret_cast<const f32x2*>(a.bias + 2 * ecp);
    // the activation as two slopes for the non-positive side (no branches in the epilogue): before the residual (0: the ReLU of
    // MTD_ACT_RELU_ADD, else 1) and after it (ReLU 0, LeakyReLU 0.2, else 1)
    const float slope_pre = a.act == MTD_ACT_RELU_ADD ? 0.f : 1.f;
    const float slope_post = a.act == MTD_ACT_RELU ? 0.f : (a.act == MTD_ACT_LRELU ? 0.2f : 1.f);
    const float esign = ei ? -1.f : 1.f;
    const int e_off = et * 32 + 4 * ((ecp >> 1) ^ (et & 7)) + 2 * (ecp & 1);
    const int m03_off = (ei ? 3 : 0) * C32_XA;
    auto load_residual = [&](f32x2 (&e1)[TWX], int epix) {
#pragma unroll
        for (int q = 0; q < TWX; ++q) {
            if constexpr (HAS_ADD) {
                const unsigned vo = (((unsigned)(epix + q) * (unsigned)a.add1_ld + 2u * (unsigned)ecp) * 4u) | ((unsigned)(epix >> 31) & 0x80000000u);
                e1[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(e1rs, vo, 0, 0));
            } else {
                e1[q] = f32x2{-0.0f, -0.0f};
            }
        }
    };
    auto load_mask = [&](f32x2 (&m)[TWX], int epix) {
#pragma unroll
        for (int q = 0; q < TWX; ++q) {
            if constexpr (MASKED2) {
                const unsigned vo = (((unsigned)(epix + q) * (unsigned)a.mask_ld + 2u * (unsigned)ecp) * 4u) | ((unsigned)(epix >> 31) & 0x80000000u);
                m[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(mrs, vo, 0, 0));
            } else {
                m[q] = f32x2{1.f, 1.f};
            }
        }
    };
    // output transform of the block in X (epi_read: the 18 LDS loads and row ei of A^T m per column), then A along the row, the
    // epilogue and the stores at pixel epix .. epix + 3 (epi_finish; epix < 0: nothing is stored)
    auto epi_read = [&](f32x2 (&t)[PX]) {
#pragma unroll
        for (int b = 0; b < PX; ++b) {
            if constexpr (!(MTD_C32_SKIP & 4)) {
                const float* col = Xs + b * (C32_T * 32) + e_off;
                const f32x2 m1 = *reinterpret_cast<const f32x2*>(col + 1 * C32_XA);
                const f32x2 m2 = *reinterpret_cast<const f32x2*>(col + 2 * C32_XA);
                const f32x2 m03 = *reinterpret_cast<const f32x2*>(col + m03_off);
                t[b] = m1 + esign * (m2 + m03);                                        // m0 + m1 + m2 | m1 - m2 - m3
            } else {
                t[b] = f32x2{(float)b, 1.f};
            }
        }
    };
    auto epi_finish = [&](const f32x2 (&t)[PX], const f32x2 (&e1)[TWX], const f32x2 (&mk)[TWX], int epix) {
        // A^T of F(4,3) = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
        f32x2 y[TWX];
        const f32x2 s12 = t[1] + t[2], d12 = t[1] - t[2], s34 = t[3] + t[4], d34 = t[3] - t[4];
        y[0] = t[0] + s12 + s34;
        y[1] = d12 + 2.f * d34;
        y[2] = s12 + 4.f * s34;
        y[3] = d12 + 8.f * d34 + t[5];
#pragma unroll
        for (int q = 0; q < TWX; ++q) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // (slopes in [0, 1]: max(v, slope v) is v for v > 0 and slope v below -- two instructions per activation)
                float v = y[q][c] + bias2[c];
                v = fmaxf(v, v * slope_pre);                                           // MTD_ACT_RELU_ADD: the residual AFTER the activation
                v += e1[q][c];
                y[q][c] = fmaxf(v, v * slope_post);
            }
            if constexpr (MASKED2) {
                const unsigned vo2 = (((unsigned)(epix + q) * (unsigned)a.out2_ld + 2u * (unsigned)ecp) * 4u) | ((unsigned)(epix >> 31) & 0x80000000u);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(c32_u32x2, y[q]), o2rs, vo2, 0, 0);      // (no second output: extent 0)
#pragma unroll
                for (int c = 0; c < 2; ++c) y[q][c] *= mk[q][c] > 0.f ? 1.f : mslope;
            }
            const unsigned vo = (((unsigned)(epix + q) * (unsigned)a.out_ld + 2u * (unsigned)ecp) * 4u) | ((unsigned)(epix >> 31) & 0x80000000u);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(c32_u32x2, y[q]), ors, vo, 0, 0);
        }
    };

    // ---- prologue: block 0 transformed into As[0]; blocks 1 and 2 in flight
    Cur curT = cur_init(blk0 * C32_T + tt), curE = cur_init(blk0 * C32_T + et);
    int ld_it = 0;                                       // walk index of the next patch request
    f32x4 D0[PX], D1[PX];
    f32x2 E0[TWX], E1[TWX];
    f32x2 M0[TWX], M1[TWX];
    load_patch(D0, curT, ld_it < nit, 0, PX); cur_step(curT); ++ld_it;
    load_patch(D1, curT, ld_it < nit, 0, PX); cur_step(curT); ++ld_it;
    {
        f32x4 r0[PX];
        row_stage(r0, D0);
        quad_stage(Ls, r0, 0, PX);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_patch(D0, curT, ld_it < nit, 0, PX); cur_step(curT); ++ld_it;
    load_residual(E1, -1);
    load_mask(M1, -1);
    int epix_prev = -1;
    __syncthreads();

    // One iteration (parity P = it & 1): see the header.  Dn = the patch rows of block it + 1 (then re-requested for it + 3);
    // Ec receives this block's residual operand, Ep holds the previous block's.
    auto body = [&](auto pc, int it, f32x4 (&Dn)[PX], f32x2 (&Ec)[TWX], const f32x2 (&Ep)[TWX], f32x2 (&Mc)[TWX], const f32x2 (&Mp)[TWX]) {
        constexpr int P = decltype(pc)::value;
        const float* Ac = Ls + P * C32_AS;
        float* An = Ls + (P ^ 1) * C32_AS;
        f32x4 af[3][2];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int h = 0; h < 2; ++h) af[x][h] = *reinterpret_cast<const f32x4*>(Ac + a_off + (h * 24 + x) * C32_PS);
        __builtin_amdgcn_sched_barrier(0);
        // Four scheduling regions of 12 MFMAs (32 clocks of the pipe each; two waves share it) with a quarter of the other work each:
        // one MFMA : six or seven vector-ALU operations, the LDS and memory instructions in between
        auto mfmas = [&](auto hc, int s0) {
            constexpr int H = decltype(hc)::value;
            if constexpr (!(MTD_C32_SKIP & 1)) {
#pragma unroll
                for (int s = s0; s < s0 + 2; ++s)
#pragma unroll
                    for (int x = 0; x < 3; ++x)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[x][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[x][H][nb][s], af[x][H][s], acc[x][nb], 0, 0, 0);
            } else {
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const f32x4 keep = af[x][H];
                    asm volatile("" :: "v"(keep));
                }
            }
        };
        auto interleave = [&](auto valu_c, auto ds_c, auto vm_c) {
            constexpr int VALU = decltype(valu_c)::value, DS_EVERY = decltype(ds_c)::value, VM_EVERY = decltype(vm_c)::value;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
                if (DS_EVERY && (i % (DS_EVERY ? DS_EVERY : 1)) == DS_EVERY - 1) __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
                if (VM_EVERY && (i % (VM_EVERY ? VM_EVERY : 1)) == VM_EVERY - 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I6 = std::integral_constant<int, 6>;
        f32x4 r[PX];
        mfmas(std::integral_constant<int, 0>{}, 0);
        row_stage(r, Dn);
        load_patch(Dn, curT, ld_it < nit, 0, 3);
        quad_stage(An, r, 0, 2);
        interleave(I6{}, I6{}, I4{});
        mfmas(std::integral_constant<int, 0>{}, 2);
        load_patch(Dn, curT, ld_it < nit, 3, PX);
        cur_step(curT);
        ++ld_it;
        quad_stage(An, r, 2, 5);
        interleave(I6{}, I4{}, I4{});
        mfmas(std::integral_constant<int, 1>{}, 0);
        quad_stage(An, r, 5, PX);
        const int epix_cur = out_pixel_of(curE, it < nit);
        cur_step(curE);
        load_residual(Ec, epix_cur);
        load_mask(Mc, epix_cur);
        f32x2 et_[PX];
        epi_read(et_);
        using IVM = std::integral_constant<int, MASKED2 ? 1 : 3>;      // eight instead of four memory instructions in each of the last two regions
        interleave(I6{}, I1{}, IVM{});
        mfmas(std::integral_constant<int, 1>{}, 2);
        epi_finish(et_, Ep, Mp, epix_prev);
        interleave(I6{}, I0{}, IVM{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // the accumulators of this block -> X (read by the next iteration's epilogue)
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                *reinterpret_cast<f32x4*>(Xs + x_off[x] + 4 * ((4 * nb + kq) ^ (lt & 7))) = acc[x][nb];
                acc[x][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        __syncthreads();
        epix_prev = epix_cur;
    };

    // (an odd count runs one more iteration on a block past the end: its loads and stores are out of range)
#pragma unroll 1
    for (int it = 0; it < nit; it += 2) {
        body(std::integral_constant<int, 0>{}, it, D1, E0, E1, M0, M1);
        body(std::integral_constant<int, 1>{}, it + 1, D0, E1, E0, M1, M0);
    }
    f32x2 et_[PX];
    epi_read(et_);
    epi_finish(et_, E1, M1, epix_prev);
}
