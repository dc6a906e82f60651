// Winograd F(2x2, 3x3) / F(2x4, 3x3) with the products on the BF16 matrix pipe at fp32 accuracy (round 5).  Included by
// conv_winograd.hip (inside its anonymous namespace, after wino_conv_kernel, whose geometry, transforms and epilogue this
// kernel shares).
//
// Why.  gfx950 has no xf32 / TF32, and its fp32 MFMA runs at the vector FMA rate (157 TFLOP/s): every fp32 kernel of this library
// sits at 0.45 - 0.55 of that, the rest being the vector-ALU work of the transforms, which the same lanes execute
// (profiles/r4_pipe_overlap_probe.txt).  The bf16 MFMA runs at SIXTEEN times the fp32 rate.  A float is EXACTLY the sum of three
// bf16 numbers,
//     x = hi + mid + lo,   hi = the top 8 significant bits of x, mid = the top 8 bits of x - hi, lo = x - hi - mid  (<= 8 bits),
// taken by bit masks and exact subtractions (no rounding anywhere), and a product is the sum of the nine piece products, each
// EXACT in fp32 (8 x 8 bits).  The six of them down to 2^-16 of the leading one,
//     a b ~ a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1        (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a b|),
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, reproduce the fp32 result to fp32 rounding: measured against float64 on
// K = 4608 (tools/bf16x3_probe.hip, profiles/r5_bf16x3_probe.txt) max error 2.09e-6 of max|ref| (rms 3.3e-7) against 1.85e-6
// (3.5e-7) for the fp32 MFMA chain -- the same; a two-way split with three products would be 2.6e-5 and is not used.
// Six bf16 MFMAs of K = 16 replace eight fp32 MFMAs of K = 2: 192 instead of 512 matrix-pipe clocks per 16 channels.
//
// What changes against wino_conv_kernel<2, false, PX>:
//   * weights: transformed AND split once per optimizer step (wino_weights_kernel, split form) into
//     Uw3[xi][C/16][plane 0..2][N][16 c] bf16: a wave's B fragment of one (xi, n block, plane) is one coalesced 1 KB load, lane
//     (n, kh) takes the 16 bytes of channels 8 kh .. 8 kh + 7; three units (position, n block) ahead of their use in a ring
//     of three register sets;
//   * input: B^T d B in fp32 exactly as before (patch row per lane of a quad, DPP across the quad), then each value is split and the
//     three planes go to LDS as As[plane][xi][tile][16 c] bf16 (8-byte stores); the A fragment of a (position, plane) is one
//     16-byte LDS load per lane;
//   * K loop: per step of 16 channels and wave, PW positions x 2 n blocks x 6 MFMAs (PX = 6: 36 MFMAs = 1152 matrix-pipe clocks
//     where the fp32 kernel has 48 = 3072); the split costs ~5.5 vector instructions per transformed value (132 per thread and
//     step beside the transform's 116); with the matrix segment that short, every request is further ahead of its use than in
//     the fp32 kernel: the patch rows two steps (two register sets);
//   * accumulators, exchange image and epilogue are the fp32 kernel's (the bf16 MFMA has the same 32 x 32 result layout).
// Domain: everything wino_conv_kernel takes with N a multiple of 64 (after mtd_set_option("wino_split", 1) the library's plan
// sends those layers here; their weights are built in the split form: mtd_wino_weight_desc.px = 16 + PX, and a conv launch says
// so through a->w_st).
// Measured (profiles/r5_wino3_parts.txt): 62 against 76 us per launch on hot inputs (M 16384, N = C = 256), 49.5 against 53.1 us
// on the average launch of the training step, -0.27 ms per step: with two thirds of the matrix-pipe time gone a launch is bound
// by what a CU can take in per K step -- 147 KB of weight fragments and 98 KB of patch-row lines against 64 B / clock -- and that
// bound lies just under the fp32 MFMA one.  OFF by default.
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned w3_u32x2 __attribute__((ext_vector_type(2)));

#ifndef W3_SKIP
#define W3_SKIP 0        // lab (tools/wino3_variants.sh): 1 no MFMAs, 2 no transform / split arithmetic, 4 no weight loads, 8 no patch loads, 16 no LDS stores
#endif
constexpr int W3_XB = 1024 + 32;      // bytes between positions of one plane of As: 32 tiles x 16 bf16, + 32 (the four patch rows of a quad on different banks)

// upper halves of two words -> one word (first value in the low half): v_perm_b32
__device__ __forceinline__ unsigned w3_pack(unsigned lo_word, unsigned hi_word) { return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u); }

__device__ __forceinline__ f32x16 w3_mfma(w3_u32x4 a, w3_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w3_bf16x8, a), __builtin_bit_cast(w3_bf16x8, b), c, 0, 0, 0);
}

template <int PX>
__global__ __launch_bounds__(512, 1) void wino_conv3_kernel(const WinoParams wp) {
    constexpr int NB = 2;
    constexpr int NP = 4 * PX;            // transform positions xi = PX * (patch row) + (patch column)
    constexpr int PW = NP / 8;            // positions per wave
    constexpr int TWX = PX - 2;           // output pixels per tile row
    constexpr int NU = PW * NB;           // units (position, n block) per wave and K step
    constexpr int AS_BUF = 3 * NP * W3_XB;              // bytes of one input image: [plane][xi][tile][16 c] bf16
    constexpr int X_SIZE = NP * WT * WXLD * 4;          // bytes of the exchange image of the epilogue
    __shared__ __attribute__((aligned(16))) unsigned char Lb[(2 * AS_BUF > X_SIZE) ? 2 * AS_BUF : X_SIZE];
    float* const Ls = reinterpret_cast<float*>(Lb);
    const IgemmParams& p = wp.p;
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (wp.xcd_order) {                   // (XCD-contiguous workgroup orders: wino_conv_kernel)
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
    const int zk = bz;
    const int st_beg = zk * (p.c_per_split >> 4);
    const int st_end = min(wp.nchunk >> 1, st_beg + (p.c_per_split >> 4));
    const int nst = st_end - st_beg;
    const int st_last = st_end - 1;

    // ---- transform role: thread (tile tt, channel quad tq, patch row ti), as in wino_conv_kernel
    const int ti = tid & 3, tq = (tid >> 2) & 3, tt = tid >> 4;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    unsigned pbase, pvalid = 0;
    {
        const int tg = tile0 + tt;
        const bool tv = tg < wp.ntiles;
        int b, ty, tx;
        pix_decompose(tg, wp.tiles_x, g.OH >> 1, b, ty, tx);
        const int iy = 2 * ty - 1 + ti;
        pbase = (unsigned)(((((long long)b * g.IH + iy) * g.IW + (TWX * tx - 1)) * a.in_ld + 4 * tq) * 4);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int ix = TWX * tx - 1 + j;
            if (tv & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) pvalid |= 1u << j;
        }
    }
    const int px_b = a.in_ld * 4;
    f32x4 d0[PX], d1[PX];                  // two patch register sets: a step's patch is requested TWO steps ahead
    auto load_patch = [&](f32x4 (&d)[PX], int st) {
        if (W3_SKIP & 8) {
#pragma unroll
            for (int j = 0; j < PX; ++j) d[j] = f32x4{(float)st, 1.f, 2.f, (float)j};
            return;
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const unsigned vo = ((pvalid >> j) & 1u) ? pbase + (unsigned)(j * px_b) : 0x80000000u;
            d[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, st * 64, 0));
        }
    };
    const float qsign = ti == 1 ? 1.f : -1.f;
    auto row_value = [&](const f32x4 (&d)[PX], int j) -> f32x4 {
        if (W3_SKIP & 2) return d[j];
        if constexpr (PX == 4) {
            return j == 0 ? d[0] - d[2] : (j == 1 ? d[1] + d[2] : (j == 2 ? d[2] - d[1] : d[1] - d[3]));
        } else {
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
    // B^T d B, split, -> As[plane][xi = PX ti + j][tt][channels 4 tq .. 4 tq + 3] (8 bytes per plane)
    auto transform_cols = [&](const f32x4 (&d)[PX], unsigned char* As, int j0) {
        unsigned char* o = As + tt * 32 + tq * 8;
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j) {
            const f32x4 u = (W3_SKIP & 2) ? row_value(d, j) : wino_quad_rows(row_value(d, j), qsign);
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (W3_SKIP & 2) h[c] = m[c] = l[c] = __builtin_bit_cast(unsigned, u[c]);
                else w3_split(u[c], h[c], m[c], l[c]);
            }
            const int xi = PX * ti + j;
            if (W3_SKIP & 16) { asm volatile("" :: "v"(h[0]), "v"(m[1]), "v"(l[2]), "v"(h[3])); continue; }
            *reinterpret_cast<w3_u32x2*>(o + (0 * NP + xi) * W3_XB) = w3_u32x2{w3_pack(h[0], h[1]), w3_pack(h[2], h[3])};
            *reinterpret_cast<w3_u32x2*>(o + (1 * NP + xi) * W3_XB) = w3_u32x2{w3_pack(m[0], m[1]), w3_pack(m[2], m[3])};
            *reinterpret_cast<w3_u32x2*>(o + (2 * NP + xi) * W3_XB) = w3_u32x2{w3_pack(l[0], l[1]), w3_pack(l[2], l[3])};
        }
    };
    auto transform_store = [&](const f32x4 (&d)[PX], unsigned char* As) {
#pragma unroll
        for (int j0 = 0; j0 < PX; j0 += 2) transform_cols(d, As, j0);
    };

    // ---- MFMA role: positions PW wave .. PW wave + PW - 1.  B fragments of unit u = (x = u / NB, nb = u % NB): three planes
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), (short)0, (int)wp.w_bytes, 0x00020000);
    const unsigned w_lane = (unsigned)((n0 + l31) * 32 + kh * 16);
    const int plane_b = a.N * 32;                              // bytes of one plane of one (position, 16-channel chunk)
    const int chunk_b = 3 * plane_b;
    const int xi_stride_b = (wp.nchunk >> 1) * chunk_b;        // bytes between positions
    const int w_pos0 = PW * wave * xi_stride_b;
    auto load_b = [&](int st, int u, w3_u32x4 (&bf)[3]) {      // step st (absolute), unit u
        const int x = u / NB, nb = u % NB;
        if (W3_SKIP & 4) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[pl] = w3_u32x4{(unsigned)st, (unsigned)u, 0x3f803f80u, (unsigned)pl};
            return;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            bf[pl] = __builtin_bit_cast(w3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, w_lane + (unsigned)(nb * 1024), w_pos0 + x * xi_stride_b + st * chunk_b + pl * plane_b, W_AUX));
    };
    f32x16 acc[PW][NB];
#pragma unroll
    for (int x = 0; x < PW; ++x)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[x][nb][e] = 0.f;
    w3_u32x4 af[3];                                           // A fragments of the current position: three planes
    auto load_af = [&](const unsigned char* Ac, int x) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[pl] = *reinterpret_cast<const w3_u32x4*>(Ac + (pl * NP + PW * wave + x) * W3_XB + l31 * 32 + kh * 16);
    };
    auto mfma_unit = [&](int u, const w3_u32x4 (&bf)[3]) {     // the six products of one (position, n block), smallest first
        f32x16 c = acc[u / NB][u % NB];
        if (W3_SKIP & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] += __builtin_bit_cast(float, af[0][e] ^ bf[0][e] ^ af[1][e] ^ bf[1][e] ^ af[2][e] ^ bf[2][e]);
            acc[u / NB][u % NB] = c;
            return;
        }
        c = w3_mfma(af[0], bf[2], c);
        c = w3_mfma(af[2], bf[0], c);
        c = w3_mfma(af[1], bf[1], c);
        c = w3_mfma(af[0], bf[1], c);
        c = w3_mfma(af[1], bf[0], c);
        c = w3_mfma(af[0], bf[0], c);
        acc[u / NB][u % NB] = c;
    };

    // ---- K loop: one stream per wave -- the MFMAs of step j, then the transform + split of step j + 1 into the other image, one
    // barrier per step.  With the matrix segment down to 1 152 clocks nothing of a step covers a memory round trip any more
    // (profiles/r5_wino3_parts.txt: the parts of a step ADD UP), so every request is far ahead of its use: the patch rows TWO steps
    // (two register sets, the loop runs two steps per trip), the weight fragments three units in a ring of three register sets
    // (PX = 4: four sets, a whole step).  Every load is unconditional (clamped indices): ONE path through the loop, so the
    // compiler's vmcnt waits count exactly the younger requests (wino_conv_kernel).
    constexpr int RING = NU == 6 ? 3 : NU;
    static_assert(NU == 6 || NU == 4, "units per step");
    w3_u32x4 br[RING][3];
    if (nst > 0) {
        load_patch(d0, st_beg);
#pragma unroll
        for (int u = 0; u < RING; ++u) load_b(st_beg, u, br[u]);
        load_patch(d1, min(st_beg + 1, st_last));
        transform_store(d0, Lb);
        load_patch(d0, min(st_beg + 2, st_last));
    }
    __syncthreads();
    auto step = [&](int j, f32x4 (&d)[PX]) {          // d: the patch of step j + 1 (requested two steps ago); refilled for step j + 3
        const int st = st_beg + j;
        const int stn = min(st + 1, st_last);
        const unsigned char* Ac = Lb + (j & 1) * AS_BUF;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u % NB == 0) load_af(Ac, u / NB);
            // the unit's six MFMAs with the three weight requests of the unit RING units ahead BETWEEN them, two MFMAs per request (the
            // requests refill the ring slot this unit reads: each plane's request follows the last MFMA that reads that plane) --
            // not as a burst behind the unit (profiles/r5_load_spreading.txt)
            f32x16 c = acc[u / NB][u % NB];
            const w3_u32x4 (&bf)[3] = br[u % RING];
            const int lst = (u + RING < NU) ? st : stn, lu = (u + RING < NU) ? u + RING : u + RING - NU;
            const int x = lu / NB, nb = lu % NB;
            auto ld = [&](int pl) {
                return __builtin_bit_cast(w3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, w_lane + (unsigned)(nb * 1024), w_pos0 + x * xi_stride_b + lst * chunk_b + pl * plane_b, W_AUX));
            };
            if (W3_SKIP & 1) {
                mfma_unit(u, br[u % RING]);
                load_b(lst, lu, br[u % RING]);
            } else {
                c = w3_mfma(af[0], bf[2], c);
                c = w3_mfma(af[2], bf[0], c);
                const w3_u32x4 n2 = (W3_SKIP & 4) ? bf[2] : ld(2);          // plane 2 was read by the first MFMA only
                c = w3_mfma(af[1], bf[1], c);
                c = w3_mfma(af[0], bf[1], c);
                const w3_u32x4 n1 = (W3_SKIP & 4) ? bf[1] : ld(1);
                c = w3_mfma(af[1], bf[0], c);
                c = w3_mfma(af[0], bf[0], c);
                const w3_u32x4 n0 = (W3_SKIP & 4) ? bf[0] : ld(0);
                acc[u / NB][u % NB] = c;
                br[u % RING][2] = n2;
                br[u % RING][1] = n1;
                br[u % RING][0] = n0;
                if (u % NB == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        transform_store(d, Lb + ((j + 1) & 1) * AS_BUF);
        load_patch(d, min(st + 3, st_last));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
#pragma unroll 1
    for (int j = 0; j < nst; j += 2) {
        step(j, d1);
        if (j + 1 < nst) step(j + 1, d0);
    }

    // ---- epilogue: wino_conv_kernel's, on the same accumulator layout
    const ScalePair sp = load_scale(a);
    const int ei = tid & 1, enq = (tid >> 1) & 7, etl = tid >> 4;
    int epix;
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
    const bool vec = (p.wide & 1) != 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
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
            float* X = Ls + (PW * wave + x) * (WT * WXLD) + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) X[mfma32_row(e, lane) * WXLD] = acc[x][nb][e];
        }
        __syncthreads();
        f32x4 y[TWX];
        {
            f32x4 t[PX];
#pragma unroll
            for (int b = 0; b < PX; ++b) {
                const float* col = Ls + (b * WT + etl) * WXLD + 4 * enq;
                const f32x4 m1 = *reinterpret_cast<const f32x4*>(col + 1 * PX * WT * WXLD);
                const f32x4 m2 = *reinterpret_cast<const f32x4*>(col + 2 * PX * WT * WXLD);
                const f32x4 m03 = *reinterpret_cast<const f32x4*>(col + (ei ? 3 : 0) * PX * WT * WXLD);
                t[b] = ei ? m1 - m2 - m03 : m03 + m1 + m2;
            }
            if constexpr (PX == 4) {
                y[0] = t[0] + t[1] + t[2];
                y[1] = t[1] - t[2] - t[3];
            } else {
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
