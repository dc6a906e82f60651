// Implicit-GEMM convolution / data-gradient on fp32 MFMA for gfx950 (MI355X).
//
//   out[pix, n] = epilogue( sum_{tap} sum_{c} in[gather(pix, tap), c] * W(n, c, tap) )
//
// GEMM view: M = B*OH*OW pixels, N output channels, K = taps * C, walked as (tap, 32-channel chunk).
// v_mfma_f32_32x32x2_f32 takes, per lane, A[row = lane&31][k = lane>>5] and B[k = lane>>5][col = lane&31].
// The K order inside a chunk is free, so k-step kk (0..15) is defined to use channels (kk, 16+kk): lane
// (m, kh) then needs channels kh*16 .. kh*16+15 of ITS OWN pixel -- 64 contiguous bytes of the NHWC
// tensor.  The A operand is therefore loaded straight from global memory into registers (4 x 16-byte
// loads per 32-pixel tile and chunk, zero-filled outside the image), with the next chunk's loads in
// flight under the current chunk's 64-cycle MFMAs: no LDS round trip, no barrier for A.  Only the weight
// tile (shared by all waves) goes through LDS: [BN][32 (+4 pad)] floats, double-buffered, one barrier
// per chunk, read back as conflict-free ds_read_b128.  Weights are read in place from PyTorch's OIHW /
// IOHW storage through element strides.  Taps that fall outside the image for every pixel of the tile
// (2x2 / 1x1 feature maps) are skipped for the whole workgroup; small-M layers use split-K over channel
// ranges (slabs in the caller's workspace + a fused reduce/epilogue kernel, fixed order => deterministic).
//
// Roofline: fp32 MFMA, 64 FLOP/clk/SIMD (157.3 TFLOP/s chip).  Algorithmic flops = 2*M*N*K.
#include "common.h"
#ifndef S2DG_PLAN
#define S2DG_PLAN 1      // round-5 tile rules for the four-class stride-2 data gradients (make_plan)
#endif
#ifdef MTD_LAB
#define MTD_IGEMM_FIN 1
#else
#define MTD_IGEMM_FIN 0      // the in-kernel split-K finish exists in lab builds only (see "split-K finish inside the kernel")
#endif
#include "fft64.h"

// In-kernel phase stamps for the diagnostic build only (tools/igemm_stamp.hip defines MTD_STAMPS and includes this
// file); in the library build MTD_STAMP expands to nothing.
#ifdef MTD_STAMPS
__device__ unsigned long long* mtd_stamp_buf;
#define MTD_STAMP(i)                                                                                        \
    do {                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0 && blockIdx.z == 0) {                    \
            __builtin_amdgcn_sched_barrier(0);                                                              \
            unsigned long long t__;                                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                      \
            mtd_stamp_buf[blockIdx.x * 64 + (i)] = t__;                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                              \
        }                                                                                                   \
    } while (0)
#else
#define MTD_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int KC = 32;      // channels per K chunk
constexpr int BLD = 36;     // LDS row stride of the weight tile (floats): 16-B aligned, conflict-free b128

struct IgemmParams {
    mtd_conv_args a;
    int M;
    int splitk;
    int c_per_split;
    unsigned in_bytes; // extent of the input view in bytes (buffer-load range check)
    unsigned w_bytes;  // extent of the weight view in bytes
    int tap_dy[16], tap_dx[16];     // input displacement of tap t (validity test)
    int tap_delta[16];              // byte displacement of tap t in the NHWC input
    int tap_kidx[16];               // index of tap t in the kh*kw plane of the weights
    int out_identity;  // output pixel index == launch-grid pixel index
    int out_linear;    // 32 consecutive launch-grid pixels (from a multiple of 32) map to output pixels pix0 + r * out_sx
    int xcd_map;       // tiles in XCD-contiguous, n-fastest order (tile_of below)
    int nt_store;      // lab switch MTD_IGEMM_NT=1: non-temporal output stores in the block epilogue
    int fin;           // split-K: the last workgroup to arrive at a tile sums the slabs and runs the epilogue (a.tile_ctr; lab builds)
    int wide;          // igemm_body: bit 0 = every epilogue operand row is 16-byte aligned (EpiWide), bit 1 = the slabs are
};

// (m tile, n tile) of this workgroup.  Dispatch order is blockIdx.x fastest and consecutive workgroups land on different
// XCDs, so with the plain mapping the N/BN workgroups that read the same activation tile are spread over all eight L2s
// and over time.  With xcd_map every XCD walks its own contiguous run of tiles, n fastest: an activation tile is fetched
// into one L2 once and reused by all its n tiles while the (small) weight block stays resident.
__device__ __forceinline__ void tile_of(const IgemmParams& p, int& tm, int& tn) {
    if (!p.xcd_map) { tm = blockIdx.x; tn = blockIdx.y; return; }
    const int MT = gridDim.x, NT = gridDim.y;
    const int t = xcd_contiguous_block(blockIdx.y * MT + blockIdx.x, MT * NT);
    tm = t / NT;
    tn = t - tm * NT;
}

__device__ __forceinline__ long long out_pixel(const mtd_geom& g, int m, int identity) {
    if (identity) return m;
    int b, oy, ox;
    pix_decompose(m, g.OW, g.OH, b, oy, ox);
    return ((long long)b * g.OHF + (oy * g.out_sy + g.out_oy)) * g.OWF + (ox * g.out_sx + g.out_ox);
}

__device__ __forceinline__ float epilogue_value(const mtd_conv_args& a, float acc, float sc, float bias_n,
                                                long long pix, int n) {
    float v = acc * sc + bias_n;
    if (a.act == MTD_ACT_RELU_ADD) {        // the residual operands AFTER the activation (whole-slice Res-FFT block)
        v = v > 0.f ? v : 0.f;
        if (a.add1) v += a.add1[pix * a.add1_ld + n];
        if (a.add2) v += a.add2[pix * a.add2_ld + n];
        return v;
    }
    if (a.add1) v += a.add1[pix * a.add1_ld + n];
    if (a.add2) v += a.add2[pix * a.add2_ld + n];
    v = apply_act(v, a.act);
    if (a.mask) v *= (a.mask[pix * a.mask_ld + n] > 0.f) ? 1.f : a.mask_slope;
    return v;
}

// ---- epilogue of one 32 x 32 accumulator block ---------------------------------------------------------------------
// A lane holds 16 values: rows mfma32_row(e, lane) of the block, column n.  Everything that is uniform per launch (the
// output mapping, which operands exist, the activation) is decided ONCE around groups of NE values, and the operand loads
// of a group are issued back to back; the per-element form (branches and an address division chain inside the loop,
// each load followed by its use) cost ~100 instructions and a memory round trip per value.  With the identity output
// mapping a value's address is (wave-uniform row base) + (one 32-bit lane offset), so a group needs no address registers.
// The arithmetic per value and its order are those of epilogue_value().  Absent operands are the neutral elements -0.0f
// (x + -0.0f == x for every x, signed zeros included) and a mask of 1.  Groups of 8 keep the generic kernels at their
// main-loop register count; the halo-tile kernel loads all 16 before its MFMA loop.
template <int NE> struct EpiOps { float e1[NE], e2[NE], em[NE]; };

// Addresses of values E0 .. E0+NE-1 of a block whose 32 rows map to output pixels pix0 + r * step ("linear": the identity
// mapping, or a strided one on rows of >= 32 pixels -- p.out_linear).
template <bool FULL, int E0, int NE>
struct EpiAddr {
    int mu, lane, lane_row, n, step;
    long long pix0;
    __device__ __forceinline__ void init(const IgemmParams& p, int mrow0, int lane_, int n_) {
        mu = __builtin_amdgcn_readfirstlane(mrow0);
        lane = lane_;
        n = n_;
        step = p.out_identity ? 1 : p.a.g.out_sx;
        lane_row = 4 * (lane_ >> 5) * step;
        pix0 = p.out_identity ? (long long)mu : out_pixel(p.a.g, mu, 0);
    }
    __device__ __forceinline__ bool ok(const IgemmParams& p, int i) const { return FULL || mu + mfma32_row(E0 + i, lane) < p.M; }
    template <typename Tp>
    __device__ __forceinline__ Tp* at(Tp* base, int ld, int i) const {
        return base + (pix0 + (((E0 + i) & 3) + 8 * ((E0 + i) >> 2)) * step) * ld + (lane_row * ld + n);
    }
};

template <bool FULL, int E0, int NE>
__device__ __forceinline__ void epi_load(const IgemmParams& p, const EpiAddr<FULL, E0, NE>& ad, EpiOps<NE>& o) {
    const mtd_conv_args& a = p.a;
#pragma unroll
    for (int i = 0; i < NE; ++i) { o.e1[i] = -0.0f; o.e2[i] = -0.0f; o.em[i] = 1.f; }
    if (a.add1) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (ad.ok(p, i)) o.e1[i] = *ad.at(a.add1, a.add1_ld, i);
    }
    if (a.add2) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (ad.ok(p, i)) o.e2[i] = *ad.at(a.add2, a.add2_ld, i);
    }
    if (a.mask) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (ad.ok(p, i)) o.em[i] = *ad.at(a.mask, a.mask_ld, i);
    }
}

template <bool FULL, int E0, int NE>
__device__ __forceinline__ void epi_store(const IgemmParams& p, const f32x16& acc, const EpiAddr<FULL, E0, NE>& ad, const ScalePair& sp,
                                          float bias_n, const EpiOps<NE>& o) {
    const mtd_conv_args& a = p.a;
    float v[NE];
    const bool post = a.act == MTD_ACT_RELU_ADD;       // the residual operands AFTER the activation (whole-slice Res-FFT block)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        v[i] = acc[E0 + i] * pick_scale(sp, ad.mu + mfma32_row(E0 + i, ad.lane)) + bias_n;
        if (post) v[i] = v[i] > 0.f ? v[i] : 0.f;
        v[i] += o.e1[i];
        v[i] += o.e2[i];
    }
    if (a.act == MTD_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    } else if (a.act == MTD_ACT_LRELU) {
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] = v[i] > 0.f ? v[i] : 0.2f * v[i];
    }
    if (a.out2) {       // the unmasked value as well (halo-tile kernel; rejected by the dispatch elsewhere)
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (ad.ok(p, i)) *ad.at(a.out2, a.out2_ld, i) = v[i];
    }
    if (a.mask) {
        const float slope = a.mask_slope;
#pragma unroll
        for (int i = 0; i < NE; ++i) v[i] *= (o.em[i] > 0.f) ? 1.f : slope;
    }
    if (p.nt_store) {
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (ad.ok(p, i)) __builtin_nontemporal_store(v[i], ad.at(a.out, a.out_ld, i));
        return;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i)
        if (ad.ok(p, i)) *ad.at(a.out, a.out_ld, i) = v[i];
}

template <bool FULL, int E0, int NE>
__device__ __forceinline__ void epi_group(const IgemmParams& p, const f32x16& acc, int mrow0, int lane, int n, const ScalePair& sp,
                                          float bias_n) {
    EpiAddr<FULL, E0, NE> ad;
    EpiOps<NE> o;
    ad.init(p, mrow0, lane, n);
    epi_load(p, ad, o);
    epi_store(p, acc, ad, sp, bias_n, o);
    __builtin_amdgcn_sched_barrier(0);
}

// rows mrow0 .. mrow0 + 31 of the launch (mrow0 is wave-uniform and a multiple of 32), column n of this lane.  NE = values
// per group: 8 for one accumulator per wave, 4 for register-blocked tiles (keeps their register count -- and with it the
// resident waves per SIMD -- at the main loop's).
template <int NE>
__device__ __forceinline__ void epilogue16(const IgemmParams& p, const f32x16& acc, int mrow0, int lane, int n, const ScalePair& sp) {
    if (mrow0 >= p.M) return;
    const mtd_conv_args& a = p.a;
    const float bias_n = a.bias ? a.bias[n] : 0.f;
    if (p.out_linear) {
        if (mrow0 + 32 <= p.M) {
            if (NE == 8) {
                epi_group<true, 0, 8>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<true, 8, 8>(p, acc, mrow0, lane, n, sp, bias_n);
            } else {
                epi_group<true, 0, 4>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<true, 4, 4>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<true, 8, 4>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<true, 12, 4>(p, acc, mrow0, lane, n, sp, bias_n);
            }
        } else {
            if (NE == 8) {
                epi_group<false, 0, 8>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<false, 8, 8>(p, acc, mrow0, lane, n, sp, bias_n);
            } else {
                epi_group<false, 0, 4>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<false, 4, 4>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<false, 8, 4>(p, acc, mrow0, lane, n, sp, bias_n);
                epi_group<false, 12, 4>(p, acc, mrow0, lane, n, sp, bias_n);
            }
        }
        return;
    }
    // small maps with a strided output (rows shorter than a block): value by value
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mrow0 + mfma32_row(e, lane);
        if (m < p.M) {
            const long long pix = out_pixel(a.g, m, 0);
            a.out[pix * a.out_ld + n] = epilogue_value(a, acc[e], pick_scale(sp, m), bias_n, pix, n);
        }
    }
}

// ---- wide epilogue of one 32 x 32 accumulator block computed TRANSPOSED -------------------------------------------
// mfma32(b, a, acc) instead of mfma32(a, b, acc) -- the two operand fragments have the same per-lane format, so swapping
// them costs nothing -- gives the transposed block: lane (l31, kh) then holds, for ITS OWN pixel mrow0 + l31, the 16
// channels n0 + 8 g + 4 kh + j (g, j = 0..3; register 4 g + j): four groups of four CONSECUTIVE channels.  Every epilogue
// operand and the result move as 16-byte vectors, 4 instructions per tensor and block instead of 16.  That is the point:
// a vector-memory instruction costs the CU ~60-75 cycles whatever its width (in-kernel stamps and the per-launch
// instruction counts of the generator's 32-channel layers), and the dword epilogue of a block with an add, a
// mask and a second output is 64 of them for 144 MFMAs.  Same dot products in the same k order: bit-identical values.
// Needs 16-byte aligned rows: bases aligned, every ld a multiple of 4 floats (wide_epilogue_ok).
struct EpiWide {
    long long pix;      // the lane's output pixel
    int ch;             // first channel of group 0: n0 + 4 kh
    float sc;
    __device__ __forceinline__ void init(const IgemmParams& p, int mrow0, int lane, int n0, const ScalePair& sp) {
        const int m = mrow0 + (lane & 31);
        pix = p.out_identity ? (long long)m : out_pixel(p.a.g, m, 0);
        ch = n0 + 4 * (lane >> 5);
        sc = pick_scale(sp, m);
    }
};
struct EpiWideOps { f32x4 e1[4], e2[4], em[4]; };

__device__ __forceinline__ void epiw_load(const IgemmParams& p, const EpiWide& ad, EpiWideOps& o) {
    const mtd_conv_args& a = p.a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        o.e1[g] = f32x4{-0.0f, -0.0f, -0.0f, -0.0f};
        o.e2[g] = f32x4{-0.0f, -0.0f, -0.0f, -0.0f};
        o.em[g] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    if (a.add1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) o.e1[g] = *reinterpret_cast<const f32x4*>(a.add1 + ad.pix * a.add1_ld + ad.ch + 8 * g);
    }
    if (a.add2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) o.e2[g] = *reinterpret_cast<const f32x4*>(a.add2 + ad.pix * a.add2_ld + ad.ch + 8 * g);
    }
    if (a.mask) {
#pragma unroll
        for (int g = 0; g < 4; ++g) o.em[g] = *reinterpret_cast<const f32x4*>(a.mask + ad.pix * a.mask_ld + ad.ch + 8 * g);
    }
}

// bias4[g]: the bias of the lane's channel group g (zeros without a bias)
__device__ __forceinline__ void epiw_store(const IgemmParams& p, const f32x16& acc, const EpiWide& ad, const f32x4 (&bias4)[4],
                                           const EpiWideOps& o) {
    const mtd_conv_args& a = p.a;
    f32x4 v[4];
    if (a.act == MTD_ACT_RELU_ADD) {        // the residual operands AFTER the activation (no mask, no second output)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = acc[4 * g + j] * ad.sc + bias4[g][j];
                x = x > 0.f ? x : 0.f;
                x += o.e1[g][j];
                x += o.e2[g][j];
                v[g][j] = x;
            }
            *reinterpret_cast<f32x4*>(a.out + ad.pix * a.out_ld + ad.ch + 8 * g) = v[g];
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = acc[4 * g + j] * ad.sc + bias4[g][j];      // the order of epilogue_value()
            x += o.e1[g][j];
            x += o.e2[g][j];
            v[g][j] = x;
        }
    if (a.act == MTD_ACT_RELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] = v[g][j] > 0.f ? v[g][j] : 0.f;
    } else if (a.act == MTD_ACT_LRELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] = v[g][j] > 0.f ? v[g][j] : 0.2f * v[g][j];
    }
    if (a.out2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a.out2 + ad.pix * a.out2_ld + ad.ch + 8 * g) = v[g];
    }
    if (a.mask) {
        const float slope = a.mask_slope;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] *= (o.em[g][j] > 0.f) ? 1.f : slope;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a.out + ad.pix * a.out_ld + ad.ch + 8 * g) = v[g];
}

// Res-FFT block tail (igemm_c32t_kernel SPEC): out2 = act(acc * sc + bias), out = out2 + xs, xs = the lane's pieces of
// x + irfft_rows (halo tile).  No add / mask operands.
__device__ __forceinline__ void epiw_store_spec(const IgemmParams& p, const f32x16& acc, const EpiWide& ad, const f32x4 (&bias4)[4],
                                                const f32x4 (&xs)[4]) {
    const mtd_conv_args& a = p.a;
    f32x4 v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[g][j] = acc[4 * g + j] * ad.sc + bias4[g][j];
    if (a.act == MTD_ACT_RELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] = v[g][j] > 0.f ? v[g][j] : 0.f;
    } else if (a.act == MTD_ACT_LRELU) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[g][j] = v[g][j] > 0.f ? v[g][j] : 0.2f * v[g][j];
    }
    if (a.out2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a.out2 + ad.pix * a.out2_ld + ad.ch + 8 * g) = v[g];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(a.out + ad.pix * a.out_ld + ad.ch + 8 * g) = v[g] + xs[g];
}

__device__ __forceinline__ void epiw_bias(const mtd_conv_args& a, int ch, f32x4 (&bias4)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + ch + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- split-K finish inside the kernel (LAB BUILDS ONLY since round 5: the variant lost twice -- neutral in round 2, +0.2 ms in
// round 4 -- and the Winograd kernel's form of it lost in round 5, profiles/r5_winograd_in_launch_splitk_finish.txt; in the shipped
// library MTD_IGEMM_FIN is 0, p.fin is never set and the branches below are compiled out) ------------------------------------------
// Every slice stores its partial tile to its slab as before, then arrives at the tile's counter (a.tile_ctr, zero on
// entry).  The workgroup that completes the count re-reads ALL slabs of the tile in slice order 0 .. splitk-1 -- the
// sum is the one splitk_epilogue_kernel forms, bit for bit, whichever slice arrives last -- and runs the ordinary
// epilogue on it; it also puts the counter back to zero for the next launch.
// The slabs cross XCDs (one L2 each).  A device-scope release / acquire fence pair does that with a write-back and an
// invalidate of the WHOLE L2 per workgroup: measured +40 us per launch (full step 42.3 -> 48.2 ms).  Instead the slab
// stores and the re-reads of a finishing launch are device-scope accesses themselves (relaxed atomics: written through
// to / read from the memory side, no line left dirty or stale in an L2), and the arrival waits for the stores'
// acknowledgements (vmcnt) before the counter is bumped.
__device__ __forceinline__ void slab_store(float* p, float v, bool through) {
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

__device__ __forceinline__ bool splitk_last_arrival(const IgemmParams& p, int tile) {
    __shared__ unsigned last_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.a.tile_ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)p.splitk - 1u;
        if (last) __hip_atomic_store(p.a.tile_ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = last ? 1u : 0u;
    }
    __syncthreads();
    return last_s != 0u;
}

// 16 bytes of a slab, device scope (the finishing workgroup: see above)
__device__ __forceinline__ f32x4 slab_load4(const float* p) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __hip_atomic_load(p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}

// One output pixel x four consecutive channels: the slabs summed in slice order (from +0.0f), then the epilogue in the
// order of epilogue_value().  16-byte slab reads, up to eight slices in flight; needs splitk_vec_ok().
template <bool DEV>
__device__ __forceinline__ void splitk_finish_vec4(const IgemmParams& p, const ScalePair& sp, unsigned m, unsigned n, long long total) {
    const mtd_conv_args& a = p.a;
    const float* base = a.ws + ((long long)m * a.N + n);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= p.splitk; z += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = DEV ? slab_load4(base + (long long)(z + j) * total) : *reinterpret_cast<const f32x4*>(base + (long long)(z + j) * total);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    if (z + 4 <= p.splitk) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = DEV ? slab_load4(base + (long long)(z + j) * total) : *reinterpret_cast<const f32x4*>(base + (long long)(z + j) * total);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += v[j];
        z += 4;
    }
    for (; z < p.splitk; ++z) s += DEV ? slab_load4(base + (long long)z * total) : *reinterpret_cast<const f32x4*>(base + (long long)z * total);
    const long long pix = out_pixel(a.g, (int)m, p.out_identity);
    const float sc = pick_scale(sp, (int)m);
    f32x4 bias = {0.f, 0.f, 0.f, 0.f}, e1 = {-0.0f, -0.0f, -0.0f, -0.0f}, e2 = e1, em = {1.f, 1.f, 1.f, 1.f};
    if (a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.add1) e1 = *reinterpret_cast<const f32x4*>(a.add1 + pix * a.add1_ld + n);
    if (a.add2) e2 = *reinterpret_cast<const f32x4*>(a.add2 + pix * a.add2_ld + n);
    if (a.mask) em = *reinterpret_cast<const f32x4*>(a.mask + pix * a.mask_ld + n);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = s[j] * sc + bias[j];      // the order of epilogue_value(); absent operands are neutral elements
        v += e1[j];
        v += e2[j];
        v = apply_act(v, a.act);
        if (a.mask) v *= (em[j] > 0.f) ? 1.f : a.mask_slope;
        o[j] = v;
    }
    *reinterpret_cast<f32x4*>(a.out + pix * a.out_ld + n) = o;
}

template <bool DEV>
__device__ __forceinline__ void splitk_finish_scalar(const IgemmParams& p, const ScalePair& sp, int m, int n, long long total) {
    const mtd_conv_args& a = p.a;
    const long long idx = (long long)m * a.N + n;
    float s = 0.f;
    for (int z = 0; z < p.splitk; ++z)
        s += DEV ? __hip_atomic_load(a.ws + (long long)z * total + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.ws[(long long)z * total + idx];
    const long long pix = out_pixel(a.g, m, p.out_identity);
    const float bias_n = a.bias ? a.bias[n] : 0.f;
    a.out[pix * a.out_ld + n] = epilogue_value(a, s, pick_scale(sp, m), bias_n, pix, n);
}

// the BM x BN tile at (m0, n0), by the 256 threads of the workgroup that arrived last (p.fin: 1 = 16-byte form, 2 = scalar)
template <int BM, int BN>
__device__ __forceinline__ void splitk_finish_tile(const IgemmParams& p, int m0, int n0) {
    const long long total = (long long)p.M * p.a.N;
    const ScalePair sp = load_scale(p.a);
    if (p.fin == 1) {
        constexpr int Q = BN / 4;
        for (int i = threadIdx.x; i < BM * Q; i += 256) {
            const int m = m0 + i / Q;
            if (m < p.M) splitk_finish_vec4<true>(p, sp, (unsigned)m, (unsigned)(n0 + 4 * (i % Q)), total);
        }
    } else {
        for (int i = threadIdx.x; i < BM * BN; i += 256) {
            const int m = m0 + i / BN;
            if (m < p.M) splitk_finish_scalar<true>(p, sp, m, n0 + i % BN, total);
        }
    }
}

// zk: this workgroup's split-K slice (blockIdx.z of a single launch; blockIdx.z % splitk of a multi launch, below)
template <int WM, int WN, int WGM, int WGN, bool FIN = true>
__device__ __forceinline__ void igemm_body(const IgemmParams& p, const int zk) {
    constexpr int BM = 32 * WM * WGM, BN = 32 * WN * WGN;
    constexpr int PB = BN / 32;     // 16-byte weight vectors staged per thread and chunk (BN rows x 8 vectors)
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BLD];

    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    int tile_m, tile_n;
    tile_of(p, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cbeg = zk * p.c_per_split;
    const int cend = min(a.C, cbeg + p.c_per_split);
    const int T = g.TH * g.TW;

    MTD_STAMP(0);
    // ---- per-lane pixels (one per M tile): byte offset of tap (0,0) (mod 2^32; a valid tap always lands
    //      inside [0, in_bytes)), and the set of taps inside the image.  A is read with buffer loads:
    //      32-bit offsets, and an out-of-range offset returns 0, which gives the zero padding for free.
    unsigned boff[WM];
    unsigned okmask[WM];
    unsigned anymask = 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = m0 + (wm * WM + i) * 32 + l31;
        okmask[i] = 0;
        boff[i] = 0;
        if (m < p.M) {
            int b, oy, ox;
            pix_decompose(m, g.OW, g.OH, b, oy, ox);
            const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
            boff[i] = (unsigned)(((((long long)b * g.IH + py) * g.IW + px) * a.in_ld + kh * 16) * 4);
            for (int t = 0; t < T; ++t) {
                const int iy = py + p.tap_dy[t], ix = px + p.tap_dx[t];
                if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) okmask[i] |= 1u << t;
            }
        }
        anymask |= okmask[i];
    }
    // taps used by at least one pixel of the workgroup's tile (one LDS atomic per wave, one barrier)
    __shared__ unsigned vmask_s;
    if (tid == 0) vmask_s = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anymask |= __shfl_xor(anymask, off, 64);
    if (lane == 0) atomicOr(&vmask_s, anymask);
    __syncthreads();
    const unsigned vmask = vmask_s;
    MTD_STAMP(1);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- K iterator (workgroup-uniform)
    int tap = -1, c0 = cend, kidx = 0;      // c0 = cend: the first advance() opens the first valid tap
    // per-thread element offsets of the weight vectors it stages (chunk / tap terms are uniform adds);
    // the weight view is contiguous along c, so a row of the tile is 8 x 16 bytes
    int woff[PB], wlds[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int e = tid + i * 256;
        const int n = e >> 3, q = e & 7;
        woff[i] = (int)((long long)(n0 + n) * a.w_sn) + 4 * q;
        wlds[i] = n * BLD + 4 * q;
    }
    unsigned tapdelta = 0;
    // moves to the next valid (tap, chunk); on exhaustion returns false and leaves the state on the last
    // valid chunk so that the (masked) trailing prefetch still forms in-range weight addresses
    auto advance = [&]() -> bool {
        if (c0 + KC < cend) { c0 += KC; return true; }
        const unsigned done = (tap < 0) ? 0u : ((tap >= 31) ? 0xFFFFFFFFu : ((2u << tap) - 1u));
        const unsigned rem = vmask & ~done;                                        // taps after the current one
        if (rem == 0u || cbeg >= cend) return false;
        tap = __builtin_ctz(rem);
        c0 = cbeg;
        kidx = p.tap_kidx[tap];
        tapdelta = (unsigned)p.tap_delta[tap];
        return true;
    };
    f32x4 an[WM][4];     // next chunk's A fragments (global -> registers)
    f32x4 bn[PB];        // next chunk's weight vectors
    // The prefetch is issued in pieces BETWEEN groups of MFMAs (an in-order wave stalls at a VMEM instruction
    // while the texture-address unit is busy; with 2-4 MFMAs queued ahead of each load the matrix pipe keeps
    // running through those stalls).  Every piece is unconditional (masked out-of-range after the last chunk).
    unsigned lm = 0;
    auto load_a = [&](int i) {
        const unsigned voff = ((okmask[i] >> tap) & lm & 1u) ? (boff[i] + tapdelta + (unsigned)c0 * 4u) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            an[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 16 * j, 0));
    };
    auto load_b = [&]() {
        const int wchunk = c0 + (int)((long long)kidx * a.w_st);
#pragma unroll
        for (int i = 0; i < PB; ++i) bn[i] = *reinterpret_cast<const f32x4*>(a.w + (woff[i] + wchunk));
    };
    auto load = [&](bool live) {
        lm = 0u - (unsigned)live;
#pragma unroll
        for (int i = 0; i < WM; ++i) load_a(i);
        load_b();
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][wlds[i]]) = bn[i];
    };

    f32x4 ac[WM][4];
    f32x4 bc[WN][4];
    auto mfma_steps = [&](int k0, int k1) {
#pragma unroll
        for (int kk = k0; kk < k1; ++kk) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = mfma32(bc[j][kk >> 2][kk & 3], ac[i][kk >> 2][kk & 3], acc[i][j]);   // transposed block (EpiWide)
        }
    };

    int buf = 0;
    int it__ = 0;
    if (advance()) {
        load(true);
        MTD_STAMP(2);
        store_b(0);
        __syncthreads();
        MTD_STAMP(3);
        bool nxt;
        do {
            if (it__ < 12) MTD_STAMP(4 + 4 * it__);
            // current chunk: A fragments move out of the prefetch registers, B fragments come from LDS
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ac[i][j] = an[i][j];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const float* row = &Bs[buf][((wn * WN + j) * 32 + l31) * BLD + kh * 16];
#pragma unroll
                for (int q = 0; q < 4; ++q) bc[j][q] = *reinterpret_cast<const f32x4*>(row + 4 * q);
            }
            nxt = advance();
            lm = 0u - (unsigned)nxt;
            // next chunk's loads interleaved with this chunk's MFMAs: [2 steps][A tile 0][..][A tile 1][..][B][rest]
            __builtin_amdgcn_sched_barrier(0);
            if (it__ < 12) MTD_STAMP(5 + 4 * it__);
            mfma_steps(0, 2);
            __builtin_amdgcn_sched_barrier(0);
            load_a(0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_steps(2, 6);
            __builtin_amdgcn_sched_barrier(0);
            if (WM > 1) load_a(WM - 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_steps(6, 10);
            __builtin_amdgcn_sched_barrier(0);
            load_b();
            __builtin_amdgcn_sched_barrier(0);
            mfma_steps(10, 16);
            __builtin_amdgcn_sched_barrier(0);
            if (it__ < 12) MTD_STAMP(6 + 4 * it__);
            store_b(buf ^ 1);           // the other buffer was last read one barrier ago
            __syncthreads();
            if (it__ < 12) MTD_STAMP(7 + 4 * it__);
            ++it__;
            buf ^= 1;
        } while (nxt);
    }
    MTD_STAMP(60);

    // ---- epilogue.  The accumulator blocks are TRANSPOSED (operands swapped in mfma_steps): lane (l31, kh) holds, for its own
    //      pixel, channels nb0 + 8 g + 4 kh + j -- every operand, the result and the split-K slabs move as 16-byte vectors
    //      (EpiWide above; the dword form was 16 instructions per tensor and block, and the strided data gradients with two
    //      adds and a mask are bound by exactly that traffic).
    if (p.splitk > 1) {
        float* slab = a.ws + (long long)zk * p.M * a.N;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int m = m0 + (wm * WM + i) * 32 + l31;
                if (m < p.M) {
                    float* row = slab + (long long)m * a.N + (n0 + (wn * WN + j) * 32 + 4 * kh);
                    const bool through = MTD_IGEMM_FIN && p.fin != 0;
                    if ((p.wide & 2) && !through) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
                            *reinterpret_cast<f32x4*>(row + 8 * g4) = f32x4{acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) slab_store(row + 8 * (e >> 2) + (e & 3), acc[i][j][e], through);
                    }
                }
            }
        if constexpr (FIN) {
            if (MTD_IGEMM_FIN && p.fin && splitk_last_arrival(p, tile_m * (int)gridDim.y + tile_n)) splitk_finish_tile<BM, BN>(p, m0, n0);
        }
        return;
    }
    const ScalePair sp = load_scale(a);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int mrow0 = m0 + (wm * WM + i) * 32, nb0 = n0 + (wn * WN + j) * 32;
            if (mrow0 + l31 < p.M) {
                EpiWide wad;
                wad.init(p, mrow0, lane, nb0, sp);
                if (p.wide & 1) {
                    f32x4 bias4[4];
                    epiw_bias(a, wad.ch, bias4);
                    EpiWideOps weo;
                    epiw_load(p, wad, weo);
                    epiw_store(p, acc[i][j], wad, bias4, weo);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = wad.ch + 8 * (e >> 2) + (e & 3);
                        a.out[wad.pix * a.out_ld + n] = epilogue_value(a, acc[i][j][e], wad.sc, a.bias ? a.bias[n] : 0.f, wad.pix, n);
                    }
                }
            }
        }
    MTD_STAMP(61);
}

template <int WM, int WN, int WGM, int WGN>
__global__ __launch_bounds__(256, (WM * WN >= 4) ? 2 : 1) void igemm_kernel(const IgemmParams p) {
    igemm_body<WM, WN, WGM, WGN>(p, blockIdx.z);
}

// Several launches of ONE shape (same M, N, C, taps: the same plan) in one grid: blockIdx.z = set * splitk + split-K slice.
// Each set has its own geometry offsets, operands and outputs -- the four input-parity classes of a stride-2 data
// gradient (each a 2x2-tap stride-1 gather writing every other output pixel), which as four launches of 27-35 us ran at
// 60-79 TFLOP/s, half of them split-K with an epilogue launch each.
constexpr int MULTI_MAX = 4;
struct IgemmMulti { IgemmParams p[MULTI_MAX]; };

template <int WM, int WN, int WGM, int WGN>
__global__ __launch_bounds__(256, (WM * WN >= 4) ? 2 : 1) void igemm_multi_kernel(const IgemmMulti mp) {
    const int sk = mp.p[0].splitk;
    const int set = __builtin_amdgcn_readfirstlane(blockIdx.z / sk);
    const int zk = __builtin_amdgcn_readfirstlane(blockIdx.z - set * sk);
    // (a switch, not an index: with a dynamic index into the kernel arguments the compiler moved the tap tables of the
    // register-blocked tiles to scratch memory)
    switch (set) {
        case 0: igemm_body<WM, WN, WGM, WGN, false>(mp.p[0], zk); break;
        case 1: igemm_body<WM, WN, WGM, WGN, false>(mp.p[1], zk); break;
        case 2: igemm_body<WM, WN, WGM, WGN, false>(mp.p[2], zk); break;
        default: igemm_body<WM, WN, WGM, WGN, false>(mp.p[3], zk); break;
    }
}


// ---- tap-block variant ----------------------------------------------------------------------------
// Same operand scheme (A fragments straight from global memory, B fragments from LDS), but the weight tile staged per
// barrier covers ALL taps of one 32-channel chunk ([tap][32 n][32 c], <= 9 taps = 41 KB) instead of one tap: two
// workgroup barriers per 9 x 16 MFMAs instead of nine, none at all inside a chunk, so the four waves drift freely
// and the next chunk's weights (9 x 16 bytes per thread) are in flight during the whole chunk.  Tile 128*WM x 32.
constexpr int TB_MAXT = 9;

template <int WM>
__global__ __launch_bounds__(256, 2) void igemm_tb_kernel(const IgemmParams p) {
    constexpr int BM = 128 * WM;
    __shared__ __attribute__((aligned(16))) float Bs[TB_MAXT * 32 * BLD];
    __shared__ unsigned vmask_s;

    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    int tile_m, tile_n;
    tile_of(p, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * 32;
    const int cbeg = blockIdx.z * p.c_per_split;
    const int cend = min(a.C, cbeg + p.c_per_split);
    const int T = g.TH * g.TW;

    unsigned boff[WM];
    unsigned okmask[WM];
    unsigned anymask = 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = m0 + (wave * WM + i) * 32 + l31;
        okmask[i] = 0;
        boff[i] = 0;
        if (m < p.M) {
            int b, oy, ox;
            pix_decompose(m, g.OW, g.OH, b, oy, ox);
            const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
            boff[i] = (unsigned)(((((long long)b * g.IH + py) * g.IW + px) * a.in_ld + kh * 16) * 4);
            for (int t = 0; t < T; ++t) {
                const int iy = py + p.tap_dy[t], ix = px + p.tap_dx[t];
                if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) okmask[i] |= 1u << t;
            }
        }
        anymask |= okmask[i];
    }
    if (tid == 0) vmask_s = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anymask |= __shfl_xor(anymask, off, 64);
    if (lane == 0) atomicOr(&vmask_s, anymask);
    __syncthreads();
    const unsigned vmask = vmask_s;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);

    f32x16 acc[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // weight staging: thread -> row n = tid / 8, 16-byte vector q = tid % 8 of every tap's [32 n][32 c] tile
    const int sn = tid >> 3, sq = tid & 7;
    const int wbase = (int)((long long)(n0 + sn) * a.w_sn) + 4 * sq;
    const int lbase = sn * BLD + 4 * sq;
    f32x4 bn[TB_MAXT];
    f32x4 an[WM][4], ac[WM][4], bc[4];
    auto load_a = [&](int i, int t, int c0, bool live) {
        const unsigned voff = (((okmask[i] >> t) & 1u) && live) ? (boff[i] + (unsigned)p.tap_delta[t] + (unsigned)c0 * 4u) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            an[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 16 * j, 0));
    };
    auto mfma_steps = [&](int k0, int k1) {
#pragma unroll
        for (int kk = k0; kk < k1; ++kk)
#pragma unroll
            for (int i = 0; i < WM; ++i) acc[i] = mfma32(bc[kk >> 2][kk & 3], ac[i][kk >> 2][kk & 3], acc[i]);      // transposed block (EpiWide)
    };
    // one (tap, chunk) step; the A fragments of the following step (nt, nc) are requested in between the MFMAs
    auto step = [&](int t, int nt, int nc, bool live) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) ac[i][j] = an[i][j];
        const float* row = &Bs[t * 32 * BLD + l31 * BLD + kh * 16];
#pragma unroll
        for (int q = 0; q < 4; ++q) bc[q] = *reinterpret_cast<const f32x4*>(row + 4 * q);
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        load_a(0, nt, nc, live);
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(2, 8);
        __builtin_amdgcn_sched_barrier(0);
        if (WM > 1) load_a(WM - 1, nt, nc, live);
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(8, 16);
        __builtin_amdgcn_sched_barrier(0);
    };

    if (vmask != 0u && cbeg < cend) {
        const bool full = (vmask == ((T >= 32) ? 0xFFFFFFFFu : ((1u << T) - 1u))) && T == TB_MAXT;
        const int first = __builtin_ctz(vmask);
        // weights of the first chunk
#pragma unroll
        for (int t = 0; t < TB_MAXT; ++t)
            if (t < T) bn[t] = *reinterpret_cast<const f32x4*>(a.w + (wbase + cbeg + (int)((long long)p.tap_kidx[t] * a.w_st)));
#pragma unroll
        for (int i = 0; i < WM; ++i) load_a(i, first, cbeg, true);
        for (int c0 = cbeg; c0 < cend; c0 += KC) {
            if (c0 != cbeg) __syncthreads();            // every wave is done with the previous chunk's tiles
#pragma unroll
            for (int t = 0; t < TB_MAXT; ++t)
                if (t < T) *reinterpret_cast<f32x4*>(&Bs[t * 32 * BLD + lbase]) = bn[t];
            __syncthreads();
            // next chunk's weights (clamped on the last chunk: a harmless re-read, keeps the loads unconditional)
            const int cn = (c0 + KC < cend) ? c0 + KC : c0;
#pragma unroll
            for (int t = 0; t < TB_MAXT; ++t)
                if (t < T) bn[t] = *reinterpret_cast<const f32x4*>(a.w + (wbase + cn + (int)((long long)p.tap_kidx[t] * a.w_st)));
            const bool more = (c0 + KC < cend);
            if (full) {
#pragma unroll
                for (int t = 0; t < TB_MAXT; ++t) {
                    if (t + 1 < TB_MAXT) step(t, t + 1, c0, true);
                    else step(t, 0, c0 + KC, more);
                }
            } else {
                for (int t = 0; t < T; ++t) {
                    if (!((vmask >> t) & 1u)) continue;
                    const unsigned rem = vmask & ~((2u << t) - 1u);
                    if (rem) step(t, __builtin_ctz(rem), c0, true);
                    else step(t, first, c0 + KC, more);
                }
            }
        }
    }

    // ---- epilogue (same as igemm_body: transposed blocks, 16-byte vectors)
    if (p.splitk > 1) {
        float* slab = a.ws + (long long)blockIdx.z * p.M * a.N;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m = m0 + (wave * WM + i) * 32 + l31;
            if (m < p.M) {
                float* row = slab + (long long)m * a.N + (n0 + 4 * kh);
                const bool through = MTD_IGEMM_FIN && p.fin != 0;
                if ((p.wide & 2) && !through) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<f32x4*>(row + 8 * g4) = f32x4{acc[i][4 * g4], acc[i][4 * g4 + 1], acc[i][4 * g4 + 2], acc[i][4 * g4 + 3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) slab_store(row + 8 * (e >> 2) + (e & 3), acc[i][e], through);
                }
            }
        }
        if constexpr (WM == 1) {      // (the two-block form has no registers to spare: fill_params keeps fin off for it)
            if (MTD_IGEMM_FIN && p.fin && splitk_last_arrival(p, tile_m * (int)gridDim.y + tile_n)) splitk_finish_tile<BM, 32>(p, m0, n0);
        }
        return;
    }
    const ScalePair sp = load_scale(a);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mrow0 = m0 + (wave * WM + i) * 32;
        if (mrow0 + l31 < p.M) {
            EpiWide wad;
            wad.init(p, mrow0, lane, n0, sp);
            if (p.wide & 1) {
                f32x4 bias4[4];
                epiw_bias(a, wad.ch, bias4);
                EpiWideOps weo;
                epiw_load(p, wad, weo);
                epiw_store(p, acc[i], wad, bias4, weo);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = wad.ch + 8 * (e >> 2) + (e & 3);
                    a.out[wad.pix * a.out_ld + n] = epilogue_value(a, acc[i][e], wad.sc, a.bias ? a.bias[n] : 0.f, wad.pix, n);
                }
            }
        }
    }
}


// ---- persistent variant for C == 32, 3x3 (the generator's 32->32 convolutions and their data gradients) -------------
// All nine taps of the [32 n][32 c] weight block (41 KB with padding) are staged into LDS once per workgroup; after
// that single barrier every wave is on its own: it walks 32-pixel tiles (tile, tile + #waves, ...), nine 16-MFMA steps
// each, A fragments prefetched one tap ahead -- across the tile boundary too, so the next tile's first loads are in
// flight while this tile's epilogue stores drain.  No barrier in the steady state, no per-tile weight traffic, and the
// waves of a SIMD drift into different phases instead of running prologue / loop / epilogue in lockstep.
__global__ __launch_bounds__(256, 2) void igemm_c32p_kernel(const IgemmParams p, int ntiles) {
    constexpr int T = 9;
    __shared__ __attribute__((aligned(16))) float Bs[T * 32 * BLD];
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.y * 32;
    {
        const int sn = tid >> 3, sq = tid & 7;
        const int wbase = (int)((long long)(n0 + sn) * a.w_sn) + 4 * sq;
        f32x4 bn[T];
#pragma unroll
        for (int t = 0; t < T; ++t) bn[t] = *reinterpret_cast<const f32x4*>(a.w + (wbase + (int)((long long)p.tap_kidx[t] * a.w_st)));
#pragma unroll
        for (int t = 0; t < T; ++t) *reinterpret_cast<f32x4*>(&Bs[t * 32 * BLD + sn * BLD + 4 * sq]) = bn[t];
    }
    __syncthreads();
    const int wstride = gridDim.x * 4;
    int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    const ScalePair sp = load_scale(a);

    unsigned boff = 0, okmask = 0;
    auto setup = [&](int tl, unsigned& bo, unsigned& ok) {
        const int m = tl * 32 + l31;
        bo = 0;
        ok = 0;
        if (tl < ntiles && m < p.M) {
            int b, oy, ox;
            pix_decompose(m, g.OW, g.OH, b, oy, ox);
            const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
            bo = (unsigned)(((((long long)b * g.IH + py) * g.IW + px) * a.in_ld + kh * 16) * 4);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int iy = py + p.tap_dy[t], ix = px + p.tap_dx[t];
                if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) ok |= 1u << t;
            }
        }
    };
    // A fragments are requested PF taps ahead (a load under contention takes ~2 us; one 16-MFMA step of the two waves
    // that share a SIMD covers 0.85 us), in a ring of PF register sets; the ring runs across the tile boundary.
    constexpr int PF = 3;
    f32x4 ring[PF][4], ac[4], bc[4];
    auto load_a = [&](int slot, int t, unsigned bo, unsigned ok) {
        const unsigned voff = ((ok >> t) & 1u) ? (bo + (unsigned)p.tap_delta[t]) : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ring[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 16 * j, 0));
    };
    setup(tile, boff, okmask);
#pragma unroll
    for (int t = 0; t < PF; ++t) load_a(t, t, boff, okmask);
    while (true) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int next = tile + wstride;
        unsigned boff2 = 0, ok2 = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ac[j] = ring[t % PF][j];
            const float* row = &Bs[t * 32 * BLD + l31 * BLD + kh * 16];
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[q] = *reinterpret_cast<const f32x4*>(row + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma32(ac[kk >> 2][kk & 3], bc[kk >> 2][kk & 3], acc);
            __builtin_amdgcn_sched_barrier(0);
            if (t + PF < T) {
                load_a(t % PF, t + PF, boff, okmask);
            } else {
                if (t + PF == T) setup(next, boff2, ok2);   // past the last tile: masks are empty, the loads return zeros
                load_a(t % PF, t + PF - T, boff2, ok2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 4; kk < 16; ++kk) acc = mfma32(ac[kk >> 2][kk & 3], bc[kk >> 2][kk & 3], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue of this tile (the next tile's first fragments are already in flight)
        epilogue16<8>(p, acc, tile * 32, lane, n0 + l31, sp);
        if (next >= ntiles) break;
        tile = next;
        boff = boff2;
        okmask = ok2;
    }
}


// ---- halo-tile variant of the same layers (C == 32, 3x3, stride 1, "same" padding, 64-pixel image rows) ----------------
// The persistent kernel above re-reads its 32 pixels x 128 bytes for every tap (9 x 16.8 MB of L2 traffic per launch at
// M = 131072).  Here a workgroup of eight waves owns four image rows (256 pixels) at a time: the 6 x 66-pixel halo tile
// goes global -> LDS ONCE by LDS-DMA (out-of-image pixels are out-of-range offsets and arrive as zeros, which is the
// padding), and the nine taps read their A fragments from it -- conflict-free b128 reads through the piece ^ (pixel & 7)
// permutation, applied on the DMA's source side.  The nine [32 n][32 c] weight blocks are DMA'd once per workgroup, same
// layout.  One workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...; the next tile's DMA and this tile's epilogue
// operands are requested before the MFMA loop and waited for after it, and the output stores drain under the next
// tile's MFMAs, so memory phases and MFMA phases of a CU overlap instead of alternating chip-wide.
// LDS: 2 x 50 KB tiles + 36 KB weights.
constexpr int C32T_W = 64, C32T_R = 4, C32T_HW = C32T_W + 2;

// R image rows per tile, 2 R waves per workgroup.  DB: two halo buffers, one workgroup per CU, the next tile's DMA under this
// tile's MFMAs (R = 4).  !DB (R = 2, lab variant MTD_C32T_VARIANT=1): one halo buffer of half the size, TWO workgroups per CU
// that are meant to alternate -- one in its memory phase (epilogue stores, next tile's DMA) while the other has the MFMA
// pipes -- with the second workgroup of a CU held back by `stagger` x 64 clocks at the start.
//
// SPEC (R = 4, DB, WIDE; mtd_resfft_block_tail): the launch is the tail of a Res-FFT-Conv block (arch/Ours/networks.py:32-36).
// The tile's rows are COMPLETE image rows, so the inverse row transform of the spectral branch -- mtd_irfft_rows, a launch of
// its own that re-read x and the conv branch -- rides on the same accumulator layout as a 33-step GEMM on the matrix cores:
//     irfft_rows[p][c] = sum_kw  D[p][kw, re] * T[kw][re][c] + D[p][kw, im] * T[kw][im][c],
//     D[p][kw, re] = w cos(2 pi kw p / 64) / 8,  D[p][kw, im] = -w sin(2 pi kw p / 64) / 8,  w = 1 for kw in {0, 32}, else 2
// (the Hermitian half folded into w; the sines vanish at kw = 0 and 32, which IS torch.fft.irfft2's "imaginary parts of columns
// 0 and W/2 are ignored").  A wave's lanes already are (pixel l31, k half kh) for the conv; the spectrum row
// T[b][kw][h][kh][c = l31] is the other operand, one coalesced 256-byte load per kw requested before the conv's MFMA loop.
// 33 MFMAs on top of the conv's 144 per tile and wave -- no barrier, no LDS round trip, no vector-ALU transform (a first form
// with a quad-split register FFT between the MFMA loop and the epilogue cost two barriers and ~5 us per tile).  D (64 x 66
// floats) sits in LDS; x is read from the halo tile:   out2 = act(conv + bias)   out = out2 + x + irfft_rows(specT).
template <int R, bool DB, bool WIDE = false, bool SPEC = false>
__global__ __launch_bounds__(128 * R, DB ? 1 : 2) void igemm_c32t_kernel(const IgemmParams p, int ntiles, int stagger,
                                                                         const float* __restrict__ specT) {
    static_assert(!SPEC || (R == 4 && DB && WIDE), "the spectral tail needs four-row tiles, two halo buffers and the wide epilogue");
    constexpr int T = 9, NW = 2 * R, HP = (R + 2) * C32T_HW, NI = (HP + 7) / 8;
    // The two halo buffers are separate LDS OBJECTS and the tile loop is unrolled by two: with one array indexed by `cur` the
    // compiler cannot tell the pending LDS-DMA of the NEXT tile from one into the buffer it is about to read, and put a
    // vmcnt(0) in front of every tile's first LDS read -- the double buffering overlapped nothing.
    __shared__ __attribute__((aligned(1024))) float Hs0[NI * 8 * 32];
    __shared__ __attribute__((aligned(1024))) float Hs1[DB ? NI * 8 * 32 : 64];
    __shared__ __attribute__((aligned(1024))) float Bs[T * 32 * 32];
    constexpr int DLD = 67;                                       // row stride of the inverse-DFT matrix (odd: conflict-free over l31)
    __shared__ float Ds[SPEC ? 64 * DLD : 1];
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.y * 32;
    const int tiles_per_image = g.OH / R;
    typedef __attribute__((address_space(3))) float lds_f;
    if (SPEC) {
        for (int e = tid; e < 64 * 66; e += 128 * R) {
            const int px = e / 66, kap = e - px * 66, kw = kap >> 1;
            const int ang = (kw * px) & 63;                                   // angle 2 pi ang / 64: the transforms' own tables
            const float tv = (kap & 1) ? SIN64[ang & 31] : COS64[ang & 31];
            const float wgt = (kw == 0 || kw == 32) ? 0.125f : 0.25f;
            Ds[px * DLD + kap] = (((ang & 32) != 0) != ((kap & 1) != 0)) ? -wgt * tv : wgt * tv;      // cos, sin (t + pi) = -cos, -sin t; the sine enters negated
        }
    }
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    const int rsub = lane >> 3, piece = (lane & 7) ^ rsub;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SPEC ? specT : a.in), (short)0,
                                                                         SPEC ? g.B * NKW * 16384 : 0, 0x00020000);

    // DMA instruction i of a tile moves halo pixels 8i .. 8i+7: lane L -> pixel 8i + (L >> 3), LDS piece L & 7
    auto stage_tile = [&](int tile, float* Hd) {
        const int b = tile / tiles_per_image;
        const int oy0 = (tile - b * tiles_per_image) * R;
        for (int i = wave; i < NI; i += NW) {
            const int hp = 8 * i + rsub;
            const int hr = hp / C32T_HW, hc = hp - hr * C32T_HW;
            const int iy = oy0 - 1 + hr, ix = hc - 1;
            const bool ok = (hp < HP) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
            const unsigned voff = ok ? (unsigned)(((((long long)b * g.IH + iy) * g.IW + ix) * a.in_ld + piece * 4) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_f*)(Hd + i * 256), 16, voff, 0, 0, 0);
        }
    };
    MTD_STAMP(0);
    int tile = blockIdx.x;
    stage_tile(tile, Hs0);
    {   // weights: row r = tap * 32 + n of a [288][32 c] matrix, same piece permutation
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), (short)0, (int)p.w_bytes, 0x00020000);
        for (int i = wave; i < T * 4; i += NW) {
            const int t = i >> 2, nn = 8 * (i & 3) + rsub;
            const unsigned voff = (unsigned)(((long long)(n0 + nn) * a.w_sn + (long long)p.tap_kidx[t] * a.w_st + piece * 4) * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_f*)&Bs[i * 256], 16, voff, 0, 0, 0);
        }
    }
    MTD_STAMP(1);
    const ScalePair sp = load_scale(a);
    const int n = n0 + l31;
    const float bias_n = a.bias ? a.bias[n] : 0.f;
    f32x4 bias4[4];
    epiw_bias(a, n0 + 4 * kh, bias4);
    const int bsw = l31 & 7;
    // this wave's 32 pixels of a tile: tile row wave >> 1, columns 32 * (wave & 1) + l31; halo index of the pixel itself:
    const int hp0 = ((wave >> 1) + 1) * C32T_HW + (wave & 1) * 32 + l31 + 1;
    if (!DB && stagger > 0 && (int)blockIdx.x >= ((int)gridDim.x >> 1)) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);      // 64 clocks each
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MTD_STAMP(2);
    // one tile from halo buffer H (DB: the next tile's DMA goes to Hn meanwhile); false = that was the last tile
    auto one_tile = [&](float* H, float* Hn, int cur) -> bool {
        const int next = tile + gridDim.x;
        if (DB && next < ntiles) stage_tile(next, Hn);           // every wave is past its MFMAs on that buffer (barrier below)
        const int mbase = tile * (R * C32T_W) + wave * 32;
        EpiAddr<true, 0, 16> ead;
        EpiOps<16> eo;
        EpiWide wad;
        EpiWideOps weo;
        // SPEC: this wave's spectrum row as MFMA operand fragments, tb[kw] = T[b][kw][h][kh][c = l31] (one lane offset + a
        // scalar offset per load)
        float tb[NKW];
        if (SPEC) {
            const int b = tile / tiles_per_image;
            const int h = (tile - b * tiles_per_image) * R + (wave >> 1);
            const unsigned base = (unsigned)((((b * NKW) * 64 + h) * 64 + kh * 32 + l31) * 4);
#pragma unroll
            for (int kw = 0; kw < NKW; ++kw)
                tb[kw] = (stagger & 2) ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(trs, base, kw * 16384, 0));
            wad.init(p, mbase, lane, n0, sp);
        } else if (WIDE) {
            wad.init(p, mbase, lane, n0, sp);
            epiw_load(p, wad, weo);
        } else {
            ead.init(p, mbase, lane, n);
            epi_load(p, ead, eo);
        }
        __builtin_amdgcn_sched_barrier(0);
        auto frag = [&](int t, f32x4* af, f32x4* bf) {
            int hp = hp0 + (g.off_y + p.tap_dy[t]) * C32T_HW + (g.off_x + p.tap_dx[t]);
            // SPEC: the swizzled piece addresses are re-derived per tap (about eight VALU operations under sixteen 64-clock MFMAs).
            // Left to itself the compiler keeps all 9 taps x 4 pieces x 2 halo buffers = 72 address registers live across the
            // whole tile loop, which with the spectrum's 32 registers no longer fits the 256 of two waves per SIMD.
            if (SPEC) asm volatile("" : "+v"(hp));
            const float* px = &H[hp * 32];
            const int sw = hp & 7;
            const float* row = &Bs[(t * 32 + l31) * 32];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                af[q] = *reinterpret_cast<const f32x4*>(px + (((kh * 4 + q) ^ sw) << 2));
                bf[q] = *reinterpret_cast<const f32x4*>(row + (((kh * 4 + q) ^ bsw) << 2));
            }
        };
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        f32x4 af[2][4], bf[2][4];
        frag(0, af[0], bf[0]);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) frag(t + 1, af[(t + 1) & 1], bf[(t + 1) & 1]);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                acc = WIDE ? mfma32(bf[t & 1][kk >> 2][kk & 3], af[t & 1][kk >> 2][kk & 3], acc)      // transposed block (EpiWide)
                           : mfma32(af[t & 1][kk >> 2][kk & 3], bf[t & 1][kk >> 2][kk & 3], acc);
        }
        f32x16 acc2;
        if (SPEC) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
            const float* drow = &Ds[((wave & 1) * 32 + l31) * DLD + kh];
            if (!(stagger & 1))
#pragma unroll
            for (int kw = 0; kw < NKW; ++kw) acc2 = mfma32(tb[kw], drow[2 * kw], acc2);      // (channel, pixel) block like the conv's
        }
        MTD_STAMP(3 + 3 * (cur));
        if (DB) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile and this tile's epilogue operands have landed
            MTD_STAMP(4 + 3 * (cur));
            if (SPEC) {
                const float* px = &H[hp0 * 32];
                const int sw = hp0 & 7;
                f32x4 xs[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    xs[g] = *reinterpret_cast<const f32x4*>(px + (((kh + 2 * g) ^ sw) << 2));      // x: the pixel itself, in the halo tile
#pragma unroll
                    for (int j = 0; j < 4; ++j) xs[g][j] += acc2[4 * g + j];
                }
                epiw_store_spec(p, acc, wad, bias4, xs);
            } else if (WIDE) epiw_store(p, acc, wad, bias4, weo);
            else epi_store(p, acc, ead, sp, bias_n, eo);           // stores drain under the next tile's MFMAs
            MTD_STAMP(5 + 3 * (cur));
            if (next >= ntiles) return false;
            __syncthreads();
            tile = next;
        } else {
            __syncthreads();                                       // every wave is done reading the halo tile
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's epilogue operands have landed
            if (next < ntiles) stage_tile(next, Hs0);              // the next tile's DMA in flight under the stores ...
            if (WIDE) epiw_store(p, acc, wad, bias4, weo);
            else epi_store(p, acc, ead, sp, bias_n, eo);           // ... and under the other workgroup's MFMAs
            if (next >= ntiles) return false;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            tile = next;
        }
        return true;
    };
    if (DB) {
        while (one_tile(Hs0, Hs1, 0) && one_tile(Hs1, Hs0, 1)) {}
    } else {
        while (one_tile(Hs0, Hs0, 0)) {}
    }
}

// 16-byte epilogue vectors (EpiWide): every operand row 16-byte aligned
bool wide_epilogue_ok(const mtd_conv_args& a) {
    if (!aligned16(a.out) || (a.out_ld % 4)) return false;
    if (a.bias && !aligned16(a.bias)) return false;
    if (a.add1 && (!aligned16(a.add1) || (a.add1_ld % 4))) return false;
    if (a.add2 && (!aligned16(a.add2) || (a.add2_ld % 4))) return false;
    if (a.mask && (!aligned16(a.mask) || (a.mask_ld % 4))) return false;
    if (a.out2 && (!aligned16(a.out2) || (a.out2_ld % 4))) return false;
    return true;
}

// the halo-tile kernel's geometry: 3x3, stride 1, every tap within one pixel of the output position, 64-pixel rows
bool c32t_eligible(const mtd_conv_args& a) {
    const mtd_geom& g = a.g;
    if (a.C != 32 || g.TH != 3 || g.TW != 3 || g.in_sy != 1 || g.in_sx != 1) return false;
    if (g.OW != C32T_W || g.IW != C32T_W || g.IH != g.OH || (g.OH % C32T_R)) return false;
    if (!(g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW)) return false;
    for (int i = 0; i < 3; ++i) {
        const int dy = g.off_y + i * g.tap_dy, dx = g.off_x + i * g.tap_dx;
        if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    }
    return true;
}


// ---- v2: 128 x 128 workgroup tile, both operands through LDS by LDS-DMA ---------------------------------------------
// The 3x3 layers are bound by L2 bandwidth in the kernels above: every 32-wide n-tile re-reads its activation tile for
// each of the nine taps (measured: the generator conv's loads alone take 20 us = 7.4 TB/s of L2 traffic).  Here one
// workgroup owns 128 pixels x 128 output channels, so a (tap, 32-channel) step moves 16 KB of activations and 16 KB of
// weights for 256 MFMAs -- 4x less L2 traffic per flop.  Both tiles go global -> LDS with buffer_load_dwordx4 ... lds
// (no staging registers; out-of-image pixels are out-of-range offsets and arrive as zeros), double-buffered, one
// workgroup barrier per step with the next step's DMA in flight under the current step's MFMAs.
// LDS image of a tile: row r (pixel or output channel) = 128 bytes = 8 pieces of 16 bytes; piece q sits at position
// q ^ (r & 7).  The DMA destination is lane-linear, so the permutation is applied on the source side (lane L of an
// instruction covering rows 8i..8i+7 fetches piece (L & 7) ^ (L >> 3) of row 8i + (L >> 3)); fragment reads
// (lane = row, 4 x ds_read_b128) are then conflict-free.  Wave tile 64 x 64 (2 x 2 accumulators).
template <int DUMMY>
__global__ __launch_bounds__(256, 2) void igemm_v2_kernel(const IgemmParams p) {
    __shared__ __attribute__((aligned(1024))) float Ls[2][2][128 * 32];      // [buffer][A | B][row][32]
    const mtd_conv_args& a = p.a;
    const mtd_geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
    const int cbeg = blockIdx.z * p.c_per_split;
    const int cend = min(a.C, cbeg + p.c_per_split);
    const int T = g.TH * g.TW;
    const int nchunk = (cend - cbeg) / KC;
    const int S = T * nchunk;

    // rows this lane moves in every step: 8 * (wave * 4 + ii) + (lane >> 3), piece (lane & 7) ^ (lane >> 3)
    const int rsub = lane >> 3;
    const int piece = (lane & 7) ^ rsub;
    unsigned aoff[4], okm[4], woff[4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
        const int r = 8 * (wave * 4 + ii) + rsub;
        const int m = m0 + r;
        aoff[ii] = 0;
        okm[ii] = 0;
        if (m < p.M) {
            int b, oy, ox;
            pix_decompose(m, g.OW, g.OH, b, oy, ox);
            const int py = oy * g.in_sy + g.off_y, px = ox * g.in_sx + g.off_x;
            aoff[ii] = (unsigned)(((((long long)b * g.IH + py) * g.IW + px) * a.in_ld + piece * 4) * 4);
            for (int t = 0; t < T; ++t) {
                const int iy = py + p.tap_dy[t], ix = px + p.tap_dx[t];
                if (((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW)) okm[ii] |= 1u << t;
            }
        }
        woff[ii] = (unsigned)(((long long)(n0 + r) * a.w_sn + piece * 4) * 4);
    }
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), (short)0, (int)p.w_bytes, 0x00020000);

    typedef __attribute__((address_space(3))) float lds_f;
    auto stage = [&](int s, int buf) {
        const int t = s / nchunk, c0 = cbeg + (s - t * nchunk) * KC;
        const unsigned adelta = (unsigned)p.tap_delta[t] + (unsigned)c0 * 4u;
        const unsigned wdelta = (unsigned)(((long long)p.tap_kidx[t] * a.w_st + c0) * 4);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const unsigned voff = ((okm[ii] >> t) & 1u) ? aoff[ii] + adelta : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_f*)&Ls[buf][0][(wave * 4 + ii) * 256], 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_f*)&Ls[buf][1][(wave * 4 + ii) * 256], 16, woff[ii] + wdelta, 0, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses (floats) inside a tile: row * 32 + ((kh * 4 + q) ^ (row & 7)) * 4
    int afr[2], bfr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        afr[i] = (wm * 64 + i * 32 + l31) * 32;
        bfr[i] = (wn * 64 + i * 32 + l31) * 32;
    }
    const int sw = l31 & 7;

    if (S > 0) stage(0, 0);
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA of step s has landed
        __syncthreads();                                      // everyone's has; everyone is done reading the other buffer
        if (s + 1 < S) stage(s + 1, buf ^ 1);
        f32x4 af[2][4], bf[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                af[i][q] = *reinterpret_cast<const f32x4*>(&Ls[buf][0][afr[i] + (((kh * 4 + q) ^ sw) << 2)]);
                bf[i][q] = *reinterpret_cast<const f32x4*>(&Ls[buf][1][bfr[i] + (((kh * 4 + q) ^ sw) << 2)]);
            }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i][kk >> 2][kk & 3], bf[j][kk >> 2][kk & 3], acc[i][j]);
    }

    // ---- epilogue
    if (p.splitk > 1) {
        float* slab = a.ws + (long long)blockIdx.z * p.M * a.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 64 + i * 32 + mfma32_row(e, lane);
                    if (m < p.M) slab_store(&slab[(long long)m * a.N + n], acc[i][j][e], MTD_IGEMM_FIN && p.fin != 0);
                }
            }
        if (MTD_IGEMM_FIN && p.fin && splitk_last_arrival(p, (int)(blockIdx.x * gridDim.y + blockIdx.y))) splitk_finish_tile<128, 128>(p, m0, n0);
        return;
    }
    const ScalePair sp = load_scale(a);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            epilogue16<4>(p, acc[i][j], m0 + wm * 64 + i * 32, lane, n0 + wn * 64 + j * 32 + l31, sp);
        }
}

// sum the split-K slabs in order, then the same epilogue (value by value: any alignment)
__global__ __launch_bounds__(256) void splitk_epilogue_scalar_kernel(const IgemmParams p) {
    const mtd_conv_args& a = p.a;
    const long long total = (long long)p.M * a.N;
    const ScalePair sp = load_scale(a);
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        splitk_finish_scalar<false>(p, sp, (int)(idx / a.N), (int)(idx % a.N), total);
}

// Four consecutive output channels of one pixel per thread (splitk_finish_vec4).  Needs every row stride a multiple of
// 4 floats and 16-byte aligned bases (splitk_vec_ok); total < 2^31.
__device__ __forceinline__ void splitk_epilogue_body(const IgemmParams& p) {
    const mtd_conv_args& a = p.a;
    const unsigned total4 = (unsigned)(((long long)p.M * a.N) >> 2);
    const long long total = (long long)p.M * a.N;
    const unsigned n4n = (unsigned)a.N >> 2;
    const ScalePair sp = load_scale(a);
    for (unsigned i4 = blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += gridDim.x * 256) {
        const unsigned m = i4 / n4n;
        splitk_finish_vec4<false>(p, sp, m, (i4 - m * n4n) << 2, total);
    }
}

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const IgemmParams p) { splitk_epilogue_body(p); }
__global__ __launch_bounds__(256) void splitk_epilogue_multi_kernel(const IgemmMulti mp) { splitk_epilogue_body(mp.p[blockIdx.y]); }

bool splitk_vec_ok(const mtd_conv_args& a, long long M) {
    if (M * a.N >= (1ll << 31) || !aligned16(a.ws) || !aligned16(a.out) || (a.out_ld % 4)) return false;
    if (a.bias && !aligned16(a.bias)) return false;
    if (a.add1 && (!aligned16(a.add1) || (a.add1_ld % 4))) return false;
    if (a.add2 && (!aligned16(a.add2) || (a.add2_ld % 4))) return false;
    if (a.mask && (!aligned16(a.mask) || (a.mask_ld % 4))) return false;
    return true;
}

struct Plan { int cfg, BM, BN, splitk, c_per_split; };

int g_force_cfg = -1, g_force_split = -1;      // tuning hook (mtd_conv_igemm_override)
constexpr int NCFG = 9;
const int kCfgBM[NCFG] = {256, 128, 256, 64, 128, 32, 128, 256, 128};
const int kCfgBN[NCFG] = {32, 32, 64, 64, 128, 128, 32, 32, 128};

Plan make_plan(const mtd_conv_args& a, int sets = 1) {
    const long long M = geom_pixels(a.g) * sets;        // tile choice by the pixels of the whole grid (all sets)
    Plan pl{};
    if (g_force_cfg >= 0 && g_force_cfg < NCFG && a.N % kCfgBN[g_force_cfg] == 0 && (g_force_cfg < 6 || g_force_cfg == 8 || a.g.TH * a.g.TW <= TB_MAXT)) {
        pl.cfg = g_force_cfg; pl.BM = kCfgBM[pl.cfg]; pl.BN = kCfgBN[pl.cfg];
        int chunks = a.C / KC;
        int sk = g_force_split > 0 ? g_force_split : 1;
        if (sk > chunks) sk = chunks;
        int cps = (chunks + sk - 1) / sk;
        pl.splitk = (chunks + cps - 1) / cps;
        pl.c_per_split = cps * KC;
        return pl;
    }
    // Derived from the standalone sweep of all 109 conv shapes of the training step (tools/tune_igemm.py, profiles/):
    // fp32 MFMA is slow enough (64 clk per 32x32x2) that one 32x32 accumulator tile per wave at high occupancy beats the
    // register-blocked tiles almost everywhere; the wide tiles only pay for the huge-M, thin-K first-stage layers.
    pl.cfg = 1;
    if ((M >= 131072 && a.N >= 64) || (M >= 32768 && a.N >= 256 && a.C <= 64) || (M >= 65536 && a.N >= 128) ||
        (M >= 16384 && a.N >= 512 && a.C <= 128)) pl.cfg = 0;
    else if (M >= 32768 && a.N == 64 && a.C >= 128) pl.cfg = 3;
    // tap-block kernel (all taps of a channel chunk per barrier; 41 KB of LDS = 3 workgroups per CU): 5-9 % faster on the
    // paired-pass shapes (profiles/r1_igemm_tile_sweep.txt) when several chunks stream and the grid fits one round of residency
    if (pl.cfg == 1 && a.C >= 128 && M >= 4096 && a.g.TH * a.g.TW == TB_MAXT) {        // (1x1 layers: 30-40 % slower there)
        const long long b6 = ((M + 127) / 128) * (a.N / 32);
        const long long sk6 = b6 <= 256 ? 512 / b6 : 1;
        if (b6 * sk6 <= 768) pl.cfg = 6;
    }
    // Round 2, after the transposed accumulator blocks (the 256 x 64 tile lost its scratch spill and both tap-block forms their
    // dword epilogues): the standalone sweep of the step's 98 shapes (profiles/r2_igemm_tile_sweep.txt) puts the 256 x 64 tile
    // 5-9 % ahead on every large 3x3 grid, 1.25 ms per step over all shapes.  INSIDE the step the rules below (MTD_IGEMM_PLAN=2)
    // move 7.0 ms of launches onto that tile and 1.6 ms onto the two-block tap-block kernel and the family's total does not
    // change (20.63 -> 20.58 ms in the one-stream trace, step 40.26 vs 40.40 ms): standalone timings on repeated launches
    // do not predict the in-step ranking at this margin.  Off by default.
    static const int env_plan = [] { const char* e = mtd_lab_env("MTD_IGEMM_PLAN"); return e ? atoi(e) : 1; }();
    const int T9 = a.g.TH * a.g.TW == TB_MAXT;
    if (env_plan >= 2 && sets == 1 && T9) {
        const long long MN = M * a.N;
        if ((a.N % 64) == 0 && a.C >= 64 && MN >= (4ll << 20) && !(M >= 65536 && a.N >= 256 && a.C <= 64)) {
            pl.cfg = 2;         // 256 x 64, four blocks per wave: 5-9 % over the 128 / 256 x 32 tiles on every grid this large
        } else if (a.C >= 128 && MN >= (2ll << 20) && MN < (4ll << 20) && (M / 256) * (a.N / 32) >= 256 && a.N <= 256) {
            pl.cfg = 7;         // tap-block kernel with two blocks per wave: 16384 x 128, 32768 x 64, 8192 x 256
        }
    }
    // Round 5: the four-class stride-2 data gradients (sets == 4, 2 x 2 taps) had kept round 2's tiles; re-timed on the step's shapes
    // (tools/s2_dgrad_probe.py, us per launch, plan -> new): 65536 x 4 pixels, 64 channels 119 -> 108 (256 x 64 tile); 16384 x 4, 128:
    // 90 -> 87; 4096 x 4, 256: 95 -> 79 (256 x 32); the G step's unpaired passes 32768 x 4, 64: 61 -> 57; 8192 x 4, 128: 58 -> 48;
    // 2048 x 4, 256: 60 -> 44 (128 x 32 WITHOUT the split of K).  The 512-channel levels keep the plan.
    bool unsplit = false;
    if (S2DG_PLAN && sets == 4 && a.g.TH * a.g.TW == 4) {
        if ((a.N == 64 && M >= 131072) || (a.N == 128 && M >= 32768)) { pl.cfg = 2; unsplit = true; }
        else if (a.N == 256 && M >= 16384) { pl.cfg = 0; unsplit = true; }
        else if (a.N == 256 && M >= 8192) { pl.cfg = 1; unsplit = true; }
    }
    // ... and the first 4 x 4 stride-2 forward conv (down1: 64 -> 64 channels; tools/s2_fwd_probe.py): 65536 pixels 99 -> 82 us on the
    // 256 x 64 tile, the G step's 32768 pixels 64 -> 45 on the 64 x 64 tile, both unsplit; the deeper levels keep the plan (it is the best there)
    if (S2DG_PLAN && sets == 1 && a.g.TH * a.g.TW == 16 && a.N == 64 && a.C == 64) {
        if (M >= 65536) { pl.cfg = 2; unsplit = true; }
        else if (M >= 32768) { pl.cfg = 3; unsplit = true; }
    }
    pl.BM = kCfgBM[pl.cfg];
    pl.BN = kCfgBN[pl.cfg];
    const long long Mset = geom_pixels(a.g);
    long long blocks = ((Mset + pl.BM - 1) / pl.BM) * (a.N / pl.BN) * sets;
    int chunks = a.C / KC;
    int sk = blocks <= 256 ? (int)(512 / blocks) : 1;      // fill ~2 workgroups per CU; never split a grid that already does
    if (blocks > 256 && blocks <= 512 && chunks * a.g.TH * a.g.TW <= 32) sk = 2;      // ... unless its workgroups are short (2x2-tap data gradients: -22 %)
    if (env_plan >= 2 && sets == 1 && T9 && blocks == 256) {
        // a grid of exactly one workgroup per CU: the register-blocked tiles do not want the split at all, the tap-block
        // kernel only when its K loop is long (4096 x 256 x 256: 45 us unsplit, 51 split; 2048 x 512 x 512: 91 / 87)
        if (pl.cfg == 2 || pl.cfg == 7) sk = 1;
        else if (pl.cfg == 6 && (long long)a.C * 9 < 4096) sk = 1;
    }
    if (unsplit) sk = 1;
    if (sk > chunks) sk = chunks;
    if (sk > 32) sk = 32;
    if (sk < 1) sk = 1;
    int cps = ((chunks + sk - 1) / sk);
    sk = (chunks + cps - 1) / cps;
    pl.splitk = sk;
    pl.c_per_split = cps * KC;
    return pl;
}

int check_args(const mtd_conv_args& a) {
    if (!a.in || !a.w || !a.out) return MTD_EINVAL;
    if (a.C <= 0 || a.N <= 0 || (a.C % 32) || (a.N % 32)) return MTD_EINVAL;
    const mtd_geom& g = a.g;
    if (g.B <= 0 || g.IH <= 0 || g.IW <= 0 || g.OH <= 0 || g.OW <= 0) return MTD_EINVAL;
    if (g.TH <= 0 || g.TW <= 0 || g.TH * g.TW > 16) return MTD_EINVAL;
    if (geom_pixels(g) > (1ll << 30)) return MTD_EINVAL;
    if (a.in_ld < a.C || a.out_ld < a.N || (a.in_ld % 4)) return MTD_EINVAL;
    if (!aligned16(a.in)) return MTD_EALIGN;
    if (a.w_sc != 1 || a.w_st < 0) return MTD_EINVAL;                     // packed / natively c-contiguous weight view
    if (!aligned16(a.w) || (a.w_sn % 4) || (a.g.TH * a.g.TW > 1 && (a.w_st % 4))) return MTD_EALIGN;
    if (a.add1 && a.add1_ld < a.N) return MTD_EINVAL;
    if (a.add2 && a.add2_ld < a.N) return MTD_EINVAL;
    if (a.mask && a.mask_ld < a.N) return MTD_EINVAL;
    if (a.out2 && a.out2_ld < a.N) return MTD_EINVAL;
    // the furthest output pixel must stay inside the OHF x OWF image
    if ((g.OH - 1) * g.out_sy + g.out_oy >= g.OHF || (g.OW - 1) * g.out_sx + g.out_ox >= g.OWF) return MTD_EINVAL;
    return MTD_OK;
}

}  // namespace

#ifndef MTD_NO_API      // (conv_c32_bwd.hip includes this file for its kernels and helpers only)
extern "C" int mtd_conv_igemm_override(int cfg, int splitk) {
    g_force_cfg = cfg;
    g_force_split = splitk;
    return MTD_OK;
}

// Does mtd_conv_igemm take these arguments with act = MTD_ACT_RELU_ADD?  (The persistent kernel of the generator-shaped layers:
// its two epilogue forms implement it; no split-K, no mask, no second output.)
extern "C" int mtd_conv_relu_add_ok(const mtd_conv_args* a) {
    if (!a || check_args(*a) != MTD_OK || a->out2 || a->mask) return 0;
    const bool gen_shape = a->C == 32 && a->g.TH * a->g.TW == 9 && geom_pixels(a->g) >= 32768;
    return (g_force_cfg == -1 || g_force_cfg == 9) && gen_shape ? 1 : 0;
}

extern "C" size_t mtd_conv_igemm_ws_bytes(const mtd_conv_args* a) {
    if (!a || check_args(*a) != MTD_OK) return 0;
    Plan pl = make_plan(*a);
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * (size_t)geom_pixels(a->g) * a->N * sizeof(float);
}

#endif  // MTD_NO_API

namespace {

// everything of IgemmParams that follows from the arguments and the plan
int fill_params(const mtd_conv_args* a, const Plan& pl, IgemmParams& p) {
    p.a = *a;
    p.M = (int)geom_pixels(a->g);
    p.splitk = pl.splitk;
    p.c_per_split = pl.c_per_split;
    {
        const long long npix = (long long)a->g.B * a->g.IH * a->g.IW;
        const long long bytes = ((npix - 1) * a->in_ld + a->C) * 4;
        if (bytes >= (1ll << 31)) return MTD_EINVAL;      // 32-bit buffer offsets (largest tensor of the step: 64 MiB)
        p.in_bytes = (unsigned)bytes;
    }
    {
        const mtd_geom& gg = a->g;
        for (int t = 0; t < gg.TH * gg.TW; ++t) {
            const int ty = t / gg.TW, tx = t % gg.TW;
            p.tap_dy[t] = ty * gg.tap_dy;
            p.tap_dx[t] = tx * gg.tap_dx;
            p.tap_delta[t] = (int)((((long long)(ty * gg.tap_dy) * gg.IW + tx * gg.tap_dx) * a->in_ld) * 4);
            p.tap_kidx[t] = (gg.ky0 + ty * gg.ky_step) * gg.KW + (gg.kx0 + tx * gg.kx_step);
        }
        // weight element offsets are formed in 32 bits
        const long long wmax = (long long)(a->N - 1) * a->w_sn + (long long)(a->C - 1) + 16ll * a->w_st + 16;
        if (wmax >= (1ll << 31)) return MTD_EINVAL;
        int kmax = 0;
        for (int t = 0; t < gg.TH * gg.TW; ++t) kmax = p.tap_kidx[t] > kmax ? p.tap_kidx[t] : kmax;
        const long long wb = ((long long)(a->N - 1) * a->w_sn + (long long)kmax * a->w_st + a->C) * 4;
        p.w_bytes = wb >= (1ll << 31) ? 0x7FFFFFFFu : (unsigned)wb;
    }
    const mtd_geom& g = a->g;
    p.out_identity = (g.out_sy == 1 && g.out_sx == 1 && g.out_oy == 0 && g.out_ox == 0 && g.OHF == g.OH && g.OWF == g.OW);
    p.out_linear = p.out_identity || (g.OW % 32 == 0);
    static const int env_xcd = [] { const char* e = mtd_lab_env("MTD_IGEMM_XCD"); return e ? atoi(e) : 1; }();
    p.xcd_map = env_xcd;
    static const int env_nt = [] { const char* e = mtd_lab_env("MTD_IGEMM_NT"); return e ? atoi(e) : 0; }();
    p.nt_store = env_nt;
    p.wide = (wide_epilogue_ok(*a) ? 1 : 0) | ((pl.splitk > 1 && aligned16(a->ws)) ? 2 : 0);
    p.fin = 0;
    if (pl.splitk > 1) {
        size_t need = (size_t)pl.splitk * (size_t)p.M * a->N * sizeof(float);
        if (!a->ws || a->ws_bytes < need) return MTD_EWS;
        // finish inside the kernel when the caller brought arrival counters for every output tile (MTD_SPLITK_FIN=0: lab
        // switch back to the separate epilogue launch)
        static const int env_fin = [] { const char* e = mtd_lab_env("MTD_SPLITK_FIN"); return e ? atoi(e) : 1; }();
        const long long tiles = (long long)((p.M + pl.BM - 1) / pl.BM) * (a->N / pl.BN);
        static const int env_fin_max = [] { const char* e = mtd_lab_env("MTD_SPLITK_FIN_MAX"); return e ? atoi(e) : 8; }();
        p.fin = (MTD_IGEMM_FIN && env_fin && pl.cfg != 7 && a->tile_ctr && tiles <= (long long)a->tile_ctr_len && pl.splitk <= env_fin_max) ? (splitk_vec_ok(*a, p.M) ? 1 : 2) : 0;
    }
    return MTD_OK;
}

double algorithmic_bytes(const mtd_conv_args* a) {
    // input image, weight taps, result, epilogue operands -- each once
    return 4.0 * ((double)a->g.B * a->g.IH * a->g.IW * a->C + (double)a->g.TH * a->g.TW * a->N * a->C +
                  (double)geom_pixels(a->g) * a->N * (1 + (a->add1 != nullptr) + (a->add2 != nullptr) + (a->mask != nullptr) + (a->out2 != nullptr)));
}

}  // namespace

#ifndef MTD_NO_API
extern "C" int mtd_conv_igemm(const mtd_conv_args* a, void* stream) {
    if (!a) return MTD_EINVAL;
    int rc = check_args(*a);
    if (rc != MTD_OK) return rc;
    Plan pl = make_plan(*a);
    IgemmParams p;
    rc = fill_params(a, pl, p);
    if (rc != MTD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const double alg_bytes = algorithmic_bytes(a);      // profiler record
    const bool gen_shape = a->C == 32 && a->g.TH * a->g.TW == 9 && p.M >= 32768;
    if ((g_force_cfg == -1 || g_force_cfg == 10) && gen_shape && c32t_eligible(*a) && a->act != MTD_ACT_RELU_ADD) {
        // generator-shaped layers on 64-pixel rows: halo tiles of four image rows, one persistent workgroup per CU
        const int prof = mtd_prof_begin(0, 10, 1, p.M, a->N, a->C, 9, s, alg_bytes);
        static const int env_variant = [] { const char* e = mtd_lab_env("MTD_C32T_VARIANT"); return e ? atoi(e) : 0; }();
        static const int env_stagger = [] { const char* e = mtd_lab_env("MTD_C32T_STAGGER"); return e ? atoi(e) : 0; }();
        if (env_variant == 1) {
            const int ntiles = p.M / (2 * C32T_W);
            MTD_LAUNCH((igemm_c32t_kernel<2, false>), dim3(ntiles < 512 ? ntiles : 512, a->N / 32), dim3(256), 0, s, p, ntiles, env_stagger, (const float*)nullptr);
        } else {
            const int ntiles = p.M / (C32T_R * C32T_W);
            static const int env_wide = [] { const char* e = mtd_lab_env("MTD_C32T_WIDE"); return e ? atoi(e) : 1; }();
            if (env_wide && wide_epilogue_ok(*a))
                MTD_LAUNCH((igemm_c32t_kernel<C32T_R, true, true>), dim3(ntiles < 256 ? ntiles : 256, a->N / 32), dim3(512), 0, s, p, ntiles, 0, (const float*)nullptr);
            else
            MTD_LAUNCH((igemm_c32t_kernel<C32T_R, true>), dim3(ntiles < 256 ? ntiles : 256, a->N / 32), dim3(512), 0, s, p, ntiles, 0, (const float*)nullptr);
        }
        mtd_prof_end(prof, s);
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    if (a->out2) return MTD_EINVAL;               // second output: halo-tile kernel only
    if (a->act == MTD_ACT_RELU_ADD && !((g_force_cfg == -1 || g_force_cfg == 9) && gen_shape && !a->mask)) return MTD_EINVAL;
    if ((g_force_cfg == -1 || g_force_cfg == 9) && gen_shape) {
        // generator-shaped layers: persistent kernel, two 32-pixel tiles per wave at M = 131072
        const int ntiles = (p.M + 31) / 32;
        int wgs = (ntiles + 7) / 8;
        if (wgs > 512) wgs = 512;
        const int prof = mtd_prof_begin(0, 9, 1, p.M, a->N, a->C, 9, s, alg_bytes);
        MTD_LAUNCH(igemm_c32p_kernel, dim3(wgs, a->N / 32), dim3(256), 0, s, p, ntiles);
        mtd_prof_end(prof, s);
        MTD_LAUNCH_CHECK();
        return MTD_OK;
    }
    dim3 grid((p.M + pl.BM - 1) / pl.BM, a->N / pl.BN, pl.splitk);
    const int prof = mtd_prof_begin(0, pl.cfg, pl.splitk, p.M, a->N, a->C, a->g.TH * a->g.TW, s, alg_bytes);
    switch (pl.cfg) {
        case 0: MTD_LAUNCH((igemm_kernel<2, 1, 4, 1>), grid, dim3(256), 0, s, p); break;
        case 1: MTD_LAUNCH((igemm_kernel<1, 1, 4, 1>), grid, dim3(256), 0, s, p); break;
        case 2: MTD_LAUNCH((igemm_kernel<2, 2, 4, 1>), grid, dim3(256), 0, s, p); break;
        case 3: MTD_LAUNCH((igemm_kernel<1, 1, 2, 2>), grid, dim3(256), 0, s, p); break;
        case 4: MTD_LAUNCH((igemm_kernel<2, 2, 2, 2>), grid, dim3(256), 0, s, p); break;
        case 6: MTD_LAUNCH((igemm_tb_kernel<1>), grid, dim3(256), 0, s, p); break;
        case 7: MTD_LAUNCH((igemm_tb_kernel<2>), grid, dim3(256), 0, s, p); break;
        case 8: MTD_LAUNCH((igemm_v2_kernel<0>), grid, dim3(256), 0, s, p); break;
        default: MTD_LAUNCH((igemm_kernel<1, 1, 1, 4>), grid, dim3(256), 0, s, p); break;
    }
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    if (pl.splitk > 1 && !p.fin) {
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

// Tail of a Res-FFT-Conv block in one launch (arch/Ours/networks.py:32-36):
//     out2 = act(conv3x3(in) + bias)          (the spatial branch; the backward pass needs it as its ReLU mask)
//     out  = in + out2 + irfft_rows(T)        (T: output of mtd_spec_mix_fwd*, [B][33][64][2][32])
// = mtd_conv_igemm(out = out2) followed by mtd_irfft_rows(T, out, add1 = in, add2 = out2), without the second launch and its
// re-reads of `in` and `out2`.  Halo-tile kernel only: C = N = 32, 3x3 stride 1 "same", 64-pixel rows, >= 32768 pixels, no
// add / mask operands, a->out2 may be NULL.  MTD_EINVAL otherwise (the caller then issues the two launches).
extern "C" int mtd_resfft_block_tail_ok(const mtd_conv_args* a) {
    if (!a || check_args(*a) != MTD_OK) return 0;
    if (a->N != 32 || a->C != 32 || a->g.TH * a->g.TW != 9 || geom_pixels(a->g) < 32768 || !c32t_eligible(*a)) return 0;
    if (a->add1 || a->add2 || a->mask || a->scale2 || !wide_epilogue_ok(*a)) return 0;
    if (a->g.OH != 64) return 0;                     // T is the spectrum of 64 x 64 patches
    return 1;
}

extern "C" int mtd_resfft_block_tail(const mtd_conv_args* a, const float* T, void* stream) {
    if (!a || !T || !mtd_resfft_block_tail_ok(a)) return MTD_EINVAL;
    if (!aligned16(T)) return MTD_EALIGN;
    Plan pl = make_plan(*a);
    IgemmParams p;
    int rc = fill_params(a, pl, p);
    if (rc != MTD_OK) return rc;
    if ((long long)a->g.B * NKW * 4096 * 4 >= (1ll << 31)) return MTD_EINVAL;      // 32-bit offsets into T
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = p.M / (C32T_R * C32T_W);
    const int prof = mtd_prof_begin(0, 12, 1, p.M, a->N, a->C, 9, s, algorithmic_bytes(a) + 4.0 * a->g.B * NKW * 4096);
#ifdef MTD_LAB       // lab builds only (stage switches that produce WRONG results); never read from the environment by the shipped library
    static const int env_lab = [] { const char* e = mtd_lab_env("MTD_TAIL_LAB"); return e ? atoi(e) : 0; }();
#else
    const int env_lab = 0;
#endif
    MTD_LAUNCH((igemm_c32t_kernel<C32T_R, true, true, true>), dim3(ntiles < 256 ? ntiles : 256, 1), dim3(512), 0, s, p, ntiles, env_lab, T);
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// Up to four launches of one shape as ONE grid (igemm_multi_kernel).  a[0..count): identical M, N, C, taps, weight view
// strides and epilogue operand KINDS (each set brings its own pointers / geometry offsets); every set with split-K needs
// its own workspace of mtd_conv_igemm_multi_ws_bytes(a, count) bytes.  Falls back to `count` single launches when the
// plan picks a kernel without a multi form.
// does a multi call run as `count` single launches (count 1, a plan without a multi form, a forced configuration, the
// generator-shaped kernels)?  One predicate for the workspace size and the launcher.
static bool multi_falls_back(const mtd_conv_args* a, int count, const Plan& pl) {
    const bool gen_shape = a[0].C == 32 && a[0].g.TH * a[0].g.TW == 9 && geom_pixels(a[0].g) >= 32768;
    return count == 1 || pl.cfg > 5 || g_force_cfg >= 6 || gen_shape;
}

extern "C" size_t mtd_conv_igemm_multi_ws_bytes(const mtd_conv_args* a, int count) {
    if (!a || count < 1 || count > MULTI_MAX || check_args(a[0]) != MTD_OK) return 0;
    Plan pl = make_plan(a[0], count);
    if (multi_falls_back(a, count, pl)) return mtd_conv_igemm_ws_bytes(&a[0]);      // the single launches' own plan
    if (pl.splitk <= 1) return 0;
    return (size_t)pl.splitk * (size_t)geom_pixels(a[0].g) * a[0].N * sizeof(float);
}

extern "C" int mtd_conv_igemm_multi(const mtd_conv_args* a, int count, void* stream) {
    for (int i = 0; a && i < count; ++i)
        if (a[i].act == MTD_ACT_RELU_ADD) return MTD_EINVAL;
    if (!a || count < 1 || count > MULTI_MAX) return MTD_EINVAL;
    for (int i = 0; i < count; ++i) {
        int rc = check_args(a[i]);
        if (rc != MTD_OK) return rc;
        if (a[i].out2) return MTD_EINVAL;
        if (geom_pixels(a[i].g) != geom_pixels(a[0].g) || a[i].N != a[0].N || a[i].C != a[0].C || a[i].g.TH != a[0].g.TH ||
            a[i].g.TW != a[0].g.TW) return MTD_EINVAL;
        // one kernel body serves every set: the weight-view strides and the KINDS of the epilogue operands must agree (each set
        // brings its own pointers), or a set would read through another set's null operand
        if (a[i].w_sn != a[0].w_sn || a[i].w_sc != a[0].w_sc || a[i].w_st != a[0].w_st || a[i].act != a[0].act ||
            a[i].mask_slope != a[0].mask_slope || a[i].scale_split != a[0].scale_split) return MTD_EINVAL;
        if ((a[i].bias == nullptr) != (a[0].bias == nullptr) || (a[i].add1 == nullptr) != (a[0].add1 == nullptr) ||
            (a[i].add2 == nullptr) != (a[0].add2 == nullptr) || (a[i].mask == nullptr) != (a[0].mask == nullptr) ||
            (a[i].scale == nullptr) != (a[0].scale == nullptr) || (a[i].scale2 == nullptr) != (a[0].scale2 == nullptr)) return MTD_EINVAL;
    }
    Plan pl = make_plan(a[0], count);
    if (multi_falls_back(a, count, pl)) {
        for (int i = 0; i < count; ++i) {
            int rc = mtd_conv_igemm(&a[i], stream);
            if (rc != MTD_OK) return rc;
        }
        return MTD_OK;
    }
    IgemmMulti mp;
    for (int i = 0; i < MULTI_MAX; ++i) {
        int rc = fill_params(&a[i < count ? i : 0], pl, mp.p[i]);
        if (rc != MTD_OK) return rc;
    }
    // (split-K sets finish through splitk_epilogue_multi_kernel: with the in-kernel finish inlined four times the compiler
    // merges the tails and selects the argument set dynamically -- 2.4 KB of scratch per lane)
    for (int i = 0; i < MULTI_MAX; ++i) mp.p[i].fin = 0;
    hipStream_t s = (hipStream_t)stream;
    const int M = mp.p[0].M;
    double bytes = 0.0;
    for (int i = 0; i < count; ++i) bytes += algorithmic_bytes(&a[i]);
    dim3 grid((M + pl.BM - 1) / pl.BM, a[0].N / pl.BN, pl.splitk * count);
    const int prof = mtd_prof_begin(0, 16 + pl.cfg, pl.splitk, (long long)M * count, a[0].N, a[0].C, a[0].g.TH * a[0].g.TW, s, bytes);   // (16 + cfg: igemm_multi_kernel<cfg>)
    switch (pl.cfg) {
        case 0: MTD_LAUNCH((igemm_multi_kernel<2, 1, 4, 1>), grid, dim3(256), 0, s, mp); break;
        case 1: MTD_LAUNCH((igemm_multi_kernel<1, 1, 4, 1>), grid, dim3(256), 0, s, mp); break;
        case 2: MTD_LAUNCH((igemm_multi_kernel<2, 2, 4, 1>), grid, dim3(256), 0, s, mp); break;
        case 3: MTD_LAUNCH((igemm_multi_kernel<1, 1, 2, 2>), grid, dim3(256), 0, s, mp); break;
        case 4: MTD_LAUNCH((igemm_multi_kernel<2, 2, 2, 2>), grid, dim3(256), 0, s, mp); break;
        default: MTD_LAUNCH((igemm_multi_kernel<1, 1, 1, 4>), grid, dim3(256), 0, s, mp); break;
    }
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    if (pl.splitk > 1 && !mp.p[0].fin) {
        bool vec = true;
        for (int i = 0; i < count; ++i) vec = vec && splitk_vec_ok(a[i], M);
        if (vec) {
            const long long total = (long long)M * a[0].N;
            int blocks = (int)((total / 4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_epilogue_multi_kernel, dim3(blocks, count), dim3(256), 0, s, mp);
            MTD_LAUNCH_CHECK();
        } else {
            const long long total = (long long)M * a[0].N;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            for (int i = 0; i < count; ++i) {
                hipLaunchKernelGGL(splitk_epilogue_scalar_kernel, dim3(blocks), dim3(256), 0, s, mp.p[i]);
                MTD_LAUNCH_CHECK();
            }
        }
    }
    return MTD_OK;
}
#endif  // MTD_NO_API
