// Backward of one 32 -> 32 channel 3x3 layer of the generator (64-pixel rows) in ONE launch: the data gradient
// (halo-tile implicit GEMM, conv_igemm.hip igemm_c32t_kernel) and the weight + bias gradient (row-window kernel,
// conv_wgrad.hip wgrad_row_body) of arch/Ours/networks.py:21-36 / :95-164's Conv2d / ConvTranspose2d(32, 32, 3, 1, 1).
//
// Why one launch.  At 32 patches a layer is 2.4 GFLOP per kernel = 15.4 us of MFMA time, and each of the two launches
// spends as long again outside its MFMA loop (prologue DMA, the store burst of the last tile, ramp and drain of a 256
// workgroup grid): 31.3 + 29.7 us per layer, 41 layers per backward pass.  The two kernels cannot share a CU as two
// launches (2 x 172 + 227 registers per SIMD lane, 136 KB + 64 KB of LDS), so streams do not overlap them.  Here a
// 512-thread workgroup per CU splits its eight waves by ROLE: waves 0-3 walk the workgroup's halo tiles exactly like the
// halo-tile kernel (two 32-pixel blocks per wave and tile instead of one), waves 4-7 take the workgroup's share of the
// pixels through the row-window weight-gradient loop (operands straight from global memory, all nine taps of the 32 x 32 block in 144
// accumulator registers across the workgroup's whole pixel range).  A SIMD hosts one wave of each role: while one waits
// for its fragments, its epilogue operands or its stores, the other has the matrix pipe -- the pipe sees 2 x 288 MFMAs
// per tile back to back, and prologue, tail and launch cost are paid once per layer.
// The main loops have NO workgroup barrier (private halos, counted vmcnt waits per wave); the roles meet at barrier 0 (the
// weights are in LDS) and at the barriers of the weight-gradient waves' final cross-wave sum, which the data-gradient waves join.
//
// Results: the data gradient's arithmetic (tap order, accumulation order, epilogue) is the halo-tile kernel's; the weight
// gradient waves take the pixel runs the stand-alone row-window launch would give them, so at 256 workgroups (32 patches)
// the slabs are that launch's bit for bit; at other sizes the split differs and with it the rounding, not the value.
// Roofline: fp32 MFMA, 2 x 2 * M * 32 * 32 * 9 flop per launch.
#define MTD_NO_API 1
#include "conv_igemm.hip"
#include "conv_wgrad.hip"

#ifndef C32F_SPREAD
#define C32F_SPREAD 1      // weight-gradient role: a row's requests one per MFMA (0: in a burst before the row's MFMAs)
#endif
namespace {

struct C32BwdParams {
    IgemmParams d;       // the data gradient as mtd_conv_igemm would run it (halo-tile eligible)
    WgradParams w;       // the weight gradient as mtd_conv_wgrad_slabs would run it (row window, one (n, c) tile)
    int ntiles, iters;   // halo tiles of the launch; tiles per workgroup (same for every workgroup: barrier counts)
    int roles;           // lab switch MTD_C32F_ROLES: bit 0 = data-gradient waves compute, bit 1 = weight-gradient waves compute
    unsigned long long* stamps;      // lab: per workgroup 16 clock stamps (tools/c32f_probe.py), or null
    const float* specT;              // SPEC: row spectrum [B][33][64][2][32] whose inverse row transform joins the data gradient
};

#define C32F_STAMP(i)                                                                              \
    do {                                                                                           \
        if (fp.stamps && lane == 0 && blockIdx.x < 16) {                                           \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            fp.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();                       \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    } while (0)

// SPEC (mtd_conv_c32_bwd_irfft): the layer is the 3x3 conv of a Res-FFT-Conv block and the launch closes the block's backward
// pass -- data gradient + irfft_rows(specT), the rfft2-backward row transform that was mtd_irfft_rows, a launch of its own
// re-reading the data gradient -- as 33 more MFMAs per 32-pixel block against the inverse-DFT matrix in LDS, exactly as in
// the forward tail (conv_igemm.hip, igemm_c32t_kernel SPEC).  The transform's result enters the epilogue as the second add.
template <int DX, bool SPEC = false>
__global__ __launch_bounds__(512, 1) void c32_bwd_kernel(const C32BwdParams fp) {
    // Private halo per data-gradient wave: the 3 x 34 pixels around its 32-pixel block (13 DMA instructions of 8 pixels),
    // double-buffered.  Nothing in LDS is shared between waves except the weights, so the main loop has NO workgroup
    // barrier: a wave waits only for its own DMA (counted vmcnt), and the two roles run free of each other -- with one
    // barrier per shared four-row tile the weight-gradient waves stood at it for half of every tile (in-kernel stamps,
    // tools/c32f_probe.py: 53 k clocks per tile for the data-gradient role, 27 k for the other).
    constexpr int T = 9, ND = 4, PHW = C32T_W / 2 + 2, PHP = 3 * PHW, PNI = (PHP + 7) / 8;      // ND: data-gradient waves
    // (the two buffers are separate LDS OBJECTS and the block loop is unrolled by two: with one array indexed by the block's
    // parity the compiler cannot tell a pending LDS-DMA into the other buffer from one into the buffer it is about to read,
    // and put a vmcnt(0) -- a wait for the NEXT block's halo -- in front of every block's first LDS read)
    __shared__ __attribute__((aligned(1024))) float Hs0[ND][PNI * 256];
    __shared__ __attribute__((aligned(1024))) float Hs1[ND][PNI * 256];
    __shared__ __attribute__((aligned(1024))) float Bs[T * 32 * 32];
    constexpr int DLD = 67;
    __shared__ float Ds[SPEC ? 64 * DLD : 1];                // inverse row-DFT matrix (igemm_c32t_kernel SPEC)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int iters = fp.iters;
    if (SPEC) {
        for (int e = tid; e < 64 * 66; e += 512) {           // (both roles; barrier 0 follows)
            const int px = e / 66, kap = e - px * 66, kw = kap >> 1;
            const int ang = (kw * px) & 63;
            const float tv = (kap & 1) ? SIN64[ang & 31] : COS64[ang & 31];
            const float wgt = (kw == 0 || kw == 32) ? 0.125f : 0.25f;
            Ds[px * DLD + kap] = (((ang & 32) != 0) != ((kap & 1) != 0)) ? -wgt * tv : wgt * tv;
        }
    }
    const int nblk = 2 * iters;                              // 32-pixel blocks per wave, either role
    (void)fp.ntiles;

    if (wave < ND) {
        // ================================================================ data gradient (the halo-tile kernel's arithmetic)
        const IgemmParams& p = fp.d;
        const mtd_conv_args& a = p.a;
        const mtd_geom& g = a.g;
        typedef __attribute__((address_space(3))) float lds_f;
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(SPEC ? fp.specT : a.in), (short)0,
                                                                             SPEC ? g.B * NKW * 16384 : 0, 0x00020000);
        const int rsub = lane >> 3, piece = (lane & 7) ^ rsub;
        const bool on = (fp.roles & 1) != 0;
        // block k of this wave: 32 pixels from ((blockIdx.x * nblk + k) * ND + wave) * 32 -- half an image row
        auto block_m = [&](int k) { return ((blockIdx.x * nblk + k) * ND + wave) * 32; };
        auto stage = [&](int k, float* Hd) {
            const int m = block_m(k);
            const int ox0 = m % C32T_W;
            const int t2 = m / C32T_W;
            const int oy = t2 % g.OH, b = t2 / g.OH;
            const bool live = on && m < p.M;
#pragma unroll
            for (int i = 0; i < PNI; ++i) {
                const int hp = 8 * i + rsub;
                const int hr = hp / PHW, hc = hp - hr * PHW;
                const int iy = oy - 1 + hr, ix = ox0 - 1 + hc;
                const bool ok = live & (hp < PHP) & ((unsigned)iy < (unsigned)g.IH) & ((unsigned)ix < (unsigned)g.IW);
                const unsigned voff = ok ? (unsigned)(((((long long)b * g.IH + iy) * g.IW + ix) * a.in_ld + piece * 4) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_f*)(Hd + i * 256), 16, voff, 0, 0, 0);
            }
        };
        if (wave == 0) C32F_STAMP(0);
        stage(0, Hs0[wave]);
        {
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), (short)0, (int)p.w_bytes, 0x00020000);
            for (int i = wave; i < T * 4; i += ND) {
                const int t = i >> 2, nn = 8 * (i & 3) + rsub;
                const unsigned voff = (unsigned)(((long long)nn * a.w_sn + (long long)p.tap_kidx[t] * a.w_st + piece * 4) * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_f*)&Bs[i * 256], 16, voff, 0, 0, 0);
            }
        }
        const ScalePair sp = load_scale(a);
        f32x4 bias4[4];
        epiw_bias(a, 4 * kh, bias4);
        const int bsw = l31 & 7;
        const int nstores = a.out2 ? 8 : 4;                  // store instructions of one block's epilogue (16-byte vectors)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                                   // barrier 0: the weights are in LDS
        if (wave == 0) C32F_STAMP(1);
        const int hp0 = PHW + l31 + 1;                        // the lane's own pixel in the private halo
        auto one_block = [&](int k, const float* H, float* Hnext) {
            if (k + 1 < nblk) stage(k + 1, Hnext);            // (that buffer was last read by block k - 1 of this same wave)
            const int mbase = block_m(k);
            const bool live = on && mbase < p.M;
            if (live) {
                EpiWide wad;
                EpiWideOps weo;
                wad.init(p, mbase, lane, 0, sp);
                epiw_load(p, wad, weo);
                float tb[NKW];                                // SPEC: the block's spectrum row, tb[kw] = T[b][kw][oy][kh][c = l31]
                if (SPEC) {
                    const int t2 = mbase / C32T_W;
                    const int oy = t2 % g.OH, b = t2 / g.OH;
                    const unsigned base = (unsigned)((((b * NKW) * 64 + oy) * 64 + kh * 32 + l31) * 4);
#pragma unroll
                    for (int kw = 0; kw < NKW; ++kw)
                        tb[kw] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(trs, base, kw * 16384, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
                auto frag = [&](int t, f32x4* af, f32x4* bf) {
                    int hp = hp0 + (g.off_y + p.tap_dy[t]) * PHW + (g.off_x + p.tap_dx[t]);
                    if (SPEC) asm volatile("" : "+v"(hp));       // piece addresses re-derived per tap (register budget: conv_igemm.hip)
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
                    for (int kk = 0; kk < 16; ++kk) acc = mfma32(bf[t & 1][kk >> 2][kk & 3], af[t & 1][kk >> 2][kk & 3], acc);   // transposed block
                }
                if (SPEC) {
                    f32x16 acc2;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
                    const float* drow = &Ds[((mbase % C32T_W) + l31) * DLD + kh];
#pragma unroll
                    for (int kw = 0; kw < NKW; ++kw) acc2 = mfma32(tb[kw], drow[2 * kw], acc2);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                        for (int j = 0; j < 4; ++j) weo.e2[g4][j] += acc2[4 * g4 + j];      // (absent add2 = -0.0f: the sum is acc2 exactly)
                }
                epiw_store(p, acc, wad, bias4, weo);             // 16-byte vectors; the stores drain under the next block's MFMAs
            }
            // the next block's halo has landed: every memory operation older than this block's stores is complete
            // (s_waitcnt vmcnt(N): all but the N youngest vector-memory operations, loads, stores and LDS-DMA alike, in issue order)
            // (the counts assume that epiw_store issues exactly `nstores` vector-memory instructions after the DMAs of stage():
            //  roles bit 3, MTD_C32F_SAFE_WAIT=1, waits for everything instead -- tests compare the two bit for bit)
            if (!live || (fp.roles & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (nstores == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (wave == 0 && k < 4) C32F_STAMP(2 + k);
        };
#pragma unroll 1
        for (int k = 0; k < nblk; k += 2) {                  // nblk = 2 * iters is even
            one_block(k, Hs0[wave], Hs1[wave]);
            one_block(k + 1, Hs1[wave], Hs0[wave]);
        }
        // the weight-gradient waves' cross-wave sum: 2 barriers per tap, 2 for the bias row
        const bool do_bias = fp.w.a.db != nullptr;
        for (int t = 0; t < T; ++t) { __syncthreads(); __syncthreads(); }
        if (do_bias) { __syncthreads(); __syncthreads(); }
        if (wave == 0) C32F_STAMP(6);
        return;
    }

    // ==================================================================== weight gradient: row window (wgrad_row_body)
    {
        constexpr int TH = 3, TW = 3, WIN = 16 + TW - 1, NWG = 4;
        const WgradParams& p = fp.w;
        const mtd_wgrad_args& a = p.a;
        const mtd_geom& g = a.g;
        const int ntiles = fp.ntiles;
        (void)ntiles;
        const int wv = wave - ND;                                  // 0 .. 3
        const bool do_bias = a.db != nullptr;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.p), (short)0, (int)p.p_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.q), (short)0, (int)p.q_bytes, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;
        const int smin = (DX > 0) ? 0 : -(TW - 1);
        const int pstep = a.p_ld * 4, qstep = a.q_ld * 4;
        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        float bsum = 0.f;
        float af[16], an[16], w0[WIN], w1[WIN];
        unsigned pbase = OOB, rowoff[TH];
        int xbase = 0;
        // chunk q of this wave: 32 pixels from (4 blockIdx.x + wv) * 64 iters + 32 q -- the run of consecutive pixels the
        // stand-alone row-window launch gives wave wv of workgroup blockIdx.x (at 256 workgroups the same pixels in the same
        // order, hence the same slabs bit for bit); past the wave's run: all loads out of range
        const int mwave0 = (blockIdx.x * NWG + wv) * (64 * iters);
        auto setup = [&](int q) {
            const int m = mwave0 + q * 32 + kh * 16;
            pbase = OOB;
            xbase = 0;
#pragma unroll
            for (int ty = 0; ty < TH; ++ty) rowoff[ty] = OOB;
            if (q < 2 * iters && m < p.M) {
                int b, oy, ox;
                pix_decompose(m, g.OW, g.OH, b, oy, ox);
                pbase = (unsigned)(((long long)m * a.p_ld + l31) * 4);
                xbase = ox + g.off_x + smin;
#pragma unroll
                for (int ty = 0; ty < TH; ++ty) {
                    const int iy = oy + g.off_y + ty * g.tap_dy;
                    if ((unsigned)iy < (unsigned)g.IH)
                        rowoff[ty] = (unsigned)(((((long long)b * g.IH + iy) * g.IW + xbase) * a.q_ld + l31) * 4);
                }
            }
        };
        auto load_p = [&](float* dst) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                dst[kk] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, pbase, kk * pstep, 0));
        };
        auto load_row = [&](int ty, float* dst) {
#pragma unroll
            for (int j = 0; j < WIN; ++j) {
                const unsigned off = (((unsigned)(xbase + j) < (unsigned)g.IW) & (rowoff[ty] != OOB)) ? rowoff[ty] + (unsigned)(j * qstep) : OOB;
                dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, off, 0, 0));
            }
        };
        auto mfma_row = [&](int ty, const float* wvv) {
#pragma unroll
            for (int tx = 0; tx < TW; ++tx) {
                const int sh = (DX > 0) ? tx : (TW - 1 - tx);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc[ty * TW + tx] = mfma32(af[kk], wvv[kk + sh], acc[ty * TW + tx]);
            }
        };
        if (wv == 0) C32F_STAMP(8);
        setup(0);
        load_p(af);
        load_row(0, w0);
        __syncthreads();                                                                   // barrier 0 (the first loads are in flight)
        if (wv == 0) C32F_STAMP(9);
        for (int q = 0; q < 2 * iters; ++q) {
            if (fp.roles & 2)
#pragma unroll
            for (int ty = 0; ty < TH; ++ty) {
                float* cur = (ty & 1) ? w1 : w0;
                float* nxt = (ty & 1) ? w0 : w1;
                __builtin_amdgcn_sched_barrier(0);
                if (ty + 1 < TH) {
                    if (!(fp.roles & 4)) load_row(ty + 1, nxt);      // (lab, roles bit 2: skip two of the three window rows -- WRONG results, timing only)
                } else {
                    setup(q + 1);
                    load_p(an);
                    load_row(0, nxt);
                }
                if (!(C32F_SPREAD)) __builtin_amdgcn_sched_barrier(0);
                if (ty == 0) {
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) bsum += af[kk];
                }
                mfma_row(ty, cur);
                if constexpr (C32F_SPREAD) {
                    // (round 5) this row's WIN (last row: 16 + WIN) dword requests one per MFMA instead of a burst in front of the
                    // row's 16 TW MFMAs (profiles/r5_load_spreading.txt)
                    if (ty + 1 < TH) {
#pragma unroll
                        for (int i = 0; i < WIN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16 + WIN; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) af[kk] = an[kk];
#pragma unroll
            for (int j = 0; j < WIN; ++j) w0[j] = w1[j];                                  // TH is odd: the next chunk's first row sits in w1
            if (wv == 0 && q < 4) C32F_STAMP(10 + q);
        }
        // ---- cross-wave sum through LDS (fixed order) and the workgroup's slab: reg_kernel_epilogue's arithmetic.  The halo
        // buffers are free: every data-gradient wave is past its last block when it joins the first barrier below.
        float* Ls = &Hs0[0][0];
        constexpr int EPW = 16 / NWG;
        float* slab = a.ws + (long long)blockIdx.x * p.slab_stride;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) Ls[wv * 1024 + e * 64 + lane] = acc[t][e];
            __syncthreads();
            float v[EPW];
#pragma unroll
            for (int i = 0; i < EPW; ++i) {
                const int e = wv * EPW + i;
                v[i] = Ls[e * 64 + lane];
#pragma unroll
                for (int w = 1; w < NWG; ++w) v[i] += Ls[w * 1024 + e * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < EPW; ++i) {
                const int e = wv * EPW + i;
                slab[((long long)t * a.N + mfma32_row(e, lane)) * a.C + l31] = v[i];
            }
        }
        if (do_bias) {
            float* red = Ls + NWG * 1024;
            __syncthreads();
            red[wv * 64 + lane] = bsum;
            __syncthreads();
            if (wv == 0 && lane < 32) {
                float s2 = 0.f;
                for (int w = 0; w < NWG; ++w) s2 += red[w * 64 + lane] + red[w * 64 + lane + 32];
                slab[(long long)p.T * a.N * a.C + lane] = s2;
            }
        }
        if (wv == 0) C32F_STAMP(14);
    }
}

unsigned long long* g_c32f_stamps = nullptr;     // lab: MTD_C32F_STAMPS=1

}  // namespace

// Eligibility: d is a launch the halo-tile kernel takes (32 input and output channels, 3x3, stride 1, 64-pixel rows, whole
// four-row tiles), w a 32 x 32 channel 3x3 weight gradient over the same pixel grid with the row-window geometry.
extern "C" int mtd_conv_c32_bwd_ok(const mtd_conv_args* d, const mtd_wgrad_args* w) {
    if (!d || !w || w->half_scale || check_args(*d) != MTD_OK || check_wargs(*w) != MTD_OK) return 0;
    if (d->N != 32 || d->C != 32 || w->N != 32 || w->C != 32 || !c32t_eligible(*d) || !wide_epilogue_ok(*d)) return 0;
    if (geom_pixels(d->g) != geom_pixels(w->g) || geom_pixels(d->g) % (C32T_R * C32T_W)) return 0;
    const mtd_geom& g = w->g;
    if (g.TH != 3 || g.TW != 3 || !row_window_ok(*w) || g.OW != C32T_W || g.IW != C32T_W || g.IH != g.OH || (g.OH % C32T_R)) return 0;
    if (g.B != d->g.B || g.OH != d->g.OH) return 0;
    return 1;
}

extern "C" size_t mtd_conv_c32_bwd_ws_bytes(const mtd_conv_args* d, const mtd_wgrad_args* w) {
    if (!mtd_conv_c32_bwd_ok(d, w)) return 0;
    const int ntiles = (int)(geom_pixels(d->g) / (C32T_R * C32T_W));
    const int grid = ntiles < 256 ? ntiles : 256;
    return (size_t)grid * (size_t)(9 * 32 * 32 + 32) * sizeof(float);
}

namespace {
int c32_bwd_launch(const mtd_conv_args* d, const mtd_wgrad_args* w, const float* specT, int* nslab, long long* slab_stride, void* stream);
}

// d: as for mtd_conv_igemm; w: as for mtd_conv_wgrad_slabs (slabs into w->ws, *nslab slabs of *slab_stride floats, to be
// summed by mtd_conv_wgrad_reduce_multi / the per-layer reduce).
extern "C" int mtd_conv_c32_bwd(const mtd_conv_args* d, const mtd_wgrad_args* w, int* nslab, long long* slab_stride, void* stream) {
    return c32_bwd_launch(d, w, nullptr, nslab, slab_stride, stream);
}

// The same launch closing the backward pass of a Res-FFT-Conv block (arch/Ours/networks.py:21-36 backward):
//     d->out = mask'( dgrad + add1 + add2 + irfft_rows(gT) )        gT = output of mtd_spec_mix_bwd / _bwd4
// = mtd_conv_c32_bwd followed by mtd_irfft_rows(gT, out, add1 = its result, mask), in one launch.  64 x 64 maps, d->add2 may
// be used as before (the transform joins it).
extern "C" int mtd_conv_c32_bwd_irfft(const mtd_conv_args* d, const mtd_wgrad_args* w, const float* gT, int* nslab, long long* slab_stride,
                                      void* stream) {
    if (!gT || !d || d->g.OH != 64) return MTD_EINVAL;
    if (!aligned16(gT)) return MTD_EALIGN;
    if ((long long)d->g.B * NKW * 16384 >= (1ll << 31)) return MTD_EINVAL;
    return c32_bwd_launch(d, w, gT, nslab, slab_stride, stream);
}

namespace {
int c32_bwd_launch(const mtd_conv_args* d, const mtd_wgrad_args* w, const float* specT, int* nslab, long long* slab_stride, void* stream) {
    if (!nslab || !slab_stride || !mtd_conv_c32_bwd_ok(d, w)) return MTD_EINVAL;
    C32BwdParams fp;
    Plan pl{};
    pl.cfg = 10; pl.splitk = 1; pl.c_per_split = 32;
    int rc = fill_params(d, pl, fp.d);
    if (rc != MTD_OK) return rc;
    WgradParams& p = fp.w;
    p.a = *w;
    p.M = (int)geom_pixels(w->g);
    p.T = 9;
    p.ppw = 0;
    p.nCt = 1;
    p.slab_stride = (long long)p.T * w->N * w->C + w->N;
    {
        const mtd_geom& gg = w->g;
        for (int t = 0; t < 9; ++t) {
            const int ty = t / gg.TW, tx = t % gg.TW;
            p.tap_dy[t] = ty * gg.tap_dy;
            p.tap_dx[t] = tx * gg.tap_dx;
            p.tap_delta[t] = (int)((((long long)(ty * gg.tap_dy) * gg.IW + tx * gg.tap_dx) * w->q_ld) * 4);
        }
        const long long pb = (((long long)p.M - 1) * w->p_ld + w->N) * 4;
        const long long qb = (((long long)gg.B * gg.IH * gg.IW - 1) * w->q_ld + w->C) * 4;
        if (pb >= (1ll << 31) || qb >= (1ll << 31)) return MTD_EINVAL;
        p.p_bytes = (unsigned)pb;
        p.q_bytes = (unsigned)qb;
    }
    fp.ntiles = p.M / (C32T_R * C32T_W);
    const int grid = fp.ntiles < 256 ? fp.ntiles : 256;
    fp.iters = (fp.ntiles + grid - 1) / grid;
    p.nslab = grid;
    static const int env_roles = [] { const char* e = mtd_lab_env("MTD_C32F_ROLES"); return e ? atoi(e) : 3; }();
    fp.roles = env_roles | (mtd_option(MTD_OPT_C32F_SAFE_WAIT) ? 8 : 0);      // (mtd_set_option("c32f_safe_wait", 1): the test flips it)
    static const bool want_stamps = mtd_lab_env("MTD_C32F_STAMPS") != nullptr;
    if (want_stamps && !g_c32f_stamps && hipMalloc(&g_c32f_stamps, 256 * sizeof(unsigned long long)) != hipSuccess) g_c32f_stamps = nullptr;
    fp.stamps = g_c32f_stamps;
    fp.specT = specT;
    if (!w->ws || w->ws_bytes < (size_t)grid * (size_t)p.slab_stride * sizeof(float)) return MTD_EWS;
    hipStream_t s = (hipStream_t)stream;
    const int prof = mtd_prof_begin(0, specT ? 13 : 11, 1, 2ll * p.M, 32, 32, 9, s,
                                    algorithmic_bytes(d) + 4.0 * ((double)p.M * 32 + (double)p.M * 32 + 9.0 * 32 * 32) +
                                        (specT ? 4.0 * d->g.B * NKW * 4096 : 0.0));
    if (specT) {
        if (w->g.tap_dx > 0) MTD_LAUNCH((c32_bwd_kernel<1, true>), dim3(grid), dim3(512), 0, s, fp);
        else MTD_LAUNCH((c32_bwd_kernel<-1, true>), dim3(grid), dim3(512), 0, s, fp);
    } else {
        if (w->g.tap_dx > 0) MTD_LAUNCH((c32_bwd_kernel<1>), dim3(grid), dim3(512), 0, s, fp);
        else MTD_LAUNCH((c32_bwd_kernel<-1>), dim3(grid), dim3(512), 0, s, fp);
    }
    mtd_prof_end(prof, s);
    MTD_LAUNCH_CHECK();
    *nslab = grid;
    *slab_stride = p.slab_stride;
    return MTD_OK;
}
}  // namespace

// lab (tools/c32f_probe.py): copy of the clock stamps of the last fused launch (MTD_C32F_STAMPS=1), 256 values
extern "C" int mtd_conv_c32_bwd_stamps(unsigned long long* host256) {
    if (!host256 || !g_c32f_stamps) return MTD_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return MTD_EINVAL;
    return hipMemcpy(host256, g_c32f_stamps, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? MTD_OK : MTD_EINVAL;
}
