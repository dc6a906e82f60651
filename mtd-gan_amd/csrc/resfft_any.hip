// Res-FFT-Conv spectral path for square maps of any power-of-two size 64 .. 512 (inference on whole slices:
// reference engine.py:89,129 runs the generator on 512 x 512 images, where rfft2 is a 512-point transform).
// The 64 x 64 training path keeps its register-resident kernels (resfft.hip); here a workgroup of 1024 threads (512 at S = 512: 256 registers per lane for the prefetched line) owns one image
// line at a time (a pair of rows, or a frequency column) for all 32 channels and transforms it in LDS: [S points][32 channels]
// re + im = S * 256 bytes (128 KB at S = 512, one workgroup per CU), radix-4 passes, lane = channel so every LDS access of the
// transform is stride-1 across lanes.  Forward transforms are decimation-in-frequency (natural in, bit-reversed out), inverse
// ones decimation-in-time (bit-reversed in, natural out): the 1x1 spectral conv between them is per frequency, so the column
// kernel never reorders anything.  Spectra: [B][kw 0..S/2][h 0..S-1][Re 32 | Im 32], ortho scaling 1/sqrt(S) per dimension.
// HBM-bound: a block moves 2.1 GB at S = 512, B = 8 (rows 0.54, columns 0.54, rows back 1.07 with the two residual operands).
// Round 4 (the round-1 kernels loaded, transformed and stored one line per workgroup, serially, with 4-byte accesses: 2.7 /
// 0.9 / 2.0 TB/s): the workgroups are PERSISTENT and walk their lines; the next line's global loads are issued as 16-byte
// vectors into registers before the current line's transform and land under it (one workgroup per CU has nothing else to hide
// them under), stores are 16-byte vectors, and the channel mix runs on the matrix cores (v_mfma_f32_32x32x2_f32 with the
// spectrum as the B operand: rows of the LDS column are padded to 33 floats so that lanes along the frequency index hit 32
// different banks) instead of 64 broadcast-read FMAs per lane and frequency.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int brev_n(int k, int logS) { return (int)(__brev((unsigned)k) >> (32 - logS)); }

// twiddle table exp(-i * pi * k / (S/2)), k = 0 .. S/2-1, written once per workgroup (cos | sin)
__device__ __forceinline__ void fill_twiddles(float* tw, int S) {
    const int half = S >> 1;
    const float inv = 1.f / (float)half;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        float sn, cs;
        sincospif((float)k * inv, &sn, &cs);
        tw[k] = cs;
        tw[half + k] = sn;
    }
}

// in-place FFT of re/im[S][32]; every thread of the workgroup takes part.  SIGN -1: forward, +1: inverse.
// Two radix-2 stages are fused per pass over LDS (four points in registers: half the LDS traffic and half the barriers
// of a stage-by-stage radix-2); an odd log2(S) leaves one single stage.
template <int SIGN>
__device__ __forceinline__ void twiddle(const float* tw, int hS, int idx, float& cs, float& sn) {
    cs = tw[idx];
    sn = (SIGN < 0) ? -tw[hS + idx] : tw[hS + idx];
}

// (NT = threads per workgroup as a compile-time constant: the butterfly loops then have constant trip counts and are unrolled
// by UNR, so that the LDS reads of UNR butterflies are in flight together)
// V = channels per lane: 1 (lane = channel, 4-byte LDS accesses) or 2 (lane = channel pair, 8-byte accesses: half the LDS
// instructions per butterfly -- the column kernel's transforms are bound by LDS instruction issue; LD must then be even).
template <int V> struct FftVec { typedef float type; };
template <> struct FftVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

template <int SIGN, bool DIF, int LD, int NT = 1024, int UNR = 2, int V = 1>
__device__ __forceinline__ void lds_fft(float* re, float* im, const float* tw, int S, int logS) {
    typedef typename FftVec<V>::type VT;
    static_assert(V == 1 || (V == 2 && (LD % 2) == 0), "8-byte accesses need even row strides");
    constexpr int CPL = 32 / V, CSH = V == 2 ? 4 : 5;          // lanes per point, log2 of it
#define FLD(P, I) (*reinterpret_cast<const VT*>((P) + (I)))
#define FST(P, I, X) (*reinterpret_cast<VT*>((P) + (I)) = (X))
    const int hS = S >> 1;
    int st = 0;
    // single radix-2 stage when log2(S) is odd: the first stage for DIF (half = S/2), the first for DIT (half = 1)
    // (Round 6: a radix-8 form -- three stages per pass, S = 512 in three passes instead of five -- was built, passed the whole-slice tests
    // and measured level with this one on one box, 20.76-21.01 against 20.80-21.55 ms per 8 slices (profiles/r6_whole_slice_radix8.txt): the
    // column kernel is not bound by its LDS passes.  Removed; the history has it.)
    if (logS & 1) {
        const int lh = DIF ? (logS - 1) : 0;
        const int half = 1 << lh, tshift = logS - 1 - lh;
#pragma unroll UNR
        for (int e = threadIdx.x; e < hS * CPL; e += (NT > 0 ? NT : (int)blockDim.x)) {
            const int c = (e & (CPL - 1)) * V, pidx = e >> CSH;
            const int grp = pidx >> lh, j = pidx & (half - 1);
            const int i0 = (((grp << 1) << lh) + j) * LD + c, i1 = i0 + half * LD;
            float cs, sn;
            twiddle<SIGN>(tw, hS, j << tshift, cs, sn);
            const VT ur = FLD(re, i0), ui = FLD(im, i0), vr = FLD(re, i1), vi = FLD(im, i1);
            if (DIF) {
                const VT dr = ur - vr, di = ui - vi;
                FST(re, i0, ur + vr); FST(im, i0, ui + vi);
                FST(re, i1, dr * cs - di * sn); FST(im, i1, dr * sn + di * cs);
            } else {
                const VT wr = vr * cs - vi * sn, wi = vr * sn + vi * cs;
                FST(re, i0, ur + wr); FST(im, i0, ui + wi);
                FST(re, i1, ur - wr); FST(im, i1, ui - wi);
            }
        }
        __syncthreads();
        st = 1;
    }
    const int nq = (S >> 2) * CPL;                    // 4-point groups per fused pass (x channel)
    for (; st < logS; st += 2) {
        if (DIF) {
            // stages with half = H and H/2;  points a, b = a + H/2, c = a + H, d = a + 3H/2 of a block of 2H
            const int lH = logS - 1 - st;             // log2(H)
            const int H = 1 << lH, Q = H >> 1;
            const int ts1 = logS - 1 - lH, ts2 = ts1 + 1;
#pragma unroll UNR
            for (int e = threadIdx.x; e < nq; e += (NT > 0 ? NT : (int)blockDim.x)) {
                const int c = (e & (CPL - 1)) * V, q = e >> CSH;
                const int blk = q >> (lH - 1), j = q & (Q - 1);
                const int ia = ((blk << (lH + 1)) + j) * LD + c, ib = ia + Q * LD, ic = ia + H * LD, id = ic + Q * LD;
                float c1, s1, c2, s2, c3, s3;
                twiddle<SIGN>(tw, hS, j << ts1, c1, s1);
                twiddle<SIGN>(tw, hS, (j + Q) << ts1, c2, s2);
                twiddle<SIGN>(tw, hS, j << ts2, c3, s3);
                const VT ar = FLD(re, ia), ai = FLD(im, ia), br = FLD(re, ib), bi = FLD(im, ib), cr = FLD(re, ic), ci = FLD(im, ic), dr = FLD(re, id), di = FLD(im, id);
                const VT a1r = ar + cr, a1i = ai + ci, tr = ar - cr, ti = ai - ci;
                const VT c1r = tr * c1 - ti * s1, c1i = tr * s1 + ti * c1;
                const VT b1r = br + dr, b1i = bi + di, ur = br - dr, ui = bi - di;
                const VT d1r = ur * c2 - ui * s2, d1i = ur * s2 + ui * c2;
                FST(re, ia, a1r + b1r); FST(im, ia, a1i + b1i);
                { const VT xr = a1r - b1r, xi = a1i - b1i; FST(re, ib, xr * c3 - xi * s3); FST(im, ib, xr * s3 + xi * c3); }
                FST(re, ic, c1r + d1r); FST(im, ic, c1i + d1i);
                { const VT xr = c1r - d1r, xi = c1i - d1i; FST(re, id, xr * c3 - xi * s3); FST(im, id, xr * s3 + xi * c3); }
            }
        } else {
            // stages with half = h and 2h;  points a, b = a + h, c = a + 2h, d = a + 3h of a block of 4h
            const int lh = st;
            const int h = 1 << lh;
            const int ts1 = logS - 1 - lh, ts2 = ts1 - 1;
#pragma unroll UNR
            for (int e = threadIdx.x; e < nq; e += (NT > 0 ? NT : (int)blockDim.x)) {
                const int c = (e & (CPL - 1)) * V, q = e >> CSH;
                const int blk = q >> lh, j = q & (h - 1);
                const int ia = ((blk << (lh + 2)) + j) * LD + c, ib = ia + h * LD, ic = ib + h * LD, id = ic + h * LD;
                float c1, s1, c2, s2, c3, s3;
                twiddle<SIGN>(tw, hS, j << ts1, c1, s1);
                twiddle<SIGN>(tw, hS, j << ts2, c2, s2);
                twiddle<SIGN>(tw, hS, (j + h) << ts2, c3, s3);
                const VT ar = FLD(re, ia), ai = FLD(im, ia), br = FLD(re, ib), bi = FLD(im, ib), cr = FLD(re, ic), ci = FLD(im, ic), dr = FLD(re, id), di = FLD(im, id);
                const VT bwr = br * c1 - bi * s1, bwi = br * s1 + bi * c1;
                const VT dwr = dr * c1 - di * s1, dwi = dr * s1 + di * c1;
                const VT a1r = ar + bwr, a1i = ai + bwi, b1r = ar - bwr, b1i = ai - bwi;
                const VT c1r = cr + dwr, c1i = ci + dwi, d1r = cr - dwr, d1i = ci - dwi;
                const VT cwr = c1r * c2 - c1i * s2, cwi = c1r * s2 + c1i * c2;
                const VT ewr = d1r * c3 - d1i * s3, ewi = d1r * s3 + d1i * c3;
                FST(re, ia, a1r + cwr); FST(im, ia, a1i + cwi);
                FST(re, ic, a1r - cwr); FST(im, ic, a1i - cwi);
                FST(re, ib, b1r + ewr); FST(im, ib, b1i + ewi);
                FST(re, id, b1r - ewr); FST(im, id, b1i - ewi);
            }
        }
        __syncthreads();
    }
#undef FLD
#undef FST
}

// Lab knobs (compile time; tools/any_variants.sh builds and times them on the GPU box).  Measured at S = 512, B = 8, ms per 21
// blocks (rows / columns / rows back): default 2.50 / 6.69 / 4.52; butterfly loops with compile-time trip counts
// (-DMTD_ANY_CT) 2.54 / 8.27 / 4.54 and unrolled by 2 (-DMTD_ANY_UNR=2) 2.54 / 8.08 / 4.52 -- the compiler then keeps more
// butterflies in flight than a lane's 128 registers hold and spills; 512 threads per workgroup (-DMTD_ANY_NT512=512: 256
// registers per lane, no spills) 2.63 / 6.82 / 4.82.
#ifndef MTD_ANY_SKIP      // lab (wrong results): bit 0 skips the forward column transform, bit 1 the mix's MFMAs, bit 2 the inverse transform
#define MTD_ANY_SKIP 0
#endif
#ifndef MTD_ANY_FFTV      // channels per lane in the column kernel's transforms (1 or 2)
#define MTD_ANY_FFTV 2
#endif
#ifndef MTD_ANY_NT512
#define MTD_ANY_NT512 1024
#endif
#ifndef MTD_ANY_UNR
#define MTD_ANY_UNR 1
#endif
constexpr int FFT_UNR = MTD_ANY_UNR;
#ifdef MTD_ANY_CT       // trip counts of the butterfly loops from the template constant instead of blockDim.x
#define FFT_NT(NT) NT
#else
#define FFT_NT(NT) 0
#endif
constexpr int NCU = 256;       // MI355X: persistent grids are sized in workgroups per CU

template <int S> struct Log2 { static constexpr int v = 1 + Log2<S / 2>::v; };
template <> struct Log2<1> { static constexpr int v = 0; };

// rows forward: a pair of image rows (b, h), (b, h + 1) per step -- the two real rows are the real and imaginary part of one
// complex transform Z; X_h[k] = (Z[k] + conj Z[-k]) / 2, X_{h+1}[k] = (Z[k] - conj Z[-k]) / 2i.  unit u = b * S/2 + h/2.
template <int S, int NT>
__global__ __launch_bounds__(NT) void rfft_rows_any_kernel(const float* __restrict__ x, int x_ld, float* __restrict__ R, int units, int rev) {
    constexpr int logS = Log2<S>::v, NV = (S * 16 + NT - 1) / NT, nkw = S / 2 + 1;
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + S * 32;
    float* tw = lds + S * 64;
    fill_twiddles(tw, S);
    const int tid = threadIdx.x;
    f32x4 v[NV];
    auto issue = [&](int u) {
        const int uu = rev ? units - 1 - u : u, b = uu / (S / 2), h = (uu % (S / 2)) * 2;
        const float* src = x + ((long long)(b * S + h) * S) * x_ld;           // rows h, h + 1: 2 S consecutive pixels
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = tid + NT * j, p = q >> 3, c4 = q & 7;
            if (NV * NT == S * 16 || q < S * 16) v[j] = *reinterpret_cast<const f32x4*>(src + (long long)p * x_ld + c4 * 4);
        }
    };
    int u = blockIdx.x;
    if (u < units) issue(u);
    for (; u < units; u += gridDim.x) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = tid + NT * j, p = q >> 3, c4 = q & 7;
            if (NV * NT == S * 16 || q < S * 16) *reinterpret_cast<f32x4*>((p >= S ? im : re) + (p & (S - 1)) * 32 + c4 * 4) = v[j];
        }
        __syncthreads();
        if (u + (int)gridDim.x < units) issue(u + gridDim.x);                  // lands under this pair's transform
        lds_fft<-1, true, 32, FFT_NT(NT), FFT_UNR>(re, im, tw, S, logS);
        const int uu = rev ? units - 1 - u : u, b = uu / (S / 2), h = (uu % (S / 2)) * 2;
        const float sc = 0.5f * rsqrtf((float)S);
        for (int it = tid; it < nkw * 8; it += NT) {
            const int kw = it >> 3, c4 = it & 7;
            const int pk = brev_n(kw, logS) * 32 + c4 * 4, pm = brev_n((S - kw) & (S - 1), logS) * 32 + c4 * 4;
            const f32x4 zkr = *reinterpret_cast<const f32x4*>(re + pk), zki = *reinterpret_cast<const f32x4*>(im + pk);
            const f32x4 zmr = *reinterpret_cast<const f32x4*>(re + pm), zmi = *reinterpret_cast<const f32x4*>(im + pm);
            float* o = R + (((long long)(b * nkw + kw) * S + h) * 64) + c4 * 4;
            *reinterpret_cast<f32x4*>(o) = (zkr + zmr) * sc;
            *reinterpret_cast<f32x4*>(o + 32) = (zki - zmi) * sc;
            *reinterpret_cast<f32x4*>(o + 64) = (zki + zmi) * sc;
            *reinterpret_cast<f32x4*>(o + 96) = (zmr - zkr) * sc;
        }
        __syncthreads();
    }
}

// columns + channel mix + columns back: a column (b, kw) per step.  LDS rows are CLD = 33 floats apart (see the header).
//
// PACK (round 6): the columns kw = 0 and kw = S/2 of an image hold REAL sequences (the row transform of real data is real there:
// rfft_rows_any_kernel writes exact zeros into their imaginary halves) and only the REAL part of their way back is ever used
// (irfft_rows_any_kernel ignores the imaginary parts of columns 0 and S/2, as torch's c2r does) -- so the two go through ONE
// complex transform each way, like the row pairs of the row kernels: z = x_0 + i x_{S/2}; after the forward transform
// A[k] = (Z[k] + conj Z[-k]) / 2 and B[k] = (Z[k] - conj Z[-k]) / 2i are the two columns' spectra, both Hermitian, kept IN PLACE (A[k] in
// row k, B[k] in row S - k, 0 < k < S/2; rows 0 and S/2 hold the real pairs (A, B) as they are); the mix of a row X gives
// Y(X) = relu(W [Re X; Im X] + b) and Y(conj X) from the same two half sums P (real parts) and Q (imaginary parts) -- P + Q and P - Q --,
// and what the way back needs is the Hermitian part (Y(X) + conj Y(conj X)) / 2 of the column, whose inverse transform is the real
// part of the column's; the two Hermitian columns go back as U + i V in one transform, real part = column 0, imaginary part =
// column S/2.  An image is then S/2 units instead of S/2 + 1: at S = 512 and 8 slices 2 048 units on 256 CUs are eight rounds,
// 2 056 were nine (the ninth on eight CUs).  A packed unit costs the same MFMAs, two more passes over the column in LDS and two
// 32 x 32 products on the vector ALU for rows 0 and S/2.  The imaginary halves of columns 0 and S/2 of T are written as zeros.
constexpr int CLD = 34;        // even (8-byte transform accesses), 2-way bank conflicts for the MFMA operand reads along the frequency index
template <int S, int NT, bool PACK>
__global__ __launch_bounds__(NT) void spec_mix_any_kernel(const float* __restrict__ R, const float* __restrict__ w2t,
                                                           const float* __restrict__ b2, float* __restrict__ T, int units, int rev) {
    constexpr int logS = Log2<S>::v, NV = (S * 16 + NT - 1) / NT, nkw = S / 2 + 1;
    static_assert(NT % 16 == 0, "a thread keeps its 16-byte part of a spectrum row over a unit");
#ifdef MTD_ANY_EARLY      /* lab: how many of the next column's NV load vectors go out before the mix / before the forward transform */
    constexpr int EARLY = MTD_ANY_EARLY < NV ? MTD_ANY_EARLY : NV;
#else
    constexpr int EARLY = NV / 2;
#endif
#ifdef MTD_ANY_EARLY_TOP
    constexpr int EARLY_TOP = MTD_ANY_EARLY_TOP < EARLY ? MTD_ANY_EARLY_TOP : EARLY;
#else
    constexpr int EARLY_TOP = 0;
#endif
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + S * CLD;
    float* tw = lds + 2 * S * CLD;
    float* wl = tw + S;                    // w2t [k 64][o 64] + bias [64]
    fill_twiddles(tw, S);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int part = tid & 15;             // which 16 bytes of a 256-byte spectrum row (Re 0..7 | Im 8..15) this thread moves
    for (int i = tid; i < 64 * 64 + 64; i += NT) wl[i] = i < 4096 ? w2t[i] : b2[i - 4096];
    // unit u -> (image, column); PACK: S/2 units per image, column 0 standing for the pair (0, S/2), and the images' pairs rotated
    // over the workgroups (with S/2 units per image and a grid of S/2 they would all be workgroup 0's)
    auto unit_col = [&](int uw, int& b) {
        const int u = rev ? units - 1 - uw : uw;      // (rev: the walk runs from the last image to the first)
        if (!PACK) { b = u / nkw; return u - b * nkw; }
        b = u / (S / 2);
        return (u - b * (S / 2) + b) & (S / 2 - 1);
    };
    // a unit's place in a spectrum buffer (uniform over the workgroup: a scalar base) and what this thread adds to its 4 q floats: a
    // packed unit's imaginary parts (part >= 8) are the REAL halves of column S/2
    constexpr int PDELTA = (S / 2) * S * 64 - 32;
    auto unit_base = [&](int u, int& poff) -> long long {
        int b;
        const int kw = unit_col(u, b);
        poff = (PACK && kw == 0 && part >= 8) ? PDELTA : 0;
        return ((long long)(b * nkw + kw) * S) * 64;
    };
    f32x4 v[NV];
    auto issue = [&](int un, int j0, int j1) {
        int poff;
        const float* src = R + unit_base(un, poff);
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const int q = tid + NT * j;
            if (NV * NT == S * 16 || q < S * 16) v[j] = *reinterpret_cast<const f32x4*>(src + (q * 4 + poff));
        }
    };
    int u = blockIdx.x;
    if (u < units) issue(u, 0, NV);
    const float sc = rsqrtf((float)S);
    for (; u < units; u += gridDim.x) {
        int b_cur;
        const bool packed = PACK && unit_col(u, b_cur) == 0;
        const int un = u + (int)gridDim.x;
        const bool more = un < units;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = tid + NT * j, h = q >> 4;
            if (NV * NT == S * 16 || q < S * 16) {
                float* d = (part >= 8 ? im : re) + h * CLD + (part & 7) * 4;
                d[0] = v[j][0]; d[1] = v[j][1]; d[2] = v[j][2]; d[3] = v[j][3];
            }
        }
        __syncthreads();
        if (EARLY_TOP > 0 && more) issue(un, 0, EARLY_TOP);
        if (!(MTD_ANY_SKIP & 1)) lds_fft<-1, true, CLD, FFT_NT(NT), FFT_UNR, MTD_ANY_FFTV>(re, im, tw, S, logS);
        // packed unit, rows k and S - k of the transform (bit-reversed places): Z[k], Z[S-k] -> A[k], B[k] (to_spectra) and, after
        // the mix, the Hermitian columns U[k] (row k), V[k] (row S - k) -> (U + i V)[k], (U + i V)[S-k]
        auto pair_pass = [&](bool to_spectra) {
            int e = tid;
            asm volatile("" : "+v"(e));      // (opaque: the pass's row addresses are not to become invariants of the unit loop, held in registers over it)
#pragma nounroll
            for (; e < (S / 2 - 1) * 16; e += NT) {
                const int p = 1 + (e >> 4), c = (e & 15) * 2;
                const int rk = brev_n(p, logS) * CLD + c, rm = brev_n(S - p, logS) * CLD + c;
                const f32x2 kr = *reinterpret_cast<const f32x2*>(re + rk), ki = *reinterpret_cast<const f32x2*>(im + rk);
                const f32x2 mr = *reinterpret_cast<const f32x2*>(re + rm), mi = *reinterpret_cast<const f32x2*>(im + rm);
                if (to_spectra) {
                    *reinterpret_cast<f32x2*>(re + rk) = 0.5f * (kr + mr);
                    *reinterpret_cast<f32x2*>(im + rk) = 0.5f * (ki - mi);
                    *reinterpret_cast<f32x2*>(re + rm) = 0.5f * (ki + mi);
                    *reinterpret_cast<f32x2*>(im + rm) = 0.5f * (mr - kr);
                } else {
                    *reinterpret_cast<f32x2*>(re + rk) = kr - mi;
                    *reinterpret_cast<f32x2*>(im + rk) = ki + mr;
                    *reinterpret_cast<f32x2*>(re + rm) = kr + mi;
                    *reinterpret_cast<f32x2*>(im + rm) = mr - ki;
                }
            }
            __syncthreads();
        };
        if (packed) pair_pass(true);
        // channel mix at every frequency on the matrix cores: D[o][n] = sum_k W[k][o] * Z[n][k], k = (re 0..31 | im 32..63).
        // A operand: lane (o = l & 31, k = l >> 5) of W from LDS; B operand: lane (n = l & 31, k = l >> 5) of the column.
        // work items = (32-row tile, output half): S / 32 * 2 over the 16 waves; results stay in registers until every
        // wave has read its operands (two waves may share a tile), then replace the column.
        // the next column's loads in two halves: one requested here, before the mix (16 registers beside its accumulators: more do
        // not fit a 1024-thread workgroup's 128), the other after it -- with all of them after the mix their way from memory had
        // only the inverse transform to hide under, and 256 CUs asking for 33 MB at once need longer than that (6.28 -> 5.99 ms)
        if (more && !packed) issue(un, EARLY_TOP, EARLY);      // (a packed unit's two accumulator sets leave no room: all after the mix)
        constexpr int NW = NT / 64, ITEMS = S / 32 * 2, PER = (ITEMS + NW - 1) / NW;
        if (!packed) {
            f32x16 acc[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int item = wv + NW * i;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
                if (item < ITEMS && !(MTD_ANY_SKIP & 2)) {
                    const int tile = item % (S / 32), ob = item / (S / 32);
                    const float* zr = re + (tile * 32 + (lane & 31)) * CLD + (lane >> 5);
                    const float* zi = im + (tile * 32 + (lane & 31)) * CLD + (lane >> 5);
                    const float* wp = wl + (lane >> 5) * 64 + ob * 32 + (lane & 31);
#pragma unroll 4
                    for (int s2 = 0; s2 < 16; ++s2) acc[i] = mfma32(wp[s2 * 128], zr[s2 * 2], acc[i]);
#pragma unroll 4
                    for (int s2 = 0; s2 < 16; ++s2) acc[i] = mfma32(wp[(16 + s2) * 128], zi[s2 * 2], acc[i]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int item = wv + NW * i;
                if (item < ITEMS) {
                    const int tile = item % (S / 32), ob = item / (S / 32);
                    // (output channel of accumulator element r: mfma32_row(r, lane) = 4 (lane >> 5) + a constant -- ONE address register each
                    // for the row and the bias, the sixteen places as immediate offsets)
                    float* dst = (ob ? im : re) + (tile * 32 + (lane & 31)) * CLD + 4 * (lane >> 5);
                    const float* bia = wl + 4096 + ob * 32 + 4 * (lane >> 5);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int oc = (r & 3) + 8 * (r >> 2);
                        dst[oc] = fmaxf(acc[i][r] * sc + bia[oc], 0.f);
                    }
                }
            }
        } else {
            // the two half sums apart: P over the real parts, Q over the imaginary parts; Y(X) = relu((P + Q) sc + b), Y(conj X) =
            // relu((P - Q) sc + b); the row becomes (Y(X) + conj Y(conj X)) / 2.  Rows 0 and S/2 (places 0 and 1) hold two REAL
            // values (A, B): their results are relu(W_rr A sc + b_r) and relu(W_rr B sc + b_r) (the real part of the mix of a real
            // vector), 2 x 2 x 32 sums of 32 products by the first 128 threads.
            // (opaque copies of the thread's indices: what this rare branch derives from them -- sixteen bias and sixteen store addresses --
            // would otherwise become invariants of the unit loop, computed ahead of it and held, or spilled, over every unit)
            int tid_p = tid;
            asm volatile("" : "+v"(tid_p));
            const int lane_p = tid_p & 63, wv_p = __builtin_amdgcn_readfirstlane(tid_p >> 6);
            f32x16 outv[PER];
            float edge = 0.f;
            if (tid_p < 128) {
                const float* src = ((tid_p & 32) ? im : re) + (tid_p >> 6) * CLD;
                const int o = tid_p & 31;
                float t = 0.f;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) t += wl[k * 64 + o] * src[k];
                edge = fmaxf(t * sc + wl[4096 + o], 0.f);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int item = wv_p + NW * i;
                f32x16 accP, accQ;
#pragma unroll
                for (int r = 0; r < 16; ++r) accP[r] = accQ[r] = 0.f;
                if (item < ITEMS && !(MTD_ANY_SKIP & 2)) {
                    const int tile = item % (S / 32), ob = item / (S / 32);
                    const float* zr = re + (tile * 32 + (lane_p & 31)) * CLD + (lane_p >> 5);
                    const float* zi = im + (tile * 32 + (lane_p & 31)) * CLD + (lane_p >> 5);
                    const float* wp = wl + (lane_p >> 5) * 64 + ob * 32 + (lane_p & 31);
#pragma unroll 4
                    for (int s2 = 0; s2 < 16; ++s2) accP = mfma32(wp[s2 * 128], zr[s2 * 2], accP);
#pragma unroll 4
                    for (int s2 = 0; s2 < 16; ++s2) accQ = mfma32(wp[(16 + s2) * 128], zi[s2 * 2], accQ);
                    const float sgn = ob ? -1.f : 1.f;
                    const float* bia = wl + 4096 + ob * 32 + 4 * (lane_p >> 5);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bb = bia[(r & 3) + 8 * (r >> 2)];
                        const float y1 = fmaxf((accP[r] + accQ[r]) * sc + bb, 0.f), y2 = fmaxf((accP[r] - accQ[r]) * sc + bb, 0.f);
                        outv[i][r] = 0.5f * (y1 + sgn * y2);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int item = wv_p + NW * i;
                if (item < ITEMS) {
                    const int tile = item % (S / 32), ob = item / (S / 32);
                    const int row = tile * 32 + (lane_p & 31);
                    float* dst = (ob ? im : re) + row * CLD + 4 * (lane_p >> 5);
                    if (row >= 2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[(r & 3) + 8 * (r >> 2)] = outv[i][r];
                    }
                }
            }
            if (tid_p < 128) (((tid_p & 32) ? im : re) + (tid_p >> 6) * CLD)[tid_p & 31] = edge;
        }
        __syncthreads();
        if (packed) pair_pass(false);
        // the next column's loads: issued here, after the mix (their 32 registers are not live under its accumulators -- with
        // 1024 threads a lane has 128), they land under the inverse transform
        if (more) {      // (constant bounds in either call: a run-time first index would send the prefetch registers to scratch memory)
            if (packed) issue(un, EARLY_TOP, NV);
            else issue(un, EARLY, NV);
        }
        if (!(MTD_ANY_SKIP & 4)) lds_fft<+1, false, CLD, FFT_NT(NT), FFT_UNR, MTD_ANY_FFTV>(re, im, tw, S, logS);
        int poff;
        float* dstg = T + unit_base(u, poff);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int q = tid + NT * j, h = q >> 4;
            if (NV * NT == S * 16 || q < S * 16) {
                const float* d = (part >= 8 ? im : re) + h * CLD + (part & 7) * 4;
                f32x4 o = {d[0] * sc, d[1] * sc, d[2] * sc, d[3] * sc};
                *reinterpret_cast<f32x4*>(dstg + (q * 4 + poff)) = o;
                if (packed) *reinterpret_cast<f32x4*>(dstg + (q * 4 + poff + 32)) = f32x4{0.f, 0.f, 0.f, 0.f};      // (the imaginary halves)
            }
        }
        __syncthreads();
    }
}

// rows back (c2r): a pair of image rows per step; out = y + add1 + add2.  With A = X_h, B = X_{h+1} (Hermitian, the imaginary
// parts of columns 0 and S/2 ignored as torch's c2r does) the complex spectrum Z = A + iB transforms back to row h in the
// real part and row h + 1 in the imaginary part.
template <int S, int NT>
__global__ __launch_bounds__(NT) void irfft_rows_any_kernel(const float* __restrict__ T, float* __restrict__ out, int out_ld,
                                                              const float* __restrict__ add1, int add1_ld,
                                                              const float* __restrict__ add2, int add2_ld, int units, int rev) {
    constexpr int logS = Log2<S>::v, nkw = S / 2 + 1, NI = (nkw * 8 + NT - 1) / NT, NV = (S * 16 + NT - 1) / NT;
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + S * 32;
    float* tw = lds + S * 64;
    fill_twiddles(tw, S);
    const int tid = threadIdx.x;
    f32x4 t0[NI], t1[NI], t2[NI], t3[NI];
    auto issue = [&](int u) {
        const int uu = rev ? units - 1 - u : u, b = uu / (S / 2), h = (uu % (S / 2)) * 2;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int it = tid + NT * j, kw = it >> 3, c4 = it & 7;
            if (it < nkw * 8) {
                const float* t = T + (((long long)(b * nkw + kw) * S + h) * 64) + c4 * 4;
                t0[j] = *reinterpret_cast<const f32x4*>(t);
                t1[j] = *reinterpret_cast<const f32x4*>(t + 32);
                t2[j] = *reinterpret_cast<const f32x4*>(t + 64);
                t3[j] = *reinterpret_cast<const f32x4*>(t + 96);
            }
        }
    };
    int u = blockIdx.x;
    if (u < units) issue(u);
    const float sc = rsqrtf((float)S);
    for (; u < units; u += gridDim.x) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int it = tid + NT * j, kw = it >> 3, c4 = it & 7;
            if (it < nkw * 8) {
                const bool edge = (kw == 0 || kw == S / 2);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const f32x4 ar = t0[j], ai = edge ? zero : t1[j], br = t2[j], bi = edge ? zero : t3[j];
                const int p0 = brev_n(kw, logS) * 32 + c4 * 4;
                *reinterpret_cast<f32x4*>(re + p0) = ar - bi;
                *reinterpret_cast<f32x4*>(im + p0) = ai + br;
                if (!edge) {
                    const int p1 = brev_n(S - kw, logS) * 32 + c4 * 4;       // Z[S-k] = conj(A[k]) + i conj(B[k])
                    *reinterpret_cast<f32x4*>(re + p1) = ar + bi;
                    *reinterpret_cast<f32x4*>(im + p1) = br - ai;
                }
            }
        }
        __syncthreads();
        if (u + (int)gridDim.x < units) issue(u + gridDim.x);
        lds_fft<+1, false, 32, FFT_NT(NT), FFT_UNR>(re, im, tw, S, logS);
        const int uu = rev ? units - 1 - u : u, b = uu / (S / 2), h = (uu % (S / 2)) * 2;
        const long long rowpix = (long long)(b * S + h) * S;                  // rows h, h + 1: 2 S consecutive pixels
        constexpr int CH = NV < 2 ? NV : 2;                                   // residual operands in chunks: 2 x CH vectors in flight
#pragma nounroll
        for (int j0 = 0; j0 < NV; j0 += CH) {
            f32x4 a1[CH], a2[CH];
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int q = tid + NT * (j0 + jj), p = q >> 3, c4 = q & 7;
                if (NV * NT == S * 16 || q < S * 16) {
                    if (add1) a1[jj] = *reinterpret_cast<const f32x4*>(add1 + (rowpix + p) * add1_ld + c4 * 4);
                    if (add2) a2[jj] = *reinterpret_cast<const f32x4*>(add2 + (rowpix + p) * add2_ld + c4 * 4);
                }
            }
#pragma unroll
            for (int jj = 0; jj < CH; ++jj) {
                const int q = tid + NT * (j0 + jj), p = q >> 3, c4 = q & 7;
                if (NV * NT == S * 16 || q < S * 16) {
                    f32x4 o = *reinterpret_cast<const f32x4*>((p >= S ? im : re) + (p & (S - 1)) * 32 + c4 * 4) * sc;
                    if (add1) o += a1[jj];
                    if (add2) o += a2[jj];
                    *reinterpret_cast<f32x4*>(out + (rowpix + p) * out_ld + c4 * 4) = o;
                }
            }
        }
        __syncthreads();
    }
}

int log2_exact(int S) {
    int l = 0;
    while ((1 << l) < S) ++l;
    return ((1 << l) == S) ? l : -1;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return MTD_OK;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? MTD_OK : (int)e;
}

// Which of the three kernels walk their lines from the LAST image to the first (bit 0 rows, 1 columns, 2 rows back).  The 256 MB
// memory-side cache holds what the previous kernel touched last: a kernel that starts where its producer stopped finds part of its
// input there.  The conv kernels before and after walk forward, so: rows backward, columns forward, rows back backward (5) --
// 20.24 -> 20.03 ms per 8 slices (six A/B pairs, profiles/r6_ab_experiments.txt item 21); the other seven settings lie between
// (lab library: MTD_ANY_REV).
inline int any_rev() {
    static const int env_rev = [] { const char* e = mtd_lab_env("MTD_ANY_REV"); return e ? atoi(e) : 5; }();
    return env_rev;
}

// persistent grid: as many workgroups as fit the chip at this LDS footprint (160 KB per CU, 2048 threads per CU)
inline int persistent_grid(int units, size_t lds_bytes) {
    int per_cu = (int)((160 * 1024) / (lds_bytes + 1024));
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    const int g = NCU * per_cu;
    return units < g ? units : g;
}

template <int S>
int launch_rfft_rows(const float* x, int x_ld, float* R, int B, hipStream_t s) {
    const size_t lds = (size_t)S * 256 + (size_t)S * 4;
    constexpr int NT = S >= 512 ? MTD_ANY_NT512 : 1024;
    int rc = set_lds(rfft_rows_any_kernel<S, NT>, lds);
    if (rc != MTD_OK) return rc;
    const int units = B * S / 2;
    const int prof = mtd_prof_begin(2, 0, 1, (long long)B * S * S, 32, 32, 0, s, 4.0 * B * S * 32.0 * (S + 2.0 * (S / 2 + 1)));
    MTD_LAUNCH((rfft_rows_any_kernel<S, NT>), dim3(persistent_grid(units, lds)), dim3(NT), lds, s, x, x_ld, R, units, any_rev() & 1);
    mtd_prof_end(prof, s);
    return MTD_OK;
}

template <int S, bool PACK>
int launch_spec_mix_form(const float* R, const float* w2t, const float* b2, float* T, int B, hipStream_t s) {
    const size_t lds = (size_t)2 * S * CLD * 4 + (size_t)S * 4 + (64 * 64 + 64) * 4;
    constexpr int NT = S >= 512 ? MTD_ANY_NT512 : 1024;
    int rc = set_lds(spec_mix_any_kernel<S, NT, PACK>, lds);
    if (rc != MTD_OK) return rc;
    const int units = B * (PACK ? S / 2 : S / 2 + 1);
    const int prof = mtd_prof_begin(2, 1, 1, (long long)B * S * (S / 2 + 1), 64, 64, 0, s, 2.0 * 4.0 * B * S * 64.0 * (S / 2 + 1));
    MTD_LAUNCH((spec_mix_any_kernel<S, NT, PACK>), dim3(persistent_grid(units, lds)), dim3(NT), lds, s, R, w2t, b2, T, units, (any_rev() >> 1) & 1);
    mtd_prof_end(prof, s);
    return MTD_OK;
}

template <int S>
int launch_spec_mix(const float* R, const float* w2t, const float* b2, float* T, int B, hipStream_t s) {
    // (lab library: MTD_ANY_PACK=0 keeps the columns 0 and S/2 as units of their own)
    static const int env_pack = [] { const char* e = mtd_lab_env("MTD_ANY_PACK"); return e ? atoi(e) : 1; }();
    return env_pack ? launch_spec_mix_form<S, true>(R, w2t, b2, T, B, s) : launch_spec_mix_form<S, false>(R, w2t, b2, T, B, s);
}

template <int S>
int launch_irfft_rows(const float* T, float* out, int out_ld, const float* add1, int add1_ld, const float* add2, int add2_ld, int B,
                      hipStream_t s) {
    const size_t lds = (size_t)S * 256 + (size_t)S * 4;
    constexpr int NT = S >= 512 ? MTD_ANY_NT512 : 1024;
    int rc = set_lds(irfft_rows_any_kernel<S, NT>, lds);
    if (rc != MTD_OK) return rc;
    const int units = B * S / 2;
    const int prof = mtd_prof_begin(2, 2, 1, (long long)B * S * S, 32, 32, 0, s,
                                    4.0 * B * S * 32.0 * (2.0 * (S / 2 + 1) + S * (1.0 + (add1 ? 1 : 0) + (add2 ? 1 : 0))));
    MTD_LAUNCH((irfft_rows_any_kernel<S, NT>), dim3(persistent_grid(units, lds)), dim3(NT), lds, s, T, out, out_ld, add1, add1_ld, add2,
               add2_ld, units, (any_rev() >> 2) & 1);
    mtd_prof_end(prof, s);
    return MTD_OK;
}

}  // namespace

// (16-byte accesses: pixel strides in multiples of 4 floats, 16-byte aligned bases -- what the generator's NHWC tensors are)
#define MTD_BY_SIDE(S, CALL64, CALL128, CALL256, CALL512) \
    ((S) == 64 ? (CALL64) : (S) == 128 ? (CALL128) : (S) == 256 ? (CALL256) : (CALL512))

extern "C" int mtd_rfft_rows_any(const float* x, int x_ld, float* R, int B, int S, void* stream) {
    const int logS = log2_exact(S);
    if (!x || !R || B <= 0 || logS < 6 || S > 512 || x_ld < 32) return MTD_EINVAL;
    if ((x_ld % 4) || !aligned16(x) || !aligned16(R)) return MTD_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    int rc = MTD_BY_SIDE(S, launch_rfft_rows<64>(x, x_ld, R, B, s), launch_rfft_rows<128>(x, x_ld, R, B, s),
                         launch_rfft_rows<256>(x, x_ld, R, B, s), launch_rfft_rows<512>(x, x_ld, R, B, s));
    if (rc != MTD_OK) return rc;
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_spec_mix_any(const float* R, const float* w2t, const float* b2, float* T, int B, int S, void* stream) {
    const int logS = log2_exact(S);
    if (!R || !w2t || !b2 || !T || B <= 0 || logS < 6 || S > 512) return MTD_EINVAL;
    if (!aligned16(R) || !aligned16(T)) return MTD_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    int rc = MTD_BY_SIDE(S, launch_spec_mix<64>(R, w2t, b2, T, B, s), launch_spec_mix<128>(R, w2t, b2, T, B, s),
                         launch_spec_mix<256>(R, w2t, b2, T, B, s), launch_spec_mix<512>(R, w2t, b2, T, B, s));
    if (rc != MTD_OK) return rc;
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_irfft_rows_any(const float* T, float* out, int out_ld, const float* add1, int add1_ld, const float* add2,
                                  int add2_ld, int B, int S, void* stream) {
    const int logS = log2_exact(S);
    if (!T || !out || B <= 0 || logS < 6 || S > 512 || out_ld < 32) return MTD_EINVAL;
    if ((add1 && add1_ld < 32) || (add2 && add2_ld < 32)) return MTD_EINVAL;
    if ((out_ld % 4) || !aligned16(T) || !aligned16(out) || (add1 && ((add1_ld % 4) || !aligned16(add1))) ||
        (add2 && ((add2_ld % 4) || !aligned16(add2))))
        return MTD_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    int rc = MTD_BY_SIDE(S, launch_irfft_rows<64>(T, out, out_ld, add1, add1_ld, add2, add2_ld, B, s),
                         launch_irfft_rows<128>(T, out, out_ld, add1, add1_ld, add2, add2_ld, B, s),
                         launch_irfft_rows<256>(T, out, out_ld, add1, add1_ld, add2, add2_ld, B, s),
                         launch_irfft_rows<512>(T, out, out_ld, add1, add1_ld, add2, add2_ld, B, s));
    if (rc != MTD_OK) return rc;
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
