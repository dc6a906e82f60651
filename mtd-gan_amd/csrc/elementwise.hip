// HBM-bound element-wise kernels of the MTD-GAN step: activation-gradient masks, channel-slice copies
// (torch.cat / chunk at networks.py:421-465), bilinear x2 up-sampling (nn.Upsample, align_corners =
// False) with its adjoint, PixelShuffle(2) with its adjoint.  All NHWC with explicit leading
// dimensions; lanes run over channels so every access is a contiguous run of the pixel's channels
// (16 B per lane where channel counts allow).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void act_grad_kernel(const float* __restrict__ g, int g_ld, const float* __restrict__ y, int y_ld,
                                                       float* __restrict__ out, int out_ld, long long total, int C, float slope) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long p = idx / C;
        const int c = (int)(idx % C);
        const float gv = g[p * g_ld + c];
        out[p * out_ld + c] = y[p * y_ld + c] > 0.f ? gv : gv * slope;
    }
}

__global__ __launch_bounds__(256) void act_grad4_kernel(const float* __restrict__ g, int g_ld, const float* __restrict__ y, int y_ld,
                                                        float* __restrict__ out, int out_ld, long long total4, int C4, float slope) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long long)gridDim.x * 256) {
        const long long p = idx / C4;
        const int c = (int)(idx % C4) * 4;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + p * g_ld + c);
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + p * y_ld + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = yv[e] > 0.f ? gv[e] : gv[e] * slope;
        *reinterpret_cast<f32x4*>(out + p * out_ld + c) = o;
    }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ a, int a_ld, float* __restrict__ out, int out_ld,
                                                            long long total, int C, int accumulate) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long p = idx / C;
        const int c = (int)(idx % C);
        const float v = a[p * a_ld + c];
        float* d = out + p * out_ld + c;
        *d = accumulate ? *d + v : v;
    }
}

__global__ __launch_bounds__(256) void copy_channels4_kernel(const float* __restrict__ a, int a_ld, float* __restrict__ out, int out_ld,
                                                             long long total4, int C4, int accumulate) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long long)gridDim.x * 256) {
        const long long p = idx / C4;
        const int c = (int)(idx % C4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(a + p * a_ld + c);
        f32x4* d = reinterpret_cast<f32x4*>(out + p * out_ld + c);
        if (accumulate) {
            f32x4 o = *d;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += o[e];
        }
        *d = v;
    }
}

// bilinear x2, align_corners=False: out[2i] = .25 in[i-1] + .75 in[i], out[2i+1] = .75 in[i] + .25 in[i+1], clamped
__device__ __forceinline__ void up_src(int o, int n, int& i0, int& i1, float& w0, float& w1) {
    const int i = o >> 1;
    if (o & 1) { i0 = i; i1 = min(i + 1, n - 1); w0 = 0.75f; w1 = 0.25f; }
    else { i0 = max(i - 1, 0); i1 = i; w0 = 0.25f; w1 = 0.75f; }
}

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int out_ld,
                                                             int B, int H, int W, int C) {
    const long long total = (long long)B * 2 * H * 2 * W * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long long t = idx / C;
        const int ox = (int)(t % (2 * W));
        t /= 2 * W;
        const int oy = (int)(t % (2 * H));
        const int b = (int)(t / (2 * H));
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        up_src(oy, H, y0, y1, wy0, wy1);
        up_src(ox, W, x0, x1, wx0, wx1);
        const float* base = in + (long long)b * H * W * in_ld + c;
        const float v00 = base[((long long)y0 * W + x0) * in_ld], v01 = base[((long long)y0 * W + x1) * in_ld];
        const float v10 = base[((long long)y1 * W + x0) * in_ld], v11 = base[((long long)y1 * W + x1) * in_ld];
        out[(((long long)b * 2 * H + oy) * 2 * W + ox) * out_ld + c] = wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
    }
}

// Four channels of one output pixel per thread, 32-bit index arithmetic: four 16-byte reads (the input is a quarter of the
// output: L1 / L2 hits) and one 16-byte store.  The scalar kernel above (one float per thread, three 64-bit divisions, four
// dword gathers) wrote the 67 MB of the largest decoder level in 83 us (1.0 TB/s).  Same products and sums per value.
__global__ __launch_bounds__(256) void upsample2x_fwd4_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int out_ld,
                                                              int B, int H, int W, int C4) {
    const unsigned total = (unsigned)B * 4u * H * W * C4;
    for (unsigned idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const unsigned c = (idx % (unsigned)C4) * 4;
        unsigned t = idx / (unsigned)C4;
        const int ox = (int)(t % (unsigned)(2 * W));
        t /= (unsigned)(2 * W);
        const int oy = (int)(t % (unsigned)(2 * H));
        const int b = (int)(t / (unsigned)(2 * H));
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        up_src(oy, H, y0, y1, wy0, wy1);
        up_src(ox, W, x0, x1, wx0, wx1);
        const float* base = in + (long long)b * H * W * in_ld + c;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + (long long)(y0 * W + x0) * in_ld);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + (long long)(y0 * W + x1) * in_ld);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + (long long)(y1 * W + x0) * in_ld);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + (long long)(y1 * W + x1) * in_ld);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
        *reinterpret_cast<f32x4*>(out + ((long long)(b * 2 * H + oy) * (2 * W) + ox) * out_ld + c) = o;
    }
}

// adjoint: gin[i] = sum over the (<= 4 per axis) outputs that read i.  Gather form => deterministic.
__device__ __forceinline__ int up_adj(int i, int n, int (&o)[4], float (&w)[4]) {
    // outputs reading input i:  o=2i (w .75), o=2i+1 (w .75), o=2i+2 (w .25, via i0=i), o=2i-1 (w .25, via i1=i)
    // plus the clamped edges: o=0 reads in[0] with extra .25, o=2n-1 reads in[n-1] with extra .25
    int k = 0;
    o[k] = 2 * i; w[k++] = 0.75f;
    o[k] = 2 * i + 1; w[k++] = 0.75f;
    if (i + 1 <= n - 1) { o[k] = 2 * i + 2; w[k++] = 0.25f; } else { o[k] = 2 * i + 1; w[k++] = 0.25f; }   // clamp at the end
    if (i - 1 >= 0) { o[k] = 2 * i - 1; w[k++] = 0.25f; } else { o[k] = 0; w[k++] = 0.25f; }                  // clamp at the start
    return k;
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ gout, int gout_ld, float* __restrict__ gin, int gin_ld,
                                                             int B, int H, int W, int C) {
    const long long total = (long long)B * H * W * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long long t = idx / C;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        int oy[4], ox[4];
        float wy[4], wx[4];
        const int ny = up_adj(y, H, oy, wy), nx = up_adj(x, W, ox, wx);
        const float* base = gout + (long long)b * 4 * H * W * gout_ld + c;
        float s = 0.f;
        for (int i = 0; i < ny; ++i) {
            float r = 0.f;
            for (int j = 0; j < nx; ++j) r += wx[j] * base[((long long)oy[i] * 2 * W + ox[j]) * gout_ld];
            s += wy[i] * r;
        }
        gin[(((long long)b * H + y) * W + x) * gin_ld + c] = s;
    }
}

// four channels per thread, optional LeakyReLU-gradient mask of the consumer fused in (gin = upsampled^T(gout) * (y > 0 ? 1 : slope))
__global__ __launch_bounds__(256) void upsample2x_bwd4_kernel(const float* __restrict__ gout, int gout_ld, float* __restrict__ gin, int gin_ld,
                                                              const float* __restrict__ y, int y_ld, float slope, int B, int H, int W, int C4) {
    const unsigned total = (unsigned)B * H * W * C4;
    for (unsigned idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const unsigned c = (idx % (unsigned)C4) * 4;
        unsigned t = idx / (unsigned)C4;
        const int x = (int)(t % (unsigned)W);
        t /= (unsigned)W;
        const int yy = (int)(t % (unsigned)H);
        const int b = (int)(t / (unsigned)H);
        int oy[4], ox[4];
        float wy[4], wx[4];
        const int ny = up_adj(yy, H, oy, wy), nx = up_adj(x, W, ox, wx);
        const float* base = gout + (long long)b * 4 * H * W * gout_ld + c;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < ny; ++i) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < nx; ++j) r += wx[j] * *reinterpret_cast<const f32x4*>(base + ((long long)oy[i] * 2 * W + ox[j]) * gout_ld);
            s += wy[i] * r;
        }
        const long long pix = ((long long)b * H + yy) * W + x;
        if (y) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + pix * y_ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = yv[e] > 0.f ? s[e] : s[e] * slope;
        }
        *reinterpret_cast<f32x4*>(gin + pix * gin_ld + c) = s;
    }
}

// PixelShuffle(2): out[b, 2h+i, 2w+j, c] = in[b, h, w, 4c + 2i + j]
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld,
                                                             int B, int H, int W, int C, int inverse) {
    const long long total = (long long)B * 2 * H * 2 * W * C;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long long t = idx / C;
        const int ox = (int)(t % (2 * W));
        t /= 2 * W;
        const int oy = (int)(t % (2 * H));
        const int b = (int)(t / (2 * H));
        const long long big = (((long long)b * 2 * H + oy) * 2 * W + ox);           // pixel in the 2H x 2W image
        const long long small = (((long long)b * H + (oy >> 1)) * W + (ox >> 1));  // pixel in the H x W image
        const int cs = 4 * c + 2 * (oy & 1) + (ox & 1);
        if (!inverse) dst[big * dst_ld + c] = src[small * src_ld + cs];
        else dst[small * dst_ld + cs] = src[big * src_ld + c];
    }
}

// One thread per 16 bytes of the H x W side (channels 4c .. 4c+3 = the four sub-pixels of output channel c): consecutive
// lanes are consecutive c, so the wide side moves as 1 KB per wave and each of the four scalar accesses of the 2H x 2W side
// covers 256 contiguous bytes.  (The kernel above walks the big side and gathers dwords 16 bytes apart from the other.)
__global__ __launch_bounds__(256) void pixel_shuffle2_vec_kernel(const float* __restrict__ small, int small_ld, float* __restrict__ big,
                                                                 int big_ld, int B, int H, int W, int C, int inverse) {
    const unsigned total = (unsigned)B * H * W * C;
    for (unsigned idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const unsigned c = idx % (unsigned)C;
        unsigned t = idx / (unsigned)C;
        const int x = (int)(t % (unsigned)W);
        t /= (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        const int b = (int)(t / (unsigned)H);
        float* sp = const_cast<float*>(small) + ((long long)(b * H + y) * W + x) * small_ld + 4 * c;
        float* bp = big + ((long long)(b * 2 * H + 2 * y) * (2 * W) + 2 * x) * big_ld + c;
        const long long row = (long long)2 * W * big_ld;
        if (!inverse) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sp);
            bp[0] = v[0];
            bp[big_ld] = v[1];
            bp[row] = v[2];
            bp[row + big_ld] = v[3];
        } else {
            const f32x4 v = {bp[0], bp[big_ld], bp[row], bp[row + big_ld]};
            *reinterpret_cast<f32x4*>(sp) = v;
        }
    }
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] * b[i];
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void add4_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) out[i] = a[i] + b[i];
}

// [tap][n][c] re-layout of weight views W(n,c,t) = src[n*sn + c*sc + t].  A workgroup owns a 32 n x 32 c tile with all its
// taps: it reads the tile in the source's own order (for OIHW storage the 32 c x T taps of one n are one contiguous run, for
// a transposed-conv view the 32 n x T taps of one c), turns it in LDS and writes 128-byte rows of c.  (The first version
// read one element per thread in destination order: every source line was fetched once per tap, 1 TB/s.)
constexpr int PACK_MAXT = 16;
constexpr int PACK_TS = 552;                   // LDS floats per tap: 16 rows of 33 (or 32 rows of 17) + pad, taps eight banks apart
__device__ __forceinline__ int pack_div(int x, float inv) { return __float2int_rz(((float)x + 0.5f) * inv); }   // x < 2^14, exact
// tile: 16 values of the source's slow index x 32 of its fast index x all taps (2144 B of LDS per tap: 4 workgroups per CU
// with the 4x4 layers in the launch)
__global__ __launch_bounds__(256) void pack_weights_kernel(const mtd_pack_desc* __restrict__ D, int count) {
    extern __shared__ float tile[];               // [max T of the launch][PACK_TS]
    int acc = 0, di = -1, local = 0;
    for (int t = 0; t < count; ++t) {
        const int nb = (D[t].N >> 5) * (D[t].C >> 5) * 2;
        if ((int)blockIdx.x < acc + nb) { di = t; local = blockIdx.x - acc; break; }
        acc += nb;
    }
    if (di < 0) return;
    const mtd_pack_desc d = D[di];
    const bool c_inner = d.sc <= d.sn;            // which of n / c is the faster index in the source
    const int ctiles = c_inner ? (d.C >> 5) : (d.C >> 4);
    const int n0 = (local / ctiles) << (c_inner ? 4 : 5), c0 = (local % ctiles) << (c_inner ? 5 : 4);
    const int T = d.T, per = 32 * T, total = 16 * per;
    const float inv_per = 1.f / (float)per, inv_t = 1.f / (float)T;
    const int nrow = c_inner ? 33 : 17;            // LDS row stride of an n
    const float* src = d.src + (long long)n0 * d.sn + (long long)c0 * d.sc;
    // OIHW storage (the 32 c x T taps of one n are one contiguous, 16-byte aligned run): 16-byte reads, four at a time
    const bool vec = c_inner && d.sc == T && (d.sn % 4) == 0 && ((((uintptr_t)d.src) & 15) == 0) && (d.C % 4) == 0 &&
                     ((((uintptr_t)d.dst) & 15) == 0);
    if (vec) {
        const int per4 = per / 4, total4 = 16 * per4;
        for (int base = 0; base < total4; base += 4 * 256) {
            f32x4 v[4];
            int i4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                i4[u] = base + u * 256 + threadIdx.x;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i4[u] < total4) {
                    const int outer = i4[u] / per4, r4 = i4[u] - outer * per4;
                    v[u] = *reinterpret_cast<const f32x4*>(src + (long long)outer * d.sn + 4 * r4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i4[u] < total4) {
                    const int outer = i4[u] / per4, rem = 4 * (i4[u] - outer * per4);
                    int inner = pack_div(rem, inv_t), t = rem - inner * T;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        tile[t * PACK_TS + outer * nrow + inner] = v[u][e];
                        if (++t == T) { t = 0; ++inner; }
                    }
                }
            }
        }
        __syncthreads();
        float* dst = d.dst + (long long)n0 * d.C + c0;
        const long long tstride = (long long)d.N * d.C;
        for (int i = threadIdx.x; i < total / 4; i += 256) {       // rows of 32 c as eight 16-byte stores
            const int c = (i & 7) * 4, n = (i >> 3) & 15, t = i >> 7;
            const float* tp = tile + t * PACK_TS + n * nrow + c;
            *reinterpret_cast<f32x4*>(dst + t * tstride + (long long)n * d.C + c) = f32x4{tp[0], tp[1], tp[2], tp[3]};
        }
        return;
    }
    for (int base = 0; base < total; base += 8 * 256) {      // eight loads in flight per thread
        float v[8];
        int li[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + threadIdx.x;
            li[u] = -1;
            v[u] = 0.f;
            if (i < total) {
                const int outer = pack_div(i, inv_per), rem = i - outer * per;
                const int inner = pack_div(rem, inv_t), t = rem - inner * T;
                const int n = c_inner ? outer : inner, c = c_inner ? inner : outer;
                v[u] = src[(long long)n * d.sn + (long long)c * d.sc + t];
                li[u] = t * PACK_TS + n * nrow + c;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (li[u] >= 0) tile[li[u]] = v[u];
    }
    __syncthreads();
    float* dst = d.dst + (long long)n0 * d.C + c0;
    const long long tstride = (long long)d.N * d.C;
    const int cbits = c_inner ? 5 : 4, cmask = (1 << cbits) - 1, nmask = c_inner ? 15 : 31;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i & cmask, n = (i >> cbits) & nmask, t = i >> 9;
        dst[t * tstride + (long long)n * d.C + c] = tile[t * PACK_TS + n * nrow + c];
    }
}

// ---- the step's small bookkeeping ops as library launches (so that a recorded launch list holds the whole iteration) ----
__global__ __launch_bounds__(256) void dropout_mask_kernel(const float* __restrict__ r, float p, float keep_scale, float* __restrict__ out,
                                                           long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = r[i] >= p ? keep_scale : 0.f;
}

__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ a, const float* __restrict__ s, float* __restrict__ out, long long n) {
    const float f = s[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = a[i] * f;
}

// out[i] = sum(a[0 .. na)) + sum(b[0 .. nb)), left to right in fp32 (a handful of scalars: task losses, logged values)
__global__ __launch_bounds__(64) void scalar_sums_kernel(const mtd_sum_desc* __restrict__ T, int count, float* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const mtd_sum_desc d = T[i];
    float acc = 0.f;
    for (int j = 0; j < d.na; ++j) acc += d.a[j];
    for (int j = 0; j < d.nb; ++j) acc += d.b[j];
    out[i] = acc;
}

// zero many small tensors in one launch: block b clears floats [256 b, 256 b + 256) of the concatenation
__global__ __launch_bounds__(256) void zero_multi_kernel(const mtd_zero_desc* __restrict__ T, int count) {
    long long first = (long long)blockIdx.x * 1024;
    int t = 0;
    long long base = 0;
    while (t < count && first >= base + ((T[t].n + 1023) / 1024) * 1024) { base += ((T[t].n + 1023) / 1024) * 1024; ++t; }
    if (t >= count) return;
    const mtd_zero_desc d = T[t];
    const long long local = first - base;
    for (int i = threadIdx.x; i < 1024; i += 256)
        if (local + i < d.n) d.p[local + i] = 0.f;
}

// sums[t] = sum of the 32-bit patterns of tensor t (mod 2^64): an integer checksum, so the order of the atomic additions
// does not matter -- equal tensors give equal sums on every rank, whatever the timing (replica agreement, parallel.py).
// Block b takes floats [4096 b', 4096 b' + 4096) of its tensor (same block -> tensor walk as zero_multi_kernel).
__global__ __launch_bounds__(256) void checksum_multi_kernel(const mtd_zero_desc* __restrict__ T, int count, unsigned long long* __restrict__ sums) {
    long long first = (long long)blockIdx.x * 4096;
    int t = 0;
    long long base = 0;
    while (t < count && first >= base + ((T[t].n + 4095) / 4096) * 4096) { base += ((T[t].n + 4095) / 4096) * 4096; ++t; }
    if (t >= count) return;
    const mtd_zero_desc d = T[t];
    const long long local = first - base;
    const unsigned* bits = reinterpret_cast<const unsigned*>(d.p);
    unsigned long long acc = 0;
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (local + i < d.n) acc += bits[local + i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sums[t], acc);
}
__global__ void zero_u64_kernel(unsigned long long* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int mtd_act_grad(const float* g, int g_ld, const float* y, int y_ld, float* out, int out_ld, long long npix, int C,
                            float slope, void* stream) {
    if (!g || !y || !out || npix <= 0 || C <= 0 || g_ld < C || y_ld < C || out_ld < C) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((C % 4 == 0) && (g_ld % 4 == 0) && (y_ld % 4 == 0) && (out_ld % 4 == 0) && aligned16(g) && aligned16(y) && aligned16(out)) {
        long long total4 = npix * (C / 4);
        hipLaunchKernelGGL(act_grad4_kernel, dim3(grid_for(total4)), dim3(256), 0, s, g, g_ld, y, y_ld, out, out_ld, total4, C / 4, slope);
    } else {
        long long total = npix * C;
        hipLaunchKernelGGL(act_grad_kernel, dim3(grid_for(total)), dim3(256), 0, s, g, g_ld, y, y_ld, out, out_ld, total, C, slope);
    }
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_mul(const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// out[i] = r[i] >= p ? keep_scale : 0: the multiplier of nn.Dropout(p) (networks.py c_drop, train mode) from uniform draws r
// (the draws stay torch's generator: `torch.manual_seed` governs them as in the reference); keep_scale = 1 / (1 - p).
extern "C" int mtd_dropout_mask(const float* r, float p, float keep_scale, float* out, long long n, void* stream) {
    if (!r || !out || n <= 0 || !(p >= 0.f && p < 1.f)) return MTD_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, r, p, keep_scale, out, n);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// out = a * s[0] (s: one float in device memory): the upstream scalar of g_loss.backward() applied to the cotangent
extern "C" int mtd_scale_by(const float* a, const float* s, float* out, long long n, void* stream) {
    if (!a || !s || !out || n <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, s, out, n);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_scalar_sums(const mtd_sum_desc* table_dev, int count, float* out, void* stream) {
    if (!table_dev || !out || count <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(scalar_sums_kernel, dim3((count + 63) / 64), dim3(64), 0, (hipStream_t)stream, table_dev, count, out);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_zero_multi(const mtd_zero_desc* table_dev, const mtd_zero_desc* table_host, int count, void* stream) {
    if (!table_dev || !table_host || count <= 0) return MTD_EINVAL;
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!table_host[i].p || table_host[i].n <= 0) return MTD_EINVAL;
        blocks += (table_host[i].n + 1023) / 1024;
    }
    if (blocks >= (1ll << 31)) return MTD_EINVAL;
    hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table_dev, count);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_checksum_multi(const mtd_zero_desc* table_dev, const mtd_zero_desc* table_host, int count, unsigned long long* sums, void* stream) {
    if (!table_dev || !table_host || !sums || count <= 0) return MTD_EINVAL;
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!table_host[i].p || table_host[i].n <= 0) return MTD_EINVAL;
        blocks += (table_host[i].n + 4095) / 4096;
    }
    if (blocks >= (1ll << 31)) return MTD_EINVAL;
    hipLaunchKernelGGL(zero_u64_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count);
    hipLaunchKernelGGL(checksum_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table_dev, count, sums);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_add(const float* a, const float* b, float* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0) return MTD_EINVAL;
    if ((n % 4) == 0 && aligned16(a) && aligned16(b) && aligned16(out))
        hipLaunchKernelGGL(add4_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)a, (const f32x4*)b, (f32x4*)out, n / 4);
    else
        hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// Descriptor tables travel host -> device through this kernel instead of hipMemcpyAsync: the copy engine path costs
// a ~0.3 ms bubble in the stream per copy on ROCm 7.2, a kernel that reads the (device-mapped) pinned source does not.
__global__ __launch_bounds__(256) void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

extern "C" int mtd_upload(const void* src_pinned, void* dst, size_t bytes, void* stream) {
    if (!src_pinned || !dst || bytes == 0 || (bytes & 15) || (((uintptr_t)src_pinned | (uintptr_t)dst) & 15)) return MTD_EINVAL;
    if (bytes > (1u << 30)) return MTD_EINVAL;
    const int n16 = (int)(bytes / 16);
    int blocks = (n16 + 255) / 256;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(upload_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_pinned, (uint4*)dst, n16);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_pack_weights(const mtd_pack_desc* table_dev, const mtd_pack_desc* table_host, int count, void* stream) {
    if (!table_dev || !table_host || count <= 0) return MTD_EINVAL;
    long long blocks = 0;
    int maxT = 1;
    for (int i = 0; i < count; ++i) {
        if (!table_host[i].src || !table_host[i].dst || table_host[i].N <= 0 || table_host[i].C <= 0 || table_host[i].T <= 0) return MTD_EINVAL;
        if ((table_host[i].N % 32) || (table_host[i].C % 32) || table_host[i].T > PACK_MAXT) return MTD_EINVAL;
        blocks += (long long)(table_host[i].N / 32) * (table_host[i].C / 32) * 2;
        if (table_host[i].T > maxT) maxT = table_host[i].T;
    }
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(256), (size_t)maxT * PACK_TS * sizeof(float), (hipStream_t)stream,
                       table_dev, count);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_copy_channels(const float* a, int a_ld, float* out, int out_ld, long long npix, int C, int accumulate, void* stream) {
    if (!a || !out || npix <= 0 || C <= 0 || a_ld < C || out_ld < C) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((C % 4 == 0) && (a_ld % 4 == 0) && (out_ld % 4 == 0) && aligned16(a) && aligned16(out)) {
        long long total4 = npix * (C / 4);
        hipLaunchKernelGGL(copy_channels4_kernel, dim3(grid_for(total4)), dim3(256), 0, s, a, a_ld, out, out_ld, total4, C / 4, accumulate);
    } else {
        long long total = npix * C;
        hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, a_ld, out, out_ld, total, C, accumulate);
    }
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_upsample2x_fwd(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || in_ld < C || out_ld < C) return MTD_EINVAL;
    if ((C % 4) == 0 && (in_ld % 4) == 0 && (out_ld % 4) == 0 && aligned16(in) && aligned16(out) && (long long)B * 4 * H * W * C < (1ll << 32))
        hipLaunchKernelGGL(upsample2x_fwd4_kernel, dim3(grid_for((long long)B * 4 * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, in,
                           in_ld, out, out_ld, B, H, W, C / 4);
    else
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((long long)B * 4 * H * W * C)), dim3(256), 0, (hipStream_t)stream, in, in_ld,
                       out, out_ld, B, H, W, C);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_upsample2x_bwd(const float* gout, int gout_ld, float* gin, int gin_ld, int B, int H, int W, int C, void* stream) {
    if (!gout || !gin || B <= 0 || H <= 0 || W <= 0 || C <= 0 || gout_ld < C || gin_ld < C) return MTD_EINVAL;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, gout, gout_ld,
                       gin, gin_ld, B, H, W, C);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_upsample2x_bwd_masked(const float* gout, int gout_ld, float* gin, int gin_ld, const float* y, int y_ld, float slope,
                                         int B, int H, int W, int C, void* stream) {
    if (!gout || !gin || B <= 0 || H <= 0 || W <= 0 || C <= 0 || gout_ld < C || gin_ld < C || (y && y_ld < C)) return MTD_EINVAL;
    if ((C % 4) || (gout_ld % 4) || (gin_ld % 4) || (y && (y_ld % 4)) || !aligned16(gout) || !aligned16(gin) || (y && !aligned16(y)) ||
        (long long)B * H * W * C >= (1ll << 32))
        return MTD_EALIGN;
    hipLaunchKernelGGL(upsample2x_bwd4_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, gout,
                       gout_ld, gin, gin_ld, y, y_ld, slope, B, H, W, C / 4);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_pixel_shuffle2_fwd(const float* in, int in_ld, float* out, int out_ld, int B, int H, int W, int C, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || in_ld < 4 * C || out_ld < C) return MTD_EINVAL;
    if ((in_ld % 4) == 0 && aligned16(in) && (long long)B * 4 * H * W * C < (1ll << 32))
        hipLaunchKernelGGL(pixel_shuffle2_vec_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, in, in_ld,
                           out, out_ld, B, H, W, C, 0);
    else
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for((long long)B * 4 * H * W * C)), dim3(256), 0, (hipStream_t)stream, in, in_ld,
                       out, out_ld, B, H, W, C, 0);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_pixel_shuffle2_bwd(const float* gout, int gout_ld, float* gin, int gin_ld, int B, int H, int W, int C, void* stream) {
    if (!gout || !gin || B <= 0 || H <= 0 || W <= 0 || C <= 0 || gout_ld < C || gin_ld < 4 * C) return MTD_EINVAL;
    if ((gin_ld % 4) == 0 && aligned16(gin) && (long long)B * 4 * H * W * C < (1ll << 32))
        hipLaunchKernelGGL(pixel_shuffle2_vec_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, gin, gin_ld,
                           const_cast<float*>(gout), gout_ld, B, H, W, C, 1);
    else
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for((long long)B * 4 * H * W * C)), dim3(256), 0, (hipStream_t)stream, gout,
                       gout_ld, gin, gin_ld, B, H, W, C, 1);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
