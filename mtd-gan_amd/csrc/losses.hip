// Loss terms of the MTD-GAN step and their gradients (losses.py:10-15, 99-138; the F.l1_loss /
// F.mse_loss / .clip call sites at arch/Ours/networks.py:1962-1977, 1998-2002).  All are tiny HBM-bound
// reductions over (B,1) scores and (B,1,64,64) maps; they are batched through descriptor tables so a
// whole d_loss / g_loss costs two launches for the values and one for all output cotangents.
//
// term kinds (per element i of a[n]):
//   0  m_i * (a_i - t_i)^2        t = b_i if b else tconst;  m_i = (mx_i - my_i != 0) if mx else 1   (ls_gan / NDS_Loss / mse)
//   1  |a_i - b_i|                                                                             (F.l1_loss)
//   2  sqrt((a_i - b_i)^2 + eps^2)                                                             (CharbonnierLoss)
// value = scale * sum_i term_i.   grad:  out_i (+)= coef * d term_i / d a_i.
// EdgeLoss: Laplacian-pyramid residual lap(e) = e - G(M(G(e))), G = 5x5 Gaussian with replicate padding,
// M = keep even pixels x4; the kernel works on e = a - b (lap is linear) with one 64x64 image per workgroup.
#include "common.h"

namespace {

constexpr int TERM_BLOCKS = 64;

__device__ __forceinline__ float term_value(const mtd_loss_term& t, long long i) {
    const float a = t.a[i];
    if (t.kind == 0) {
        const float tv = t.b ? t.b[i] : t.tconst;
        const float d = a - tv;
        float v = d * d;
        if (t.mx && !(t.mx[i] - t.my[i] != 0.f)) v = 0.f;
        return v;
    } else if (t.kind == 1) {
        return fabsf(a - t.b[i]);
    } else {
        const float d = a - t.b[i];
        return sqrtf(d * d + t.eps * t.eps);
    }
}

__device__ __forceinline__ float term_grad(const mtd_loss_term& t, long long i) {
    const float a = t.a[i];
    if (t.kind == 0) {
        const float tv = t.b ? t.b[i] : t.tconst;
        float g = 2.f * (a - tv);
        if (t.mx && !(t.mx[i] - t.my[i] != 0.f)) g = 0.f;
        return g;
    } else if (t.kind == 1) {
        const float d = a - t.b[i];
        return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    } else {
        const float d = a - t.b[i];
        return d / sqrtf(d * d + t.eps * t.eps);
    }
}

__global__ __launch_bounds__(256) void term_partial_kernel(const mtd_loss_term* __restrict__ T, double* __restrict__ partial) {
    __shared__ double red[256];
    const mtd_loss_term t = T[blockIdx.y];
    float acc = 0.f;
    double dacc = 0.0;
    int cnt = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (long long)gridDim.x * 256) {
        acc += term_value(t, i);
        if (++cnt == 32) { dacc += (double)acc; acc = 0.f; cnt = 0; }
    }
    dacc += (double)acc;
    red[threadIdx.x] = dacc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.y * TERM_BLOCKS + blockIdx.x] = red[0];
}

__global__ void term_finish_kernel(const mtd_loss_term* __restrict__ T, int nterms, const double* __restrict__ partial, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nterms) return;
    double s = 0.0;
    for (int b = 0; b < TERM_BLOCKS; ++b) s += partial[k * TERM_BLOCKS + b];
    out[k] = (float)(s * (double)T[k].scale);
}

__global__ __launch_bounds__(256) void term_grad_kernel(const mtd_loss_term* __restrict__ T) {
    const mtd_loss_term t = T[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < t.n; i += (long long)gridDim.x * 256) {
        const float g = t.coef * term_grad(t, i);
        t.grad_out[i] = t.accumulate ? t.grad_out[i] + g : g;
    }
}

__global__ __launch_bounds__(256) void clip01_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = fminf(fmaxf(x[i], 0.f), 1.f);
}
// torch clamp backward: gradient passes where min <= x <= max (inclusive)
__global__ __launch_bounds__(256) void clip01_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        out[i] = (v >= 0.f && v <= 1.f) ? g[i] : 0.f;
    }
}

// ---- EdgeLoss -----------------------------------------------------------------------------------
constexpr int ES = 64;   // image side
__device__ __constant__ const float GK[5] = {0.05f, 0.25f, 0.4f, 0.25f, 0.05f};

__device__ __forceinline__ int clampi(int v) { return v < 0 ? 0 : (v > ES - 1 ? ES - 1 : v); }

// dst = G(src) (separable, replicate padding); tmp is scratch.  All arrays ES*ES in LDS.
__device__ void gauss_fwd(const float* src, float* tmp, float* dst) {
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const int y = i / ES, x = i % ES;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) s += GK[t] * src[y * ES + clampi(x + t - 2)];
        tmp[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const int y = i / ES, x = i % ES;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) s += GK[t] * tmp[clampi(y + t - 2) * ES + x];
        dst[i] = s;
    }
    __syncthreads();
}

// adjoint of the 1-D clamped 5-tap filter along a line: in gather form (deterministic)
__device__ __forceinline__ float adj1d(const float* line, int stride, int s) {
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int i = s - t + 2;                 // outputs i that read input s through tap t without clamping
        if (i >= 0 && i < ES) r += GK[t] * line[i * stride];
    }
    if (s == 0) r += (GK[0] + GK[1]) * line[0] + GK[0] * line[stride];
    if (s == ES - 1) r += (GK[3] + GK[4]) * line[(ES - 1) * stride] + GK[4] * line[(ES - 2) * stride];
    return r;
}

__device__ void gauss_adj(const float* src, float* tmp, float* dst) {
    // forward was rows then columns, so the adjoint is columns then rows
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const int y = i / ES, x = i % ES;
        tmp[i] = adj1d(src + x, ES, y);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const int y = i / ES, x = i % ES;
        dst[i] = adj1d(tmp + y * ES, 1, x);
    }
    __syncthreads();
}

// one workgroup per image: loss partial sum and (optionally) gradient w.r.t. a
__global__ __launch_bounds__(256) void edge_loss_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partial,
                                                        float eps, float* __restrict__ grad_out, float coef, int accumulate) {
    __shared__ float E[ES * ES], T1[ES * ES], T2[ES * ES];
    __shared__ double red[256];
    const long long base = (long long)blockIdx.x * ES * ES;
    for (int i = threadIdx.x; i < ES * ES; i += 256) E[i] = a[base + i] - b[base + i];
    __syncthreads();
    gauss_fwd(E, T1, T2);                                    // T2 = G(e)
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const int y = i / ES, x = i % ES;
        T2[i] = ((y | x) & 1) ? 0.f : 4.f * T2[i];           // M
    }
    __syncthreads();
    gauss_fwd(T2, T1, T2);                                   // T2 = G(M(G(e)))  (in place is safe: rows read src into tmp first)
    double acc = 0.0;
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const float l = E[i] - T2[i];
        const float r = sqrtf(l * l + eps * eps);
        acc += (double)r;
        E[i] = l / r;                                        // d charbonnier / d lap
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
    if (!grad_out) return;
    // lap^T(g) = g - G^T(M(G^T(g)))
    gauss_adj(E, T1, T2);
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const int y = i / ES, x = i % ES;
        T2[i] = ((y | x) & 1) ? 0.f : 4.f * T2[i];
    }
    __syncthreads();
    gauss_adj(T2, T1, T2);
    for (int i = threadIdx.x; i < ES * ES; i += 256) {
        const float g = coef * (E[i] - T2[i]);
        grad_out[base + i] = accumulate ? grad_out[base + i] + g : g;
    }
}

__global__ void edge_finish_kernel(const double* __restrict__ partial, int B, float scale, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += partial[b];
    out[0] = (float)(s * (double)scale);
}

inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" size_t mtd_loss_terms_ws_bytes(int nterms) { return nterms > 0 ? (size_t)nterms * TERM_BLOCKS * sizeof(double) : 0; }

extern "C" int mtd_loss_terms(const void* terms_dev, int nterms, float* out, void* ws, void* stream) {
    if (!terms_dev || nterms <= 0 || !out || !ws) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(term_partial_kernel, dim3(TERM_BLOCKS, nterms), dim3(256), 0, s, (const mtd_loss_term*)terms_dev, (double*)ws);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(term_finish_kernel, dim3((nterms + 63) / 64), dim3(64), 0, s, (const mtd_loss_term*)terms_dev, nterms, (const double*)ws, out);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_loss_term_grads(const void* terms_dev, int nterms, void* stream) {
    if (!terms_dev || nterms <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(term_grad_kernel, dim3(128, nterms), dim3(256), 0, (hipStream_t)stream, (const mtd_loss_term*)terms_dev);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_clip01(const float* x, float* out, long long n, void* stream) {
    if (!x || !out || n <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(clip01_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_clip01_bwd(const float* g, const float* x, float* out, long long n, void* stream) {
    if (!g || !x || !out || n <= 0) return MTD_EINVAL;
    hipLaunchKernelGGL(clip01_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, x, out, n);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" size_t mtd_edge_loss_ws_bytes(int B) { return B > 0 ? (size_t)B * sizeof(double) : 0; }

extern "C" int mtd_edge_loss(const float* a, const float* b, int B, float scale, float eps, float* out, float* grad_out, float coef,
                             int accumulate, void* ws, void* stream) {
    if (!a || !b || B <= 0 || !out || !ws) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(edge_loss_kernel, dim3(B), dim3(256), 0, s, a, b, (double*)ws, eps, grad_out, coef, accumulate);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(edge_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, B, scale, out);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
