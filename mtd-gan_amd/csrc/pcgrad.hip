// PCGrad gradient surgery (module/weight_methods.py:449-464) as two HBM-bound passes over the task
// gradients: (1) the T x T Gram matrix of the ORIGINAL task gradients, (2) the merged
// gradient sum_k w_k g_k where the weights w come from replaying the reference's sequential projections
// on the Gram matrix.  The replay runs in one device thread (T <= 4), so the step has no host sync; the
// reference's python `random.shuffle` order is supplied by the host as T*T ints.
// Algorithmic bytes: pass 1 reads T*n*4, pass 2 reads T*n*4 and writes n*4.
#include "common.h"

namespace {

constexpr int MAXT = 4;
constexpr int NPAIR = MAXT * (MAXT + 1) / 2;
struct Vecs { const float* g[MAXT]; };

__global__ __launch_bounds__(256) void gram_partial_kernel(Vecs v, int T, long long n, int vec, double* __restrict__ partial) {
    __shared__ double red[256];
    float acc[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) acc[i] = 0.f;
    // fp32 accumulation over a short strided run per thread, fp64 across threads / blocks
    double dacc[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) dacc[i] = 0.0;
    int cnt = 0;
    // 16-byte loads when every vector allows them (vec = 1), the last n % 4 entries and unaligned vectors one by one
    const long long n4 = vec ? n / 4 : 0;
    for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * 256) {
        f32x4 x[MAXT];
#pragma unroll
        for (int a = 0; a < MAXT; ++a) x[a] = (a < T) ? *reinterpret_cast<const f32x4*>(v.g[a] + 4 * i4) : f32x4{0.f, 0.f, 0.f, 0.f};
        int pi = 0;
#pragma unroll
        for (int a = 0; a < MAXT; ++a)
#pragma unroll
            for (int b = a; b < MAXT; ++b) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[pi] = fmaf(x[a][j], x[b][j], acc[pi]);
                ++pi;
            }
        if (++cnt == 16) {
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) { dacc[i] += (double)acc[i]; acc[i] = 0.f; }
            cnt = 0;
        }
    }
    for (long long idx = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long long)gridDim.x * 256) {
        float x[MAXT];
#pragma unroll
        for (int a = 0; a < MAXT; ++a) x[a] = (a < T) ? v.g[a][idx] : 0.f;
        int pi = 0;
#pragma unroll
        for (int a = 0; a < MAXT; ++a)
#pragma unroll
            for (int b = a; b < MAXT; ++b) { acc[pi] = fmaf(x[a], x[b], acc[pi]); ++pi; }
        if (++cnt == 64) {
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) { dacc[i] += (double)acc[i]; acc[i] = 0.f; }
            cnt = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) dacc[i] += (double)acc[i];
    for (int i = 0; i < NPAIR; ++i) {
        red[threadIdx.x] = dacc[i];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(long long)blockIdx.x * NPAIR + i] = red[0];
        __syncthreads();
    }
}

// one block per pair: strided sums in a fixed order, then a tree (doubles, bit-reproducible)
__global__ __launch_bounds__(256) void gram_finish_kernel(const double* __restrict__ partial, int nblocks, int T, double* __restrict__ gram) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(long long)b * NPAIR + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    s = red[0];
    int pi = 0;
    for (int a = 0; a < MAXT; ++a)
        for (int b = a; b < MAXT; ++b) {
            if (pi == i && a < T && b < T) { gram[a * T + b] = s; gram[b * T + a] = s; }
            ++pi;
        }
}

// replay of PCGrad._project_conflicting on the Gram matrix (one thread)
__global__ void pcgrad_coeff_kernel(const double* __restrict__ gram, const int* __restrict__ orders, int T, float* __restrict__ coeff) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double w[MAXT] = {0, 0, 0, 0};
    for (int i = 0; i < T; ++i) {
        double c[MAXT] = {0, 0, 0, 0};
        c[i] = 1.0;
        for (int jj = 0; jj < T; ++jj) {
            const int j = orders[i * T + jj];
            double d = 0.0;
            for (int k = 0; k < T; ++k) d += c[k] * gram[k * T + j];
            if (d < 0.0) c[j] -= d / gram[j * T + j];
        }
        for (int k = 0; k < T; ++k) w[k] += c[k];
    }
    for (int k = 0; k < T; ++k) coeff[k] = (float)w[k];
}

__global__ __launch_bounds__(256) void combine_kernel(Vecs v, int T, long long n4, long long n, const float* __restrict__ coeff,
                                                      float* __restrict__ merged, float scale = 1.f) {
    float w[MAXT];
#pragma unroll
    for (int a = 0; a < MAXT; ++a) w[a] = (a < T) ? scale * coeff[a] : 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < MAXT; ++a) {
            if (a < T) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(v.g[a] + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = fmaf(w[a], x[e], s[e]);
            }
        }
        *reinterpret_cast<f32x4*>(merged + 4 * i) = s;
    }
    // tail
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int a = 0; a < T; ++a) s = fmaf(w[a], v.g[a][i], s);
        merged[i] = s;
    }
}

constexpr int GRAM_BLOCKS = 1024;

}  // namespace

extern "C" size_t mtd_pcgrad_ws_bytes(long long n, int T) {
    if (n <= 0 || T <= 0 || T > MAXT) return 0;
    return (size_t)GRAM_BLOCKS * NPAIR * sizeof(double);
}

extern "C" int mtd_pcgrad_gram(const float* g0, const float* g1, const float* g2, const float* g3, int T, long long n, double* gram,
                               void* ws, void* stream) {
    if (T <= 0 || T > MAXT || n <= 0 || !gram || !ws || !g0) return MTD_EINVAL;
    Vecs v;
    v.g[0] = g0; v.g[1] = g1; v.g[2] = g2; v.g[3] = g3;
    for (int a = 0; a < T; ++a) if (!v.g[a]) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    long long want = (n + 255) / 256;
    int blocks = (int)(want < GRAM_BLOCKS ? want : GRAM_BLOCKS);
    int vec = 1;
    for (int a = 0; a < T; ++a) vec &= (v.g[a] && aligned16(v.g[a])) ? 1 : 0;
    if (vec) {
        want = (n / 4 + 255) / 256;
        blocks = (int)(want < GRAM_BLOCKS ? (want < 1 ? 1 : want) : GRAM_BLOCKS);
    }
    hipLaunchKernelGGL(gram_partial_kernel, dim3(blocks), dim3(256), 0, s, v, T, n, vec, (double*)ws);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_finish_kernel, dim3(NPAIR), dim3(256), 0, s, (const double*)ws, blocks, T, gram);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_pcgrad_combine(const float* g0, const float* g1, const float* g2, const float* g3, int T, long long n,
                                  const double* gram, const int* orders, float* merged, float* coeff_out, void* stream) {
    if (T <= 0 || T > MAXT || n <= 0 || !gram || !orders || !merged || !coeff_out || !g0) return MTD_EINVAL;
    Vecs v;
    v.g[0] = g0; v.g[1] = g1; v.g[2] = g2; v.g[3] = g3;
    for (int a = 0; a < T; ++a) {
        if (!v.g[a]) return MTD_EINVAL;
        if (!aligned16(v.g[a])) return MTD_EALIGN;
    }
    if (!aligned16(merged)) return MTD_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pcgrad_coeff_kernel, dim3(1), dim3(64), 0, s, gram, orders, T, coeff_out);
    MTD_LAUNCH_CHECK();
    long long n4 = n / 4;
    long long want = (n4 + 255) / 256;
    int blocks = (int)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
    hipLaunchKernelGGL(combine_kernel, dim3(blocks), dim3(256), 0, s, v, T, n4, n, coeff_out, merged);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

/* ---- the two halves of mtd_pcgrad_combine on their own (module/pcgrad.py's optimizer wrapper: one coefficient replay
 * over the whole flat vector, then one axpy per run of parameters with the same reduction) ---- */
extern "C" int mtd_pcgrad_coeff(const double* gram, const int* orders, int T, float* coeff_out, void* stream) {
    if (T <= 0 || T > MAXT || !gram || !orders || !coeff_out) return MTD_EINVAL;
    hipLaunchKernelGGL(pcgrad_coeff_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gram, orders, T, coeff_out);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_pcgrad_axpy(const float* g0, const float* g1, const float* g2, const float* g3, int T, long long n,
                               const float* coeff, float scale, float* merged, void* stream) {
    if (T <= 0 || T > MAXT || n <= 0 || !coeff || !merged || !g0) return MTD_EINVAL;
    Vecs v;
    v.g[0] = g0; v.g[1] = g1; v.g[2] = g2; v.g[3] = g3;
    for (int a = 0; a < T; ++a) {
        if (!v.g[a]) return MTD_EINVAL;
        if (!aligned16(v.g[a])) return MTD_EALIGN;
    }
    if (!aligned16(merged)) return MTD_EALIGN;
    const long long n4 = n / 4;
    const long long want = (n4 + 255) / 256;
    const int blocks = (int)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
    hipLaunchKernelGGL(combine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, T, n4, n, coeff, merged, scale);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
