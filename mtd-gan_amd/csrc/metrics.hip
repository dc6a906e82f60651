// Pixel metrics of the evaluation loops (reference metrics.py:172-244, used by engine.py:78-183): squared error (RMSE,
// PSNR) and SSIM with the 11x11 Gaussian window (sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) of a batch of
// single-channel images, as sums over all pixels in double precision with a fixed-order reduction.
// HBM-bound: both images are read once (a 32x32 output tile loads a 42x42 halo of each into LDS; the Gaussian is applied
// separably to the five moment maps a, b, a^2, b^2, ab in LDS).  Algorithmic bytes: 8 per pixel.
#include "common.h"

namespace {

constexpr int MT = 32;            // output tile
constexpr int MR = 5;             // window radius
constexpr int MH = MT + 2 * MR;   // 42

__global__ __launch_bounds__(256) void image_metrics_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                                            int clip_a, double* __restrict__ partial) {
    __shared__ float As[MH * MH], Bs[MH * MH];
    __shared__ float Hs[5][MH * MT];
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * MT, ty0 = blockIdx.y * MT;
    const long long img = (long long)blockIdx.z * H * W;
    float gw[11];
    {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) { gw[k] = expf(-(float)((k - 5) * (k - 5)) / 4.5f); s += gw[k]; }
#pragma unroll
        for (int k = 0; k < 11; ++k) gw[k] /= s;
    }
    for (int e = tid; e < MH * MH; e += 256) {
        const int ly = e / MH, lx = e - ly * MH;
        const int y = ty0 + ly - MR, x = tx0 + lx - MR;
        float va = 0.f, vb = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            va = a[img + (long long)y * W + x];
            vb = b[img + (long long)y * W + x];
            if (clip_a) va = fminf(fmaxf(va, 0.f), 1.f);
        }
        As[e] = va;
        Bs[e] = vb;
    }
    __syncthreads();
    // horizontal pass: 42 rows x 32 columns, five moments
    for (int e = tid; e < MH * MT; e += 256) {
        const int ly = e / MT, lx = e - ly * MT;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float va = As[ly * MH + lx + k], vb = Bs[ly * MH + lx + k];
            m[0] = fmaf(gw[k], va, m[0]);
            m[1] = fmaf(gw[k], vb, m[1]);
            m[2] = fmaf(gw[k], va * va, m[2]);
            m[3] = fmaf(gw[k], vb * vb, m[3]);
            m[4] = fmaf(gw[k], va * vb, m[4]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) Hs[q][e] = m[q];
    }
    __syncthreads();
    double sse = 0.0, ssum = 0.0;
    for (int e = tid; e < MT * MT; e += 256) {
        const int ly = e / MT, lx = e - ly * MT;
        const int y = ty0 + ly, x = tx0 + lx;
        if (y < H && x < W) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 11; ++k)
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] = fmaf(gw[k], Hs[q][(ly + k) * MT + lx], m[q]);
            const float mu1 = m[0], mu2 = m[1];
            const float s11 = m[2] - mu1 * mu1, s22 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;
            const float c1 = 1e-4f, c2 = 9e-4f;
            const float v = ((2.f * mu1 * mu2 + c1) * (2.f * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2));
            ssum += (double)v;
            const float d = As[(ly + MR) * MH + lx + MR] - Bs[(ly + MR) * MH + lx + MR];
            sse += (double)d * (double)d;
        }
    }
    red[0][tid] = sse;
    red[1][tid] = ssum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = red[0][0];
        partial[2 * blk + 1] = red[1][0];
    }
}

__global__ __launch_bounds__(256) void image_metrics_finish_kernel(const double* __restrict__ partial, long long nblk, double* __restrict__ out2) {
    __shared__ double red[2][256];
    double s0 = 0.0, s1 = 0.0;
    for (long long i = threadIdx.x; i < nblk; i += 256) { s0 += partial[2 * i]; s1 += partial[2 * i + 1]; }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = red[0][0]; out2[1] = red[1][0]; }
}

long long metric_blocks(int B, int H, int W) { return (long long)B * ((H + MT - 1) / MT) * ((W + MT - 1) / MT); }

}  // namespace

extern "C" size_t mtd_image_metrics_ws_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)metric_blocks(B, H, W) * 2 * sizeof(double);
}

extern "C" int mtd_image_metrics(const float* a, const float* b, int B, int H, int W, int clip_a, double* out2, void* ws, void* stream) {
    if (!a || !b || !out2 || !ws || B <= 0 || H <= 0 || W <= 0 || B > 65535) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((W + MT - 1) / MT, (H + MT - 1) / MT, B);
    hipLaunchKernelGGL(image_metrics_kernel, grid, dim3(256), 0, s, a, b, H, W, clip_a, (double*)ws);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(image_metrics_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, metric_blocks(B, H, W), out2);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
