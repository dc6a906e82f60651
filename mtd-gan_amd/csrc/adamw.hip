// Fused multi-tensor AdamW (torch.optim.AdamW semantics as wired at train.py:122-126 / optimizers.py:8-9:
// decoupled weight decay, no amsgrad).  One launch updates every parameter tensor of a network:
// HBM-bound, algorithmic bytes = 4 reads + 3 writes of 4 B per element (p, g, m, v -> p, m, v).
#include "common.h"

namespace {

constexpr int AW_ELEMS = 2048;   // elements per block

__global__ __launch_bounds__(256) void adamw_kernel(const mtd_adamw_tensor* __restrict__ T, int count, float decay, float beta1,
                                                    float beta2, float step_size, float inv_sqrt_bc2, float eps,
                                                    const float* __restrict__ dyn) {
    if (dyn) {      // step-dependent scalars from device memory (hipGraph replay: kernel arguments are frozen)
        decay = dyn[0];
        step_size = dyn[1];
        inv_sqrt_bc2 = dyn[2];
    }
    int acc = 0, ti = -1, local = 0;
    for (int t = 0; t < count; ++t) {
        int nb = (int)((T[t].n + AW_ELEMS - 1) / AW_ELEMS);
        if ((int)blockIdx.x < acc + nb) { ti = t; local = blockIdx.x - acc; break; }
        acc += nb;
    }
    if (ti < 0) return;
    const mtd_adamw_tensor t = T[ti];
    const long long base = (long long)local * AW_ELEMS;
    for (int i = threadIdx.x; i < AW_ELEMS; i += 256) {
        const long long e = base + i;
        if (e < t.n) {
            const float g = t.g[e];
            float p = t.p[e] * decay;
            const float m = beta1 * t.m[e] + (1.f - beta1) * g;
            const float v = beta2 * t.v[e] + (1.f - beta2) * g * g;
            const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
            p -= step_size * (m / denom);
            t.p[e] = p;
            t.m[e] = m;
            t.v[e] = v;
        }
    }
}

}  // namespace

extern "C" int mtd_adamw_multi(const mtd_adamw_tensor* tensors_dev, const mtd_adamw_tensor* tensors_host, int count, float lr,
                               float beta1, float beta2, float eps, float wd, int step, void* stream) {
    if (!tensors_dev || !tensors_host || count <= 0 || step <= 0) return MTD_EINVAL;
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!tensors_host[i].p || !tensors_host[i].g || !tensors_host[i].m || !tensors_host[i].v || tensors_host[i].n <= 0) return MTD_EINVAL;
        blocks += (tensors_host[i].n + AW_ELEMS - 1) / AW_ELEMS;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const float decay = 1.f - lr * wd;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tensors_dev, count, decay, beta1, beta2,
                       step_size, inv_sqrt_bc2, eps, (const float*)nullptr);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// Same update with the three step-dependent scalars precomputed by the caller (FusedAdamW computes them once in
// double precision for both its eager and its captured path, so the two produce identical parameters).
extern "C" int mtd_adamw_multi_pre(const mtd_adamw_tensor* tensors_dev, const mtd_adamw_tensor* tensors_host, int count, float beta1,
                                   float beta2, float eps, float decay, float step_size, float inv_sqrt_bc2, void* stream) {
    if (!tensors_dev || !tensors_host || count <= 0) return MTD_EINVAL;
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!tensors_host[i].p || !tensors_host[i].g || !tensors_host[i].m || !tensors_host[i].v || tensors_host[i].n <= 0) return MTD_EINVAL;
        blocks += (tensors_host[i].n + AW_ELEMS - 1) / AW_ELEMS;
    }
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tensors_dev, count, decay, beta1, beta2,
                       step_size, inv_sqrt_bc2, eps, (const float*)nullptr);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// Same update with (decay, step_size, 1/sqrt(bias_correction2)) read from dyn[0..2] in device memory, so that a
// captured hipGraph can be replayed with a new step count: the host refreshes the three floats before each replay.
extern "C" int mtd_adamw_multi_dyn(const mtd_adamw_tensor* tensors_dev, const mtd_adamw_tensor* tensors_host, int count, float beta1,
                                   float beta2, float eps, const float* dyn, void* stream) {
    if (!tensors_dev || !tensors_host || count <= 0 || !dyn) return MTD_EINVAL;
    long long blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (!tensors_host[i].p || !tensors_host[i].g || !tensors_host[i].m || !tensors_host[i].v || tensors_host[i].n <= 0) return MTD_EINVAL;
        blocks += (tensors_host[i].n + AW_ELEMS - 1) / AW_ELEMS;
    }
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tensors_dev, count, 0.f, beta1, beta2, 0.f,
                       0.f, eps, dyn);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
