// Shared device helpers for the gfx950 kernels of libmtdgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/mtdgan_hip.h"

// Lab switches.  The kernel-selection / tuning environment variables of rounds 1-4 (MTD_WINO_*, MTD_WGRAD_*, MTD_IGEMM_*, ...)
// exist only in a library built with -DMTD_LAB (tools/ probes: `MTD_LAB_BUILD=1 python mtd-gan_amd/_build.py --force`).  The
// shipped library reads NO environment variable: a stray MTD_* in a user's shell cannot change which kernel runs.  The few
// options that are meant to be flipped at run time go through mtd_set_option() (api.hip), each exercised by a test.
#include <stdlib.h>
#ifdef MTD_LAB
static inline const char* mtd_lab_env(const char* name) { return getenv(name); }
#else
static inline const char* mtd_lab_env(const char*) { return nullptr; }
#endif
// run-time options (api.hip: mtd_set_option / mtd_get_option)
enum { MTD_OPT_C32F_SAFE_WAIT = 0, MTD_OPT_WINO_SPLIT, MTD_OPT_COUNT };
int mtd_option(int id);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MTD_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact f32 FMA chain.
// lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// result register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == MTD_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MTD_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    return v;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
static inline long long geom_pixels(const mtd_geom& g) { return (long long)g.B * g.OH * g.OW; }
// accumulator scale of a conv launch: one value, or two for the two batch halves of a paired discriminator pass
struct ScalePair { float s0, s1; int split; };
__device__ __forceinline__ ScalePair load_scale(const mtd_conv_args& a) {
    ScalePair r;
    r.s0 = a.scale ? *a.scale : 1.f;
    const bool two = a.scale2 != nullptr && a.scale_split > 0;
    r.s1 = two ? *a.scale2 : r.s0;
    r.split = two ? a.scale_split : 0x7fffffff;
    return r;
}
__device__ __forceinline__ float pick_scale(const ScalePair& s, int m) { return m < s.split ? s.s0 : s.s1; }

// Launch pixel m -> (image, row, column) of an OH x OW map: shifts and masks when both sides are powers of two (every map of this
// model), the divisions otherwise.  An integer division expands to ~25 vector instructions, and beside fp32 MFMAs those are paid in
// full; the branch is uniform.
__device__ __forceinline__ void pix_decompose(int m, int OW, int OH, int& b, int& oy, int& ox) {
    if (((OW & (OW - 1)) | (OH & (OH - 1))) == 0) {
        const int t = m >> __builtin_ctz(OW);
        ox = m & (OW - 1);
        oy = t & (OH - 1);
        b = t >> __builtin_ctz(OH);
    } else {
        ox = m % OW;
        const int t = m / OW;
        oy = t % OH;
        b = t / OH;
    }
}

// Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2).  Position of workgroup b in an order
// that gives every XCD one contiguous run of the nblk tiles: tiles that share operands then meet in the same L2.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int x = b & 7, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// One-launch, order-fixed sum of `ns` partial-result slabs (slab k at in + k * stride, float4 index i4) by a workgroup of
// 16 * SQ * SQ threads: thread (tx = tid & 15, ty = tid >> 4) adds the contiguous run of slabs ty * per .. of float4
// i4 = 16 * blockIdx.x + tx with eight loads in flight, the SQ * SQ run sums meet in LDS and are added in two levels of SQ
// (always the same association, so the result does not depend on timing).  Returns the total in the threads with
// ty == 0 (others get a partial); `red` needs 16 * SQ * (SQ + 1) float4.  i4 >= count4 contributes zeros.
template <int SQ>
__device__ __forceinline__ f32x4 block_slab_sum(const float* __restrict__ in, long long stride, int ns, long long i4, bool valid,
                                                f32x4* red) {
    constexpr int G = SQ * SQ;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int per = (ns + G - 1) / G;
    const int k0 = min(ns, ty * per), k1 = min(ns, k0 + per);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const float* base = in + 4 * i4;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (long long)(k + j) * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        if (k + 4 <= k1) {
            f32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (long long)(k + j) * stride);
#pragma unroll
            for (int j = 0; j < 4; ++j) s += v[j];
            k += 4;
        }
        for (; k < k1; ++k) s += *reinterpret_cast<const f32x4*>(base + (long long)k * stride);
    }
    red[ty * 16 + tx] = s;
    __syncthreads();
    f32x4* red2 = red + G * 16;
    if (ty < SQ) {
        f32x4 t = red[(ty * SQ) * 16 + tx];
#pragma unroll
        for (int j = 1; j < SQ; ++j) t += red[(ty * SQ + j) * 16 + tx];
        red2[ty * 16 + tx] = t;
    }
    __syncthreads();
    if (ty == 0) {
        s = red2[tx];
#pragma unroll
        for (int j = 1; j < SQ; ++j) s += red2[j * 16 + tx];
    }
    return s;
}

// launch profiler hooks (api.hip).  Between mtd_prof_begin and mtd_prof_end the main kernel is launched with MTD_LAUNCH:
// when the profiler is on (kernel-timestamp mode) the slot's two events travel on the kernel's dispatch packet.
int mtd_prof_begin(int kernel, int cfg, int splitk, long long M, int N, int C, int taps, hipStream_t s, double bytes = 0.0);
void mtd_prof_end(int slot, hipStream_t s);
struct MtdProfLaunch { hipEvent_t e0, e1; bool on; };
MtdProfLaunch mtd_prof_launch_events();
#define MTD_LAUNCH(kernel, grid, block, shmem, s, ...)                                                         \
    do {                                                                                                       \
        MtdProfLaunch pe__ = mtd_prof_launch_events();                                                         \
        if (pe__.on) hipExtLaunchKernelGGL(kernel, grid, block, shmem, s, pe__.e0, pe__.e1, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, s, __VA_ARGS__);                                   \
    } while (0)

