// Spectral normalisation (torch.nn.utils.spectral_norm, 1 power iteration, eps 1e-12) for all 45
// normalised layers of the multi-task discriminator in four launches per forward, HBM-bound:
//   t = W^T u                (column pass, coalesced over columns)        -> v = t / max(|t|, eps)
//   s = W v                  (row pass, one wave per row)                 -> u = s / max(|s|, eps)
//   sigma = u . s ; 1/sigma                                               (conv epilogues read 1/sigma)
// W is weight_orig viewed as (Cout, Cin*kh*kw) -- exactly PyTorch's OIHW storage, no repack; the
// normalised weight W/sigma is never materialised (the conv kernels scale their accumulators).
// Reductions are staged through fixed-order partial sums => bit-reproducible u, v, sigma.
// Backward:  g_orig = G/sigma - <G, W>/sigma^2 * u v^T  in three launches.
#include "common.h"

namespace {

constexpr float SN_EPS = 1e-12f;
constexpr int COLS_PER_BLOCK = 256;
constexpr int MAX_CHUNKS = 64;         // cols <= 16384
constexpr int ROWS_PER_BLOCK = 64;     // W^T u is split over row chunks as well, so the pass fills the chip
constexpr int MAX_ROW_CHUNKS = 32;     // rows <= 2048

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    float r = red[0];
    __syncthreads();
    return r;
}

// find (layer, local block) for a flat block index given per-layer block counts
template <typename F>
__device__ __forceinline__ bool locate(int n_layers, int bid, F count_of, int& layer, int& local) {
    int acc = 0;
    for (int l = 0; l < n_layers; ++l) {
        int c = count_of(l);
        if (bid < acc + c) { layer = l; local = bid - acc; return true; }
        acc += c;
    }
    return false;
}

__device__ __forceinline__ bool aligned16_dev(const void* p) { return (((unsigned long long)p) & 15ull) == 0; }

struct SnWs { float* t; float* partial; float* wv; float* tp; };

__device__ __forceinline__ long long col_offset(const mtd_sn_layer* L, int layer) {
    long long o = 0;
    for (int l = 0; l < layer; ++l) o += L[l].cols;
    return o;
}
__device__ __forceinline__ long long row_offset(const mtd_sn_layer* L, int layer) {
    long long o = 0;
    for (int l = 0; l < layer; ++l) o += L[l].rows;
    return o;
}

// partial column sums over one chunk of rows: tp[rc][k] = sum_{r in chunk rc} W[r][k] u[r]
__global__ __launch_bounds__(256) void sn_wtu_kernel(const mtd_sn_layer* __restrict__ L, int n_layers, SnWs ws) {
    int layer, local;
    auto blocks_of = [&](int l) {
        return ((L[l].cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK) * ((L[l].rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    };
    if (!locate(n_layers, blockIdx.x, blocks_of, layer, local)) return;
    const mtd_sn_layer ly = L[layer];
    const int nchunk = (ly.cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
    const int chunk = local % nchunk, rc = local / nchunk;
    const int k = chunk * COLS_PER_BLOCK + threadIdx.x;
    if (k >= ly.cols) return;
    const int r0 = rc * ROWS_PER_BLOCK, r1 = min(ly.rows, r0 + ROWS_PER_BLOCK);
    const float* w = ly.w + k;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = w[(long long)(r + i) * ly.cols];
        t0 = fmaf(x[0], ly.u[r], t0);
        t1 = fmaf(x[1], ly.u[r + 1], t1);
        t2 = fmaf(x[2], ly.u[r + 2], t2);
        t3 = fmaf(x[3], ly.u[r + 3], t3);
        t0 = fmaf(x[4], ly.u[r + 4], t0);
        t1 = fmaf(x[5], ly.u[r + 5], t1);
        t2 = fmaf(x[6], ly.u[r + 6], t2);
        t3 = fmaf(x[7], ly.u[r + 7], t3);
    }
    for (; r < r1; ++r) t0 = fmaf(w[(long long)r * ly.cols], ly.u[r], t0);
    ws.tp[col_offset(L, layer) * MAX_ROW_CHUNKS + (long long)rc * ly.cols + k] = (t0 + t1) + (t2 + t3);
}

// t[k] = sum of the row-chunk partials (fixed order), and the per-column-chunk partial of |t|^2
// rows_per_chunk: the rows one partial covers (ROWS_PER_BLOCK: sn_wtu_kernel's; FUSE_ROWS: sn_wv_wtu_kernel's).  from_s: the partials are
// W^T s of the previous iteration's UNNORMALISED s = W v (sn_wv_wtu_kernel; s is in ws.wv): t = (sum) / max(|s|, eps) = W^T u.
__global__ __launch_bounds__(256) void sn_tsum_kernel(const mtd_sn_layer* __restrict__ L, int n_layers, SnWs ws, int rows_per_chunk, int from_s) {
    __shared__ float red[256];
    int layer, chunk;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return (L[l].cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK; }, layer, chunk)) return;
    const mtd_sn_layer ly = L[layer];
    const int k = chunk * COLS_PER_BLOCK + threadIdx.x;
    const int nrc = (ly.rows + rows_per_chunk - 1) / rows_per_chunk;
    float inv_s = 1.f;
    if (from_s) {
        const float* wv = ws.wv + row_offset(L, layer);
        float q = 0.f;
        for (int r = threadIdx.x; r < ly.rows; r += 256) q += wv[r] * wv[r];      // (the association of sn_finish_kernel's |s|^2)
        inv_s = 1.f / fmaxf(sqrtf(block_sum(q, red)), SN_EPS);
    }
    float t = 0.f;
    if (k < ly.cols) {
        const float* src = ws.tp + col_offset(L, layer) * MAX_ROW_CHUNKS + k;
        for (int rc = 0; rc < nrc; ++rc) t += src[(long long)rc * ly.cols];
        t *= inv_s;
        ws.t[col_offset(L, layer) + k] = t;
    }
    float s = block_sum(t * t, red);
    if (threadIdx.x == 0) ws.partial[layer * MAX_CHUNKS + chunk] = s;
}

__global__ __launch_bounds__(256) void sn_norm_v_kernel(const mtd_sn_layer* __restrict__ L, int n_layers, SnWs ws) {
    int layer, chunk;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return (L[l].cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK; }, layer, chunk)) return;
    const mtd_sn_layer ly = L[layer];
    const int nchunk = (ly.cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
    float n2 = 0.f;
    for (int c = 0; c < nchunk; ++c) n2 += ws.partial[layer * MAX_CHUNKS + c];
    const float inv = 1.f / fmaxf(sqrtf(n2), SN_EPS);
    const int k = chunk * COLS_PER_BLOCK + threadIdx.x;
    if (k < ly.cols) {
        const float v = ws.t[col_offset(L, layer) + k] * inv;
        ly.v[k] = v;
        if (ly.v_save) ly.v_save[k] = v;
    }
}

__global__ __launch_bounds__(256) void sn_copy_v_kernel(const mtd_sn_layer* __restrict__ L, int n_layers) {
    int layer, chunk;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return (L[l].cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK; }, layer, chunk)) return;
    const mtd_sn_layer ly = L[layer];
    const int k = chunk * COLS_PER_BLOCK + threadIdx.x;
    if (k < ly.cols && ly.v_save) ly.v_save[k] = ly.v[k];
}

// one wave per row: s[r] = W[r,:] . v
__global__ __launch_bounds__(256) void sn_wv_kernel(const mtd_sn_layer* __restrict__ L, int n_layers, SnWs ws) {
    int layer, local;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return (L[l].rows + 3) / 4; }, layer, local)) return;
    const mtd_sn_layer ly = L[layer];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = local * 4 + wave;
    if (r >= ly.rows) return;
    const float* w = ly.w + (long long)r * ly.cols;
    float s = 0.f;
    if ((ly.cols & 3) == 0 && aligned16_dev(ly.w) && aligned16_dev(ly.v)) {
        float s1 = 0.f;
        const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
        const f32x4* v4 = reinterpret_cast<const f32x4*>(ly.v);
        const int n4 = ly.cols >> 2;
        int k = lane;
        for (; k + 64 < n4; k += 128) {
            const f32x4 a = w4[k], b = v4[k], c = w4[k + 64], d = v4[k + 64];
            s = fmaf(a[0], b[0], s); s = fmaf(a[1], b[1], s); s = fmaf(a[2], b[2], s); s = fmaf(a[3], b[3], s);
            s1 = fmaf(c[0], d[0], s1); s1 = fmaf(c[1], d[1], s1); s1 = fmaf(c[2], d[2], s1); s1 = fmaf(c[3], d[3], s1);
        }
        for (; k < n4; k += 64) {
            const f32x4 a = w4[k], b = v4[k];
            s = fmaf(a[0], b[0], s); s = fmaf(a[1], b[1], s); s = fmaf(a[2], b[2], s); s = fmaf(a[3], b[3], s);
        }
        s += s1;
    } else {
        for (int k = lane; k < ly.cols; k += 64) s = fmaf(w[k], ly.v[k], s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) ws.wv[row_offset(L, layer) + r] = s;
}

// Round 6 -- one pass over W for TWO products: s = W v (this iteration) and the partials of W^T s (the NEXT iteration's W^T u up to
// the scalar 1 / |s|, applied by sn_tsum_kernel).  The discriminator step runs its four power iterations back to back on the same
// weights (train_step.d_loss): 5 instead of 8 passes over the 260 MB.  A workgroup owns FUSE_ROWS rows; a thread holds its columns
// (float4 k4 = tid + 256 j) of four rows at a time in registers, the four dot products meet through shuffles + LDS in a fixed order,
// and the rows are used a second time from the registers for the column sums.
constexpr int FUSE_ROWS = 32;
#ifndef MTD_SN_FUSE_NT
#define MTD_SN_FUSE_NT 512
#endif
constexpr int FUSE_NT = MTD_SN_FUSE_NT;                  // threads per workgroup
constexpr int FUSE_NJ = (2304 + FUSE_NT - 1) / FUSE_NT;  // float4 columns per thread: cols <= 9216 (1024 -> 512 channels, 3x3) on the vector path
#ifndef MTD_SN_FUSE_GR
#define MTD_SN_FUSE_GR 1
#endif
constexpr int FUSE_GR = MTD_SN_FUSE_GR;   // rows a thread holds at a time (registers: (GR + 2) * 4 * NJ; measured in the step with 256-thread
                                          // workgroups: 4 rows / one wave per SIMD +0.37 ms, 2 rows +0.24 ms, 1 row / three waves -0.10 ms)

__global__ __launch_bounds__(FUSE_NT) void sn_wv_wtu_kernel(const mtd_sn_layer* __restrict__ L, int n_layers, SnWs ws) {
    constexpr int NWV = FUSE_NT / 64;
    __shared__ float red[NWV][FUSE_GR];
    __shared__ float srow[FUSE_ROWS];
    int layer, rb;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return (L[l].rows + FUSE_ROWS - 1) / FUSE_ROWS; }, layer, rb)) return;
    const mtd_sn_layer ly = L[layer];
    const int r0 = rb * FUSE_ROWS, r1 = min(ly.rows, r0 + FUSE_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wv = ws.wv + row_offset(L, layer);
    float* tp = ws.tp + col_offset(L, layer) * MAX_ROW_CHUNKS + (long long)rb * ly.cols;
    if ((ly.cols & 3) == 0 && ly.cols <= 4 * FUSE_NT * FUSE_NJ && aligned16_dev(ly.w) && aligned16_dev(ly.v)) {
        const int n4 = ly.cols >> 2;
        const f32x4* v4 = reinterpret_cast<const f32x4*>(ly.v);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 vv[FUSE_NJ], tacc[FUSE_NJ];
#pragma unroll
        for (int j = 0; j < FUSE_NJ; ++j) {
            const int k4 = tid + FUSE_NT * j;
            vv[j] = k4 < n4 ? v4[k4] : zero;
            tacc[j] = zero;
        }
        for (int r = r0; r < r1; r += FUSE_GR) {
            f32x4 wr[FUSE_GR][FUSE_NJ];
#pragma unroll
            for (int i = 0; i < FUSE_GR; ++i) {
                const f32x4* w4 = reinterpret_cast<const f32x4*>(ly.w + (long long)(r + i) * ly.cols);
#pragma unroll
                for (int j = 0; j < FUSE_NJ; ++j) {
                    const int k4 = tid + FUSE_NT * j;
                    wr[i][j] = (r + i < r1 && k4 < n4) ? w4[k4] : zero;
                }
            }
            float d[FUSE_GR];
#pragma unroll
            for (int i = 0; i < FUSE_GR; ++i) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < FUSE_NJ; ++j) {
                    a = fmaf(wr[i][j][0], vv[j][0], a); a = fmaf(wr[i][j][1], vv[j][1], a);
                    a = fmaf(wr[i][j][2], vv[j][2], a); a = fmaf(wr[i][j][3], vv[j][3], a);
                }
                for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
                d[i] = a;
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < FUSE_GR; ++i) red[wave][i] = d[i];
            }
            __syncthreads();
            float sr[FUSE_GR];
#pragma unroll
            for (int i = 0; i < FUSE_GR; ++i) {          // the waves' sums in a fixed order
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) t += red[w][i];
                sr[i] = t;
            }
            if (tid < FUSE_GR && r + tid < r1) wv[r + tid] = sr[tid];
#pragma unroll
            for (int j = 0; j < FUSE_NJ; ++j)
#pragma unroll
                for (int i = 0; i < FUSE_GR; ++i) {          // (rows past the block's end hold zeros)
                    tacc[j][0] = fmaf(wr[i][j][0], sr[i], tacc[j][0]); tacc[j][1] = fmaf(wr[i][j][1], sr[i], tacc[j][1]);
                    tacc[j][2] = fmaf(wr[i][j][2], sr[i], tacc[j][2]); tacc[j][3] = fmaf(wr[i][j][3], sr[i], tacc[j][3]);
                }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < FUSE_NJ; ++j) {
            const int k4 = tid + FUSE_NT * j;
            // (the partials' base is 4-byte aligned only: the column offsets of the layers before this one include conv11's 9)
            if (k4 < n4) { tp[4 * k4] = tacc[j][0]; tp[4 * k4 + 1] = tacc[j][1]; tp[4 * k4 + 2] = tacc[j][2]; tp[4 * k4 + 3] = tacc[j][3]; }
        }
    } else {
        // narrow or unaligned layers (conv11: 64 x 9): the rows once for s, once more (from L1 / L2) for the column sums
        __shared__ float redb[FUSE_NT];
        for (int r = r0; r < r1; ++r) {
            const float* w = ly.w + (long long)r * ly.cols;
            float a = 0.f;
            for (int k = tid; k < ly.cols; k += FUSE_NT) a = fmaf(w[k], ly.v[k], a);
            redb[tid] = a;
            __syncthreads();
            for (int st = FUSE_NT / 2; st > 0; st >>= 1) {
                if (tid < st) redb[tid] += redb[tid + st];
                __syncthreads();
            }
            if (tid == 0) { srow[r - r0] = redb[0]; wv[r] = redb[0]; }
            __syncthreads();
        }
        for (int k = tid; k < ly.cols; k += FUSE_NT) {
            float t = 0.f;
            for (int r = r0; r < r1; ++r) t = fmaf(ly.w[(long long)r * ly.cols + k], srow[r - r0], t);
            tp[k] = t;
        }
    }
}

__global__ __launch_bounds__(256) void sn_finish_kernel(const mtd_sn_layer* __restrict__ L, int n_layers, SnWs ws, int train) {
    __shared__ float red[256];
    const int layer = blockIdx.x;
    if (layer >= n_layers) return;
    const mtd_sn_layer ly = L[layer];
    const float* wv = ws.wv + row_offset(L, layer);
    float sigma;
    if (train) {
        float p = 0.f;
        for (int r = threadIdx.x; r < ly.rows; r += 256) p += wv[r] * wv[r];
        const float n2 = block_sum(p, red);
        const float inv = 1.f / fmaxf(sqrtf(n2), SN_EPS);
        float q = 0.f;
        for (int r = threadIdx.x; r < ly.rows; r += 256) {
            const float u = wv[r] * inv;
            ly.u[r] = u;
            if (ly.u_save) ly.u_save[r] = u;
            q += u * wv[r];
        }
        sigma = block_sum(q, red);
    } else {
        float q = 0.f;
        for (int r = threadIdx.x; r < ly.rows; r += 256) {
            const float u = ly.u[r];
            if (ly.u_save) ly.u_save[r] = u;
            q += u * wv[r];
        }
        sigma = block_sum(q, red);
    }
    if (threadIdx.x == 0) {
        ly.sigma[0] = sigma;
        ly.sigma[1] = 1.f / sigma;
    }
}

// ---- backward -----------------------------------------------------------------------------------
// HBM-bound: <G, W> reads 2 floats per weight, the correction reads 2 and writes 1.  16 Ki weights per block keep the
// (layer, block) lookup off the critical path; 16-byte accesses whenever the layer's pointers and row length allow.
struct SnGradWs { float* partial; float* dot; };
constexpr int GRAD_ELEMS_PER_BLOCK = 16384;

__device__ __forceinline__ int grad_blocks(const mtd_sn_grad_layer& l) {
    return (int)(((long long)l.rows * l.cols + GRAD_ELEMS_PER_BLOCK - 1) / GRAD_ELEMS_PER_BLOCK);
}
__device__ __forceinline__ bool grad_vec_ok(const mtd_sn_grad_layer& l) {
    return (l.cols & 3) == 0 && aligned16_dev(l.G) && aligned16_dev(l.w) && aligned16_dev(l.g_out) && aligned16_dev(l.G2);
}
__device__ __forceinline__ float dot4(const float4& g, const float4& w, float p) {
    p = fmaf(g.x, w.x, p); p = fmaf(g.y, w.y, p); p = fmaf(g.z, w.z, p); return fmaf(g.w, w.w, p);
}

// partial[2 * block + s] = this block's share of <G_s, W>  (s = 1 only for a paired layer)
__global__ __launch_bounds__(256) void sn_grad_dot_kernel(const mtd_sn_grad_layer* __restrict__ L, int n_layers, SnGradWs ws) {
    __shared__ float red[256];
    int layer, local;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return grad_blocks(L[l]); }, layer, local)) return;
    const mtd_sn_grad_layer ly = L[layer];
    const unsigned total = (unsigned)ly.rows * (unsigned)ly.cols;
    const unsigned base = (unsigned)local * GRAD_ELEMS_PER_BLOCK;
    const bool two = ly.sigma2 != nullptr;              // a second pass (with G2, or folded into a prescaled G)
    float p = 0.f, p2 = 0.f;
    if (ly.act_gy) {
        // activation-side form (mtd_sn_grad_layer.act_*): sigma_s * sum over the pass's pixels of gy (y - b), y from the saved activation.
        // The layer keeps its grad_blocks() blocks (the sum kernel's layout); act_M * rows <= rows * cols elements are spread over them in
        // the same fixed order, blocks past the end contribute zeros.
        const unsigned N = (unsigned)ly.rows, atotal = (unsigned)ly.act_M * N;
        const float islope = ly.act_inv_slope;
#pragma unroll 2
        for (int i = threadIdx.x * 4; i < GRAD_ELEMS_PER_BLOCK; i += 1024) {
            const unsigned e = base + i;
            if (e < atotal) {
                const unsigned pix = e / N, n = e - pix * N;                 // N % 4 == 0: the four elements share a pixel
                float4 g = *reinterpret_cast<const float4*>(ly.act_gy + (size_t)pix * ly.act_gy_ld + n);
                if (ly.act_gy2) {
                    const float4 h = *reinterpret_cast<const float4*>(ly.act_gy2 + (size_t)pix * ly.act_gy2_ld + n);
                    g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
                }
                const float4 av = *reinterpret_cast<const float4*>(ly.act_a + (size_t)pix * ly.act_a_ld + n);
                const float4 b = *reinterpret_cast<const float4*>(ly.act_bias + n);
                float t = 0.f;
                t = fmaf(g.x, (av.x > 0.f ? av.x : av.x * islope) - b.x, t);
                t = fmaf(g.y, (av.y > 0.f ? av.y : av.y * islope) - b.y, t);
                t = fmaf(g.z, (av.z > 0.f ? av.z : av.z * islope) - b.z, t);
                t = fmaf(g.w, (av.w > 0.f ? av.w : av.w * islope) - b.w, t);
                if ((int)pix < ly.act_M_first) p += t;
                else p2 += t;
            }
        }
        p *= ly.sigma[0];
        if (two) p2 *= ly.sigma2[0];
    } else if (grad_vec_ok(ly)) {
        // fixed association: lane-local sums over its float4s in index order, then the block tree
#pragma unroll 4
        for (int i = threadIdx.x * 4; i < GRAD_ELEMS_PER_BLOCK; i += 1024) {
            const unsigned e = base + i;
            if (e < total) {
                const float4 w = *reinterpret_cast<const float4*>(ly.w + e);
                p = dot4(*reinterpret_cast<const float4*>(ly.G + e), w, p);
                if (ly.G2) p2 = dot4(*reinterpret_cast<const float4*>(ly.G2 + e), w, p2);
            }
        }
    } else {
        for (int i = threadIdx.x; i < GRAD_ELEMS_PER_BLOCK; i += 256) {
            const unsigned e = base + i;
            if (e < total) {
                const float w = ly.w[e];
                p = fmaf(ly.G[e], w, p);
                if (ly.G2) p2 = fmaf(ly.G2[e], w, p2);
            }
        }
    }
    const float s = block_sum(p, red);
    if (threadIdx.x == 0) ws.partial[2 * (long long)blockIdx.x] = s;      // blockIdx.x - local = first block of this layer
    if (two) {
        const float s2 = block_sum(p2, red);
        if (threadIdx.x == 0) ws.partial[2 * (long long)blockIdx.x + 1] = s2;
    }
}

// dot[2 * layer + s] = sum of the layer's partials, one block per (layer, s)
__global__ __launch_bounds__(256) void sn_grad_sum_kernel(const mtd_sn_grad_layer* __restrict__ L, int n_layers, SnGradWs ws) {
    __shared__ float red[256];
    const int layer = blockIdx.x >> 1, s = blockIdx.x & 1;
    if (layer >= n_layers || (s && !L[layer].sigma2)) return;
    const int nb = grad_blocks(L[layer]);
    long long first = 0;
    for (int l = 0; l < layer; ++l) first += grad_blocks(L[l]);
    const float* part = ws.partial + 2 * first + s;
    float p = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) p += part[2 * i];
    const float t = block_sum(p, red);
    if (threadIdx.x == 0) ws.dot[2 * layer + s] = t;
}

__global__ __launch_bounds__(256) void sn_grad_apply_kernel(const mtd_sn_grad_layer* __restrict__ L, int n_layers, SnGradWs ws) {
    int layer, local;
    if (!locate(n_layers, blockIdx.x, [&](int l) { return grad_blocks(L[l]); }, layer, local)) return;
    const mtd_sn_grad_layer ly = L[layer];
    const unsigned total = (unsigned)ly.rows * (unsigned)ly.cols;
    const unsigned base = (unsigned)local * GRAD_ELEMS_PER_BLOCK;
    const unsigned cols = (unsigned)ly.cols;
    const bool two = ly.sigma2 != nullptr;              // two passes: a second rank-one term, and a second gradient unless G is prescaled
    const bool two_g = ly.G2 != nullptr;
    const float inv = ly.sigma[1];
    const float coef = ws.dot[2 * layer] * inv * inv;
    const float inv2 = two ? ly.sigma2[1] : 0.f;
    const float coef2 = two ? ws.dot[2 * layer + 1] * inv2 * inv2 : 0.f;
    const float ginv = ly.prescaled ? 1.f : inv;        // (prescaled: G = G_1 / sigma_1 + G_2 / sigma_2 already)
    if (grad_vec_ok(ly)) {
#pragma unroll 4
        for (int i = threadIdx.x * 4; i < GRAD_ELEMS_PER_BLOCK; i += 1024) {
            const unsigned e = base + i;
            if (e < total) {
                const unsigned r = e / cols, k = e - r * cols;          // cols % 4 == 0: the four weights share a row
                const float cu = coef * ly.u[r];
                const float4 G = *reinterpret_cast<const float4*>(ly.G + e);
                float4 o;
                o.x = G.x * ginv - cu * ly.v[k];
                o.y = G.y * ginv - cu * ly.v[k + 1];
                o.z = G.z * ginv - cu * ly.v[k + 2];
                o.w = G.w * ginv - cu * ly.v[k + 3];
                float4* dst = reinterpret_cast<float4*>(ly.g_out + e);
                if (ly.accumulate) {
                    const float4 a = *dst;
                    o.x = a.x + o.x; o.y = a.y + o.y; o.z = a.z + o.z; o.w = a.w + o.w;
                }
                if (two_g) {                                             // second pass added after the first, as two launches would
                    const float cu2 = coef2 * ly.u2[r];
                    const float4 H = *reinterpret_cast<const float4*>(ly.G2 + e);
                    o.x += H.x * inv2 - cu2 * ly.v2[k];
                    o.y += H.y * inv2 - cu2 * ly.v2[k + 1];
                    o.z += H.z * inv2 - cu2 * ly.v2[k + 2];
                    o.w += H.w * inv2 - cu2 * ly.v2[k + 3];
                } else if (two) {                                        // prescaled: the second pass's rank-one term only
                    const float cu2 = coef2 * ly.u2[r];
                    o.x -= cu2 * ly.v2[k];
                    o.y -= cu2 * ly.v2[k + 1];
                    o.z -= cu2 * ly.v2[k + 2];
                    o.w -= cu2 * ly.v2[k + 3];
                }
                *dst = o;
            }
        }
    } else {
        for (int i = threadIdx.x; i < GRAD_ELEMS_PER_BLOCK; i += 256) {
            const unsigned e = base + i;
            if (e < total) {
                const unsigned r = e / cols, k = e - r * cols;
                float g = ly.G[e] * ginv - coef * ly.u[r] * ly.v[k];
                if (ly.accumulate) g = ly.g_out[e] + g;
                if (two_g) g += ly.G2[e] * inv2 - coef2 * ly.u2[r] * ly.v2[k];
                else if (two) g -= coef2 * ly.u2[r] * ly.v2[k];
                ly.g_out[e] = g;
            }
        }
    }
}

size_t sn_ws_floats(const mtd_sn_layer* h, int n, long long* cols_total, long long* rows_total) {
    long long c = 0, r = 0;
    for (int i = 0; i < n; ++i) { c += h[i].cols; r += h[i].rows; }
    if (cols_total) *cols_total = c;
    if (rows_total) *rows_total = r;
    return (size_t)(c + (long long)n * MAX_CHUNKS + r + c * MAX_ROW_CHUNKS);
}

}  // namespace

extern "C" size_t mtd_sn_ws_bytes(const mtd_sn_layer* layers_host, int n_layers) {
    if (!layers_host || n_layers <= 0) return 0;
    return sn_ws_floats(layers_host, n_layers, nullptr, nullptr) * sizeof(float);
}

extern "C" int mtd_sn_power_iter(const mtd_sn_layer* layers_dev, const mtd_sn_layer* layers_host, int n_layers, int train, float* ws,
                                 void* stream) {
    if (!layers_dev || !layers_host || n_layers <= 0 || !ws) return MTD_EINVAL;
    long long ctot = 0, rtot = 0;
    sn_ws_floats(layers_host, n_layers, &ctot, &rtot);
    int col_blocks = 0, row_blocks = 0, wtu_blocks = 0;
    for (int i = 0; i < n_layers; ++i) {
        if (layers_host[i].cols > COLS_PER_BLOCK * MAX_CHUNKS || layers_host[i].rows <= 0 || layers_host[i].cols <= 0) return MTD_EINVAL;
        if (layers_host[i].rows > ROWS_PER_BLOCK * MAX_ROW_CHUNKS) return MTD_EINVAL;
        col_blocks += (layers_host[i].cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
        wtu_blocks += ((layers_host[i].cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK) * ((layers_host[i].rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        row_blocks += (layers_host[i].rows + 3) / 4;
    }
    SnWs w;
    w.t = ws;
    w.partial = ws + ctot;
    w.wv = w.partial + (long long)n_layers * MAX_CHUNKS;
    w.tp = w.wv + rtot;
    hipStream_t s = (hipStream_t)stream;
    if (train) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(wtu_blocks), dim3(256), 0, s, layers_dev, n_layers, w);
        MTD_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_tsum_kernel, dim3(col_blocks), dim3(256), 0, s, layers_dev, n_layers, w, ROWS_PER_BLOCK, 0);
        MTD_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_norm_v_kernel, dim3(col_blocks), dim3(256), 0, s, layers_dev, n_layers, w);
        MTD_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(sn_copy_v_kernel, dim3(col_blocks), dim3(256), 0, s, layers_dev, n_layers);
        MTD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3(row_blocks), dim3(256), 0, s, layers_dev, n_layers, w);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_finish_kernel, dim3(n_layers), dim3(256), 0, s, layers_dev, n_layers, w, train);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

// nit power iterations on the same weights, back to back (the discriminator step's four passes, train mode): iteration i reads and
// writes the layers' u, v in place like mtd_sn_power_iter and leaves its sigma / u_save / v_save where entry [i * n_layers + l] of the
// tables points (entries of one layer share w, u, v, rows, cols).  nit + 1 passes over the weights instead of 2 nit (sn_wv_wtu_kernel).
// Same ws as mtd_sn_power_iter.  The results differ from nit calls of mtd_sn_power_iter by rounding only (t = (W^T s) / |s| instead of
// W^T (s / |s|), and the row dot products of the fused pass in another fixed association).
extern "C" int mtd_sn_power_iter_multi(const mtd_sn_layer* layers_dev, const mtd_sn_layer* layers_host, int n_layers, int nit, float* ws,
                                       void* stream) {
    if (!layers_dev || !layers_host || n_layers <= 0 || nit <= 0 || !ws) return MTD_EINVAL;
    long long ctot = 0, rtot = 0;
    sn_ws_floats(layers_host, n_layers, &ctot, &rtot);
    int col_blocks = 0, row_blocks = 0, wtu_blocks = 0, fuse_blocks = 0;
    for (int i = 0; i < n_layers; ++i) {
        const mtd_sn_layer& h = layers_host[i];
        if (h.cols > COLS_PER_BLOCK * MAX_CHUNKS || h.rows <= 0 || h.cols <= 0) return MTD_EINVAL;
        if (h.rows > FUSE_ROWS * MAX_ROW_CHUNKS) return MTD_EINVAL;                 // (the fused pass writes one partial per FUSE_ROWS rows)
        for (int it = 1; it < nit; ++it) {
            const mtd_sn_layer& o = layers_host[(long long)it * n_layers + i];
            if (o.w != h.w || o.u != h.u || o.v != h.v || o.rows != h.rows || o.cols != h.cols) return MTD_EINVAL;
        }
        col_blocks += (h.cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
        wtu_blocks += ((h.cols + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK) * ((h.rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
        row_blocks += (h.rows + 3) / 4;
        fuse_blocks += (h.rows + FUSE_ROWS - 1) / FUSE_ROWS;
    }
    SnWs w;
    w.t = ws;
    w.partial = ws + ctot;
    w.wv = w.partial + (long long)n_layers * MAX_CHUNKS;
    w.tp = w.wv + rtot;
    hipStream_t s = (hipStream_t)stream;
    for (int it = 0; it < nit; ++it) {
        const mtd_sn_layer* tab = layers_dev + (long long)it * n_layers;
        if (it == 0) {
            hipLaunchKernelGGL(sn_wtu_kernel, dim3(wtu_blocks), dim3(256), 0, s, tab, n_layers, w);
            MTD_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(sn_tsum_kernel, dim3(col_blocks), dim3(256), 0, s, tab, n_layers, w, it == 0 ? ROWS_PER_BLOCK : FUSE_ROWS, it == 0 ? 0 : 1);
        MTD_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_norm_v_kernel, dim3(col_blocks), dim3(256), 0, s, tab, n_layers, w);
        MTD_LAUNCH_CHECK();
        if (it + 1 < nit) hipLaunchKernelGGL(sn_wv_wtu_kernel, dim3(fuse_blocks), dim3(FUSE_NT), 0, s, tab, n_layers, w);
        else hipLaunchKernelGGL(sn_wv_kernel, dim3(row_blocks), dim3(256), 0, s, tab, n_layers, w);
        MTD_LAUNCH_CHECK();
        hipLaunchKernelGGL(sn_finish_kernel, dim3(n_layers), dim3(256), 0, s, tab, n_layers, w, 1);
        MTD_LAUNCH_CHECK();
    }
    return MTD_OK;
}

static long long sn_grad_blocks_host(const mtd_sn_grad_layer* h, int n) {
    long long b = 0;
    for (int i = 0; i < n; ++i) b += ((long long)h[i].rows * h[i].cols + GRAD_ELEMS_PER_BLOCK - 1) / GRAD_ELEMS_PER_BLOCK;
    return b;
}

extern "C" size_t mtd_sn_grad_ws_bytes(const mtd_sn_grad_layer* layers_host, int n_layers) {
    if (!layers_host || n_layers <= 0) return 0;
    return (size_t)(2 * (sn_grad_blocks_host(layers_host, n_layers) + n_layers)) * sizeof(float);
}

extern "C" int mtd_sn_grad(const mtd_sn_grad_layer* layers_dev, const mtd_sn_grad_layer* layers_host, int n_layers, float* ws,
                           void* stream) {
    if (!layers_dev || !layers_host || n_layers <= 0 || !ws) return MTD_EINVAL;
    for (int i = 0; i < n_layers; ++i) {
        const mtd_sn_grad_layer& l = layers_host[i];
        if (l.prescaled && (!l.act_gy || l.G2)) return MTD_EINVAL;      // the dot products of a prescaled gradient come from the activation side
        if (l.G2 && !(l.u2 && l.v2 && l.sigma2)) return MTD_EINVAL;
        if (!l.act_gy) continue;
        // the activation-side dot: its elements fit the layer's blocks, float4 accesses everywhere, a second pass only with a second sigma
        if (!l.act_a || !l.act_bias || l.act_M <= 0 || l.act_M_first < 0 || l.act_M_first > l.act_M) return MTD_EINVAL;
        if ((long long)l.act_M * l.rows > (long long)l.rows * l.cols || (long long)l.act_M * l.rows >= (1ll << 31)) return MTD_EINVAL;
        if ((l.rows % 4) || (l.act_gy_ld % 4) || (l.act_a_ld % 4) || (l.act_gy2 && (l.act_gy2_ld % 4))) return MTD_EALIGN;
        if (!aligned16(l.act_gy) || !aligned16(l.act_a) || !aligned16(l.act_bias) || (l.act_gy2 && !aligned16(l.act_gy2))) return MTD_EALIGN;
        if (l.act_M_first < l.act_M && !l.sigma2) return MTD_EINVAL;
    }
    const long long nb = sn_grad_blocks_host(layers_host, n_layers);
    SnGradWs w;
    w.partial = ws;
    w.dot = ws + 2 * nb;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_grad_dot_kernel, dim3((unsigned)nb), dim3(256), 0, s, layers_dev, n_layers, w);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_grad_sum_kernel, dim3(2 * n_layers), dim3(256), 0, s, layers_dev, n_layers, w);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(sn_grad_apply_kernel, dim3((unsigned)nb), dim3(256), 0, s, layers_dev, n_layers, w);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
