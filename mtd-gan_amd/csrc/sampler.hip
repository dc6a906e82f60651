// Training-patch front end on the device (reference: create_datasets/Mayo.py:117-136, type "window_patch"; the transforms
// themselves live in monai 1.3.2, absent here -- restated in oracle/data_oracle.py, parity unpinned).
// HBM-bound byte/short work: a slice pair is 1 MiB of int16, a batch of 8 patches writes 256 KiB; one thread per output
// pixel gathers through the whole transform chain (crop box, symmetric zero pad, sample origin, quarter turns, flip,
// small-angle bilinear rotation), so no intermediate image is ever written.
#include "common.h"

namespace {

__device__ __forceinline__ float window(short hu, float a_min, float a_max) {
    float v = ((float)hu - a_min) / (a_max - a_min);       // ScaleIntensityRange: (x - a_min) / (a_max - a_min) * 1 + 0, then clip
    return fminf(fmaxf(v, 0.f), 1.f);
}

// [y0, y1, x0, x1) of HU > a_min per slice; initialised to (H, 0, W, 0) by bbox_init, empty boxes widened by bbox_fix
__global__ __launch_bounds__(256) void bbox_init_kernel(int* bbox, int n, int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { bbox[4 * i] = H; bbox[4 * i + 1] = 0; bbox[4 * i + 2] = W; bbox[4 * i + 3] = 0; }
}
__global__ __launch_bounds__(256) void bbox_kernel(const short* __restrict__ hu, int H, int W, float a_min, int* bbox) {
    const int s = blockIdx.y;
    const short* img = hu + (long long)s * H * W;
    int y0 = H, y1 = 0, x0 = W, x1 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
        if ((float)img[i] > a_min) {
            const int y = i / W, x = i - y * W;
            y0 = min(y0, y); y1 = max(y1, y + 1); x0 = min(x0, x); x1 = max(x1, x + 1);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        y0 = min(y0, __shfl_xor(y0, off, 64)); y1 = max(y1, __shfl_xor(y1, off, 64));
        x0 = min(x0, __shfl_xor(x0, off, 64)); x1 = max(x1, __shfl_xor(x1, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && y1 > 0) {      // min / max are order-independent: atomics keep the result exact
        atomicMin(&bbox[4 * s], y0); atomicMax(&bbox[4 * s + 1], y1);
        atomicMin(&bbox[4 * s + 2], x0); atomicMax(&bbox[4 * s + 3], x1);
    }
}
__global__ __launch_bounds__(256) void bbox_fix_kernel(int* bbox, int n, int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && bbox[4 * i + 1] == 0) { bbox[4 * i] = 0; bbox[4 * i + 1] = H; bbox[4 * i + 2] = 0; bbox[4 * i + 3] = W; }
}

struct SampleParams {
    const short* lo; const short* hi;
    int H, W, n, roi;
    const int* bbox; const mtd_patch_desc* descs;
    float a_min, a_max;
    float* out_lo; float* out_hi;
};

// value of the (cropped, padded) image of one dose level at integer position (y, x) of the roi x roi sample BEFORE the
// quarter turns and the flip are undone -- i.e. (y, x) index the patch as RandRotated sees it
__device__ __forceinline__ void patch_value(const SampleParams& p, const short* lo, const short* hi, int y, int x, int k, int flip,
                                            int oy, int ox, int pady, int padx, int y0, int y1, int x0, int x1, float& vlo, float& vhi) {
    const int R = p.roi;
    if (flip) { y = R - 1 - y; x = R - 1 - x; }          // np.flip over both axes
    for (int t = 0; t < k; ++t) {                        // np.rot90 once: out[i][j] = in[j][R - 1 - i]
        const int ny = x, nx = R - 1 - y;
        y = ny; x = nx;
    }
    const int cy = oy + y - pady + y0, cx = ox + x - padx + x0;      // sample origin, symmetric pad, crop box
    if (cy >= y0 && cy < y1 && cx >= x0 && cx < x1) {
        vlo = window(lo[cy * p.W + cx], p.a_min, p.a_max);
        vhi = window(hi[cy * p.W + cx], p.a_min, p.a_max);
    } else {
        vlo = 0.f; vhi = 0.f;                             // SpatialPad: constant 0 (after the window)
    }
}

__global__ __launch_bounds__(256) void sample_kernel(const SampleParams p) {
    const int R = p.roi;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (idx >= R * R) return;
    const mtd_patch_desc d = p.descs[i];
    const int y0 = p.bbox[4 * d.slice], y1 = p.bbox[4 * d.slice + 1], x0 = p.bbox[4 * d.slice + 2], x1 = p.bbox[4 * d.slice + 3];
    const int ch = y1 - y0, cw = x1 - x0;
    const int ph = max(ch, R), pw = max(cw, R);                       // SpatialPad to at least roi, symmetric (extra row after)
    const int pady = (ph - ch) / 2, padx = (pw - cw) / 2;
    const int ry = ph - R + 1, rx = pw - R + 1;                       // RandSpatialCrop: origin in [0, size - roi]
    const int oy = min((int)(d.uy * (float)ry), ry - 1), ox = min((int)(d.ux * (float)rx), rx - 1);
    const short* lo = p.lo + (long long)d.slice * p.H * p.W;
    const short* hi = p.hi + (long long)d.slice * p.H * p.W;
    const int oyp = idx / R, oxp = idx - oyp * R;
    const int k = d.rot_k & 3;
    float vlo, vhi;
    if (d.angle == 0.f) {
        patch_value(p, lo, hi, oyp, oxp, k, d.flip, oy, ox, pady, padx, y0, y1, x0, x1, vlo, vhi);
    } else {
        // src = c + R(angle) (dst - c), c = (roi - 1) / 2; bilinear, coordinates clamped to the patch (border padding)
        const float c = 0.5f * (float)(R - 1);
        const float cs = cosf(d.angle), sn = sinf(d.angle);
        const float dy = (float)oyp - c, dx = (float)oxp - c;
        float sy = c + cs * dy - sn * dx, sx = c + sn * dy + cs * dx;
        sy = fminf(fmaxf(sy, 0.f), (float)(R - 1));
        sx = fminf(fmaxf(sx, 0.f), (float)(R - 1));
        const int iy0 = (int)floorf(sy), ix0 = (int)floorf(sx);
        const int iy1 = min(iy0 + 1, R - 1), ix1 = min(ix0 + 1, R - 1);
        const float fy = sy - (float)iy0, fx = sx - (float)ix0;
        float a[4], b[4];
        patch_value(p, lo, hi, iy0, ix0, k, d.flip, oy, ox, pady, padx, y0, y1, x0, x1, a[0], b[0]);
        patch_value(p, lo, hi, iy0, ix1, k, d.flip, oy, ox, pady, padx, y0, y1, x0, x1, a[1], b[1]);
        patch_value(p, lo, hi, iy1, ix0, k, d.flip, oy, ox, pady, padx, y0, y1, x0, x1, a[2], b[2]);
        patch_value(p, lo, hi, iy1, ix1, k, d.flip, oy, ox, pady, padx, y0, y1, x0, x1, a[3], b[3]);
        const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
        vlo = ((a[0] * w00 + a[1] * w01) + a[2] * w10) + a[3] * w11;
        vhi = ((b[0] * w00 + b[1] * w01) + b[2] * w10) + b[3] * w11;
    }
    p.out_lo[(long long)i * R * R + idx] = vlo;
    p.out_hi[(long long)i * R * R + idx] = vhi;
}

__global__ __launch_bounds__(256) void hu_window_kernel(const short* __restrict__ hu, long long n, float a_min, float a_max, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = window(hu[i], a_min, a_max);
}

}  // namespace

extern "C" int mtd_foreground_bbox(const short* hu_full, int n_slices, int H, int W, float a_min, int* bbox, void* stream) {
    if (!hu_full || !bbox || n_slices <= 0 || H <= 0 || W <= 0 || (long long)H * W >= (1ll << 31)) return MTD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init_kernel, dim3((n_slices + 255) / 256), dim3(256), 0, s, bbox, n_slices, H, W);
    MTD_LAUNCH_CHECK();
    int bx = (H * W + 256 * 16 - 1) / (256 * 16);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(bbox_kernel, dim3(bx, n_slices), dim3(256), 0, s, hu_full, H, W, a_min, bbox);
    MTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(bbox_fix_kernel, dim3((n_slices + 255) / 256), dim3(256), 0, s, bbox, n_slices, H, W);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_window_patches(const short* hu_low, const short* hu_full, int n_slices, int H, int W, const int* bbox,
                                  const mtd_patch_desc* descs, int n, float a_min, float a_max, int roi, float* out_low, float* out_full,
                                  void* stream) {
    if (!hu_low || !hu_full || !bbox || !descs || !out_low || !out_full) return MTD_EINVAL;
    if (n_slices <= 0 || H <= 0 || W <= 0 || n <= 0 || roi <= 0 || !(a_max > a_min) || (long long)H * W >= (1ll << 31)) return MTD_EINVAL;
    SampleParams p;
    p.lo = hu_low; p.hi = hu_full; p.H = H; p.W = W; p.n = n; p.roi = roi;
    p.bbox = bbox; p.descs = descs; p.a_min = a_min; p.a_max = a_max; p.out_lo = out_low; p.out_hi = out_full;
    hipLaunchKernelGGL(sample_kernel, dim3((roi * roi + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, p);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}

extern "C" int mtd_hu_window(const short* hu, long long n, float a_min, float a_max, float* out, void* stream) {
    if (!hu || !out || n <= 0 || !(a_max > a_min)) return MTD_EINVAL;
    long long want = (n + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(hu_window_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, hu, n, a_min, a_max, out);
    MTD_LAUNCH_CHECK();
    return MTD_OK;
}
