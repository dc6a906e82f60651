// Diagnostic build of the halo-tile generator kernel with in-kernel s_memtime stamps (MTD_STAMPS): where one wave of the
// first 64 workgroups spends its cycles.  Never linked into libmtdgan_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/c32t_stamp.hip -o tools/c32t_stamp
#define MTD_STAMPS 1
#include "../mtd-gan_amd/csrc/conv_igemm.hip"
#include <cstdio>
#include <vector>
int mtd_prof_begin(int, int, int, long long, int, int, int, hipStream_t) { return -1; }
void mtd_prof_end(int, hipStream_t) {}

int main(int argc, char** argv) {
    const int cfg = argc > 1 ? atoi(argv[1]) : 10;
    const int B = 32, H = 64, W = 64, C = 32, N = 32;
    float *in, *w, *out, *bias, *add;
    (void)hipMalloc(&in, (size_t)B * H * W * C * 4);
    (void)hipMalloc(&out, (size_t)B * H * W * N * 4);
    (void)hipMalloc(&add, (size_t)B * H * W * N * 4);
    (void)hipMalloc(&w, 9 * N * C * 4);
    (void)hipMalloc(&bias, N * 4);
    (void)hipMemset(in, 0, (size_t)B * H * W * C * 4);
    (void)hipMemset(add, 0, (size_t)B * H * W * C * 4);
    (void)hipMemset(w, 0, 9 * N * C * 4);
    (void)hipMemset(bias, 0, N * 4);
    unsigned long long* sb;
    (void)hipMalloc(&sb, 64 * 64 * 8);
    (void)hipMemset(sb, 0, 64 * 64 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mtd_stamp_buf), &sb, sizeof(sb));
    mtd_conv_args a = {};
    mtd_geom g = {B, H, W, H, W, 1, 1, -1, -1, 1, 1, 3, 3, 3, 0, 0, 1, 1, H, W, 1, 1, 0, 0};
    a.g = g;
    a.in = in; a.in_ld = C; a.C = C;
    a.w = w; a.w_sn = C; a.w_sc = 1; a.w_st = (long long)N * C;   // packed [tap][n][c]
    a.N = N; a.out = out; a.out_ld = N; a.bias = bias; a.act = MTD_ACT_RELU;
    a.add1 = add; a.add1_ld = N;
    mtd_conv_igemm_override(cfg, 1);
    for (int r = 0; r < 3; ++r) {
        int rc = mtd_conv_igemm(&a, 0);
        if (rc) { printf("rc=%d\n", rc); return 1; }
    }
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) mtd_conv_igemm(&a, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("cfg %d: %.1f us per launch\n", cfg, ms * 1e3 / 20);
    if (cfg != 10) return 0;
    std::vector<unsigned long long> h(64 * 64);
    (void)hipMemcpy(h.data(), sb, 64 * 64 * 8, hipMemcpyDeviceToHost);
    for (int wg : {0, 1, 17, 40, 63}) {
        unsigned long long* s = &h[wg * 64];
        printf("WG %2d: dma issue %6llu | wait+barrier %6llu | tile0: load+mfma %6llu, wait %6llu, store %6llu | barrier %6llu | tile1: load+mfma %6llu, wait %6llu, store %6llu | total %6llu ticks\n",
               wg, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], 0ull, s[6] - s[5], s[7] - s[6], s[8] - s[7], s[8] - s[0]);
    }
    return 0;
}
