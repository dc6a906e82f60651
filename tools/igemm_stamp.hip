// Diagnostic build of the product implicit-GEMM kernel with in-kernel s_memtime stamps (MTD_STAMPS): prints where
// one wave of the first 64 workgroups spends its cycles.  Never linked into libmtdgan_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/igemm_stamp.hip -o tools/igemm_stamp
#define MTD_STAMPS 1
#include "../mtd-gan_amd/csrc/conv_igemm.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

static mtd_geom geom_fwd(int B, int H, int W, int k, int s, int p) {
    int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    mtd_geom g = {B, H, W, OH, OW, s, s, -p, -p, 1, 1, k, k, k, 0, 0, 1, 1, OH, OW, 1, 1, 0, 0};
    return g;
}
int mtd_prof_begin(int, int, int, long long, int, int, int, hipStream_t) { return -1; }
void mtd_prof_end(int, hipStream_t) {}
int main(int argc, char** argv) {
    // igemm_stamp [B H C N cfg]   (default: the generator's 32 -> 32 layer at batch 32 under config 1)
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 64, W = H;
    const int C = argc > 3 ? atoi(argv[3]) : 32, N = argc > 4 ? atoi(argv[4]) : 32;
    mtd_conv_igemm_override(argc > 5 ? atoi(argv[5]) : 1, 1);
    float *in, *w, *out, *bias;
    (void)hipMalloc(&in, (size_t)B * H * W * C * 4);
    (void)hipMalloc(&out, (size_t)B * H * W * N * 4);
    (void)hipMalloc(&w, (size_t)9 * N * C * 4);
    (void)hipMalloc(&bias, N * 4);
    (void)hipMemset(in, 0, (size_t)B * H * W * C * 4);
    (void)hipMemset(w, 0, (size_t)9 * N * C * 4);
    (void)hipMemset(bias, 0, N * 4);
    unsigned long long* sb;
    (void)hipMalloc(&sb, 64 * 64 * 8);
    (void)hipMemset(sb, 0, 64 * 64 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mtd_stamp_buf), &sb, sizeof(sb));
    mtd_conv_args a = {};
    a.g = geom_fwd(B, H, W, 3, 1, 1);
    a.in = in; a.in_ld = C; a.C = C;
    a.w = w; a.w_sn = C; a.w_sc = 1; a.w_st = (long long)N * C;   // packed [tap][n][c]
    a.N = N; a.out = out; a.out_ld = N; a.bias = bias; a.act = MTD_ACT_RELU;
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
    printf("stamped kernel: %.1f us per launch\n", ms * 1e3 / 20);
    std::vector<unsigned long long> h(64 * 64);
    (void)hipMemcpy(h.data(), sb, 64 * 64 * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"start->prologue done", "first loads issued", "first B stored+barrier"};
    for (int wg : {0, 1, 17, 40}) {
        unsigned long long* s = &h[wg * 64];
        printf("WG %d: prologue %llu, first-load-issue %llu, first-store+barrier %llu\n", wg, s[1] - s[0], s[2] - s[1], s[3] - s[2]);
        for (int it = 0; it < 9; ++it) {
            unsigned long long* c = &s[4 + 4 * it];
            printf("   chunk %d: top(LDS read+advance+issue loads) %6llu | mfma %6llu | store_b+barrier %6llu | (next top - end) \n", it, c[1] - c[0], c[2] - c[1], c[3] - c[2]);
        }
        printf("   loop total %llu, epilogue %llu, whole kernel %llu cycles (memtime ticks)\n", s[60] - s[3], s[61] - s[60], s[61] - s[0]);
    }
    (void)names;
    return 0;
}
