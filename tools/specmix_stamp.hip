// Diagnostic build of the Res-FFT mix kernel (forward) with in-kernel s_memtime stamps: where one wave spends its cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/specmix_stamp.hip -o tools/specmix_stamp
#define MTD_STAMPS 1
#include "../mtd-gan_amd/csrc/resfft.hip"
#include <cstdio>
#include <vector>
int main() {
    const int B = 32;
    const size_t n = (size_t)B * 33 * 64 * 64;
    float *R, *T, *S, *Z, *w, *b2;
    (void)hipMalloc(&R, n * 4); (void)hipMalloc(&T, n * 4); (void)hipMalloc(&S, n * 4); (void)hipMalloc(&Z, n * 4);
    (void)hipMalloc(&w, 4096 * 4); (void)hipMalloc(&b2, 64 * 4);
    (void)hipMemset(R, 0, n * 4); (void)hipMemset(w, 0, 4096 * 4); (void)hipMemset(b2, 0, 256);
    unsigned long long* sb;
    (void)hipMalloc(&sb, 64 * 16 * 8);
    (void)hipMemset(sb, 0, 64 * 16 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(rf_stamp_buf), &sb, sizeof(sb));
    for (int r = 0; r < 3; ++r) mtd_spec_mix_fwd(R, w, b2, T, S, Z, B, 0);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) mtd_spec_mix_fwd(R, w, b2, T, S, Z, B, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("spec_mix_fwd: %.1f us per launch\n", ms * 1e3 / 20);
    std::vector<unsigned long long> h(64 * 16);
    (void)hipMemcpy(h.data(), sb, 64 * 16 * 8, hipMemcpyDeviceToHost);
    for (int wg : {0, 9, 27, 63}) {
        unsigned long long* s = &h[wg * 16];
        printf("WG %2d: setup %5llu | loads %6llu | fft %6llu | lds+S_save %6llu | weights %6llu | mix (256 mfma + Z_save) %6llu | lds read %6llu | ifft %6llu | stores %6llu | total %6llu\n",
               wg, 0ull, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5], s[7] - s[6], s[8] - s[7], s[8] - s[0]);
    }
    return 0;
}
