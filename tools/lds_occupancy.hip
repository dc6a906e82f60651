#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ __launch_bounds__(256, 2) void k(float* out, long long spin) {
    __shared__ float L[KB * 256];
    L[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = L[1];
}
template <int KB> void run(const char* name) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<KB>, 256, 0);
    float* out; hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 768, 1024}) {
        hipLaunchKernelGGL(k<KB>, dim3(blocks), dim3(256), 0, 0, out, 200000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KB>, dim3(blocks), dim3(256), 0, 0, out, 200000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s LDS %d KB: occupancy API %d blocks/CU; %d blocks -> %.1f us\n", name, KB, nb, blocks, ms * 1e3);
    }
}
int main() { run<8>("k8"); run<32>("k32"); run<41>("k41"); run<52>("k52"); run<64>("k64"); return 0; }
