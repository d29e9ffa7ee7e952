// pcie_duplex_probe.cpp -- do the two directions of the PCIe link overlap?  SDMA copies in both directions do not on
// this box (tools/pcie_probe.py: 56 GB/s aggregate); here D2H is done by a copy KERNEL storing to pinned (GPU-mapped)
// host memory while the SDMA engine does H2D.   hipcc --offload-arch=gfx950 -O2 tools/pcie_duplex_probe.cpp -o tools/pcie_duplex_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main()
{
    const size_t N = 1ull << 30;
    void *h_src, *h_dst, *d_in, *d_out;
    CK(hipHostMalloc(&h_src, N, hipHostMallocDefault)); CK(hipHostMalloc(&h_dst, N, hipHostMallocDefault));
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_out, N));
    CK(hipMemset(d_out, 1, N));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto now = [] { return std::chrono::high_resolution_clock::now(); };
    for (int blocks : {8, 32, 128, 512}) {
        for (int mode = 0; mode < 4; mode++) {     // 0 kernel D2H alone, 1 kernel D2H + SDMA H2D, 2 kernel H2D (reads host) alone, 3 kernel H2D + SDMA D2H
            for (int rep = 0; rep < 2; rep++) {
                CK(hipDeviceSynchronize());
                auto t0 = now();
                const int R = 4;
                for (int i = 0; i < R; i++) {
                    if (mode == 0 || mode == 1) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s2, (const uint4 *)d_out, (uint4 *)h_dst, N / 16);
                    if (mode == 1) CK(hipMemcpyAsync(d_in, h_src, N, hipMemcpyHostToDevice, s1));
                    if (mode == 2 || mode == 3) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s1, (const uint4 *)h_src, (uint4 *)d_in, N / 16);
                    if (mode == 3) CK(hipMemcpyAsync(h_dst, d_out, N, hipMemcpyDeviceToHost, s2));
                }
                CK(hipDeviceSynchronize());
                const double dt = std::chrono::duration<double>(now() - t0).count() / R;
                const double bytes = (mode == 1 || mode == 3) ? 2.0 * N : 1.0 * N;
                if (rep == 1) printf("blocks %4d mode %d: %7.1f GB/s %s (%.1f ms)\n", blocks, mode, bytes / dt / 1e9,
                                     mode == 0 ? "kernel D2H alone" : mode == 1 ? "kernel D2H + SDMA H2D (aggregate)" : mode == 2 ? "kernel H2D alone" : "kernel H2D + SDMA D2H (aggregate)", dt * 1e3);
            }
        }
    }
    return 0;
}
