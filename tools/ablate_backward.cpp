// tools/ablate_backward.cpp -- diagnostic only (not part of libkpilqr.so): times the MFMA backward
// kernel with parts switched off, to see what bounds a step (HBM latency, the LDL' solve on the VALU,
// the K stores, the symmetrisation).  Build: hipcc -O3 --offload-arch=gfx950 -o ablate tools/ablate_backward.cpp
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../trajoptkp_amd/csrc/riccati_mfma.hip"

using namespace kpilqr;

__device__ double hrand(unsigned long long x)
{   // splitmix64 -> uniform (-1,1) with a full random mantissa
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
    return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
}
__global__ void fill_records_dense(RecLayout L, long long nbt, double *rec)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nbt * L.stride; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % L.stride);
        const double u = hrand((unsigned long long)i);
        double v = 0.0;
        const int n = L.n, m = L.m;
        if (e < L.off_B) { int r = e / n, c = e % n; v = (r == c ? 0.97 : 0.0) + 0.01 * u; }
        else if (e < L.off_lxx) { v = 0.004 * u; }
        else if (e < L.off_lx) { int q = e - L.off_lxx; int r = q / n, c = q % n; v = (r == c ? 0.2 : 0.0) + 0.01 * hrand((unsigned long long)(i - e) + (r < c ? r * n + c : c * n + r) + 77); }
        else if (e < L.off_luu) v = 0.05 * u;
        else if (e < L.off_lu) { int q = e - L.off_luu; int r = q / m, c = q % m; v = (r == c ? 0.02 : 0.0) + 0.001 * hrand((unsigned long long)(i - e) + (r < c ? r * m + c : c * m + r) + 99991); }
        else if (e < L.rec) v = 0.01 * u;
        rec[i] = v;
    }
}
__global__ void fill_records(RecLayout L, long long nbt, double *rec)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nbt * L.stride; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % L.stride);
        const long long bt = i / L.stride;
        const double jitter = 1e-3 * (double)((bt * 2654435761u + e * 40503u) & 1023) / 1024.0;
        double v = 0.0;
        const int n = L.n, m = L.m;
        if (e < L.off_B) { int r = e / n, c = e % n; v = (r == c ? 0.98 : 0.0) + (c == r + n / 2 ? 0.008 : 0.0) + 0.002 * jitter; }
        else if (e < L.off_lxx) { int q = e - L.off_B; int r = q / m, c = q % m; v = (r == c + n / 2 ? 0.004 : 0.0) + 1e-4 * jitter; }
        else if (e < L.off_lx) { int q = e - L.off_lxx; int r = q / n, c = q % n; v = (r == c ? 0.2 : 0.0); }
        else if (e < L.off_luu) v = 0.01 + jitter;
        else if (e < L.off_lu) { int q = e - L.off_luu; int r = q / m, c = q % m; v = (r == c ? 0.01 : 0.0); }
        else if (e < L.rec) v = 0.001 + jitter;
        rec[i] = v;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ABL>
static float run(RecLayout L, int B, int T, double *rec, double *lam, double *K, double *k, double *dJ, int *st, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_backward_mfma_excl<14, 7, ABL>), dim3(B), dim3(64), 0, 0, L, T, rec, lam, 100, K, k, dJ, st);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL((k_backward_mfma_excl<14, 7, ABL>), dim3(B), dim3(64), 0, 0, L, T, rec, lam, 100, K, k, dJ, st);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int s0; double d0; CK(hipMemcpy(&s0, st, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&d0, dJ, 8, hipMemcpyDeviceToHost));
    printf("ABL=%2d : %8.3f ms/launch  (%.0f cycles/step @2.4GHz)  status0=%d dJ0=%g\n", ABL, ms / reps,
           ms / reps * 1e-3 * 2.4e9 / T, s0, d0);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 1024, T = argc > 2 ? atoi(argv[2]) : 3000;
    RecLayout L(14, 7);
    double *rec, *lam, *K, *k, *dJ; int *st;
    const size_t nbt = (size_t)B * T;
    CK(hipMalloc(&rec, nbt * L.stride * 8)); CK(hipMalloc(&lam, B * 8)); CK(hipMalloc(&K, nbt * 98 * 8));
    CK(hipMalloc(&k, nbt * 7 * 8)); CK(hipMalloc(&dJ, B * 8)); CK(hipMalloc(&st, B * 4));
    hipLaunchKernelGGL(fill_records, dim3(4096), dim3(256), 0, 0, L, (long long)nbt, rec);
    std::vector<double> hl(B, 0.1); CK(hipMemcpy(lam, hl.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    if (argc > 3 && atoi(argv[3]) == 1) {
        hipLaunchKernelGGL(fill_records_dense, dim3(4096), dim3(256), 0, 0, L, (long long)nbt, rec);
        CK(hipDeviceSynchronize());
        printf("dense random records\n");
    }
    printf("B=%d T=%d  (backward MFMA kernel ablations)\n", B, T);
    run<0>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<1>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<2>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<4>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<8>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<1 | 2>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<1 | 2 | 4 | 8>(L, B, T, rec, lam, K, k, dJ, st, 3);
    run<0>(L, B, T, rec, lam, K, k, dJ, st, 3);
    return 0;
}
