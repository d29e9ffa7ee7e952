// diagnostic: dump V' of the tiled backward kernel for NT=2 vs NT=3 on the same n=30 problem
#define KP_DEBUG_DUMP
#ifndef KP_DEBUG_STEP
#define KP_DEBUG_STEP 1
#endif
#include <cstdio>
#include <vector>
#include <cmath>
#include "../trajoptkp_amd/csrc/tiled_mfma.hip"
using namespace kpilqr;
__device__ double hrand(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
    return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
}
__global__ void fill(RecLayout L, long long nbt, double *rec)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nbt * L.stride; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i % L.stride); const double u = hrand((unsigned long long)i);
        double v = 0.0; const int n = L.n, m = L.m;
        if (e < L.off_B) { int r = e / n, c = e % n; v = (r == c ? 0.97 : 0.0) + 0.01 * u; }
        else if (e < L.off_lxx) v = 0.004 * u;
        else if (e < L.off_lx) { int q = e - L.off_lxx; int r = q / n, c = q % n; v = (r == c ? 0.2 : 0.0) + 0.01 * hrand((unsigned long long)(i - e) + (r < c ? r * n + c : c * n + r) + 77); }
        else if (e < L.off_luu) v = 0.05 * u;
        else if (e < L.off_lu) { int q = e - L.off_luu; int r = q / m, c = q % m; v = (r == c ? 0.02 : 0.0) + 0.001 * hrand((unsigned long long)(i - e) + (r < c ? r * m + c : c * m + r) + 99991); }
        else if (e < L.rec) v = 0.01 * u;
        rec[i] = v;
    }
}
template <int NT> std::vector<double> run(RecLayout L, int T, double *rec, double *lam, double *K, double *k, double *dJ, int *st)
{
    const size_t lds = backward_tiled_lds_bytes(NT);
    hipFuncSetAttribute((const void *)k_backward_tiled<7, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipMemset(dJ, 0, (16 + 9 * 64 * 64) * 8);
    hipLaunchKernelGGL((k_backward_tiled<7, NT>), dim3(1), dim3(64), lds, 0, L, T, rec, lam, 100, K, k, dJ, st);
    hipDeviceSynchronize();
    std::vector<double> h(16 + 9 * 64 * 64);
    hipMemcpy(h.data(), dJ, h.size() * 8, hipMemcpyDeviceToHost);
    return h;
}
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 30, T = 8;
    RecLayout L(n, 7);
    double *rec, *lam, *K, *k, *dJ; int *st;
    hipMalloc(&rec, (size_t)T * L.stride * 8); hipMalloc(&lam, 8); hipMalloc(&K, (size_t)T * n * 7 * 8); hipMalloc(&k, T * 7 * 8);
    hipMalloc(&dJ, (16 + 9 * 64 * 64) * 8); hipMalloc(&st, 4);
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, L, (long long)T, rec);
    double l = 0.1; hipMemcpy(lam, &l, 8, hipMemcpyHostToDevice);
    auto a = run<2>(L, T, rec, lam, K, k, dJ, st);
    auto b = run<3>(L, T, rec, lam, K, k, dJ, st);
    const char *names[9] = {"V' in", "Tz", "Fz", "Qzz", "Quz", "X", "G", "V' acc", "V' sym"};
    for (int s = 0; s < 9; s++) {
        double md = 0; int mi = -1, mj = -1, cnt = 0;
        for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) {
            const double d = fabs(a[16 + s * 4096 + i * 64 + j] - b[16 + s * 4096 + i * 64 + j]);
            if (d > 1e-12) cnt++;
            if (d > md) { md = d; mi = i; mj = j; }
        }
        printf("%-8s: max diff %.3e at (%d,%d), %d elements differ\n", names[s], md, mi, mj, cnt);
        if (md > 1e-12) {
            int pc = 0;
            for (int i = 0; i < 32 && pc < 8; i++) for (int j = 0; j < 32 && pc < 8; j++) {
                const double d = fabs(a[16 + s * 4096 + i * 64 + j] - b[16 + s * 4096 + i * 64 + j]);
                if (d > 1e-12) { printf("   (%d,%d): %.6f vs %.6f\n", i, j, a[16 + s * 4096 + i * 64 + j], b[16 + s * 4096 + i * 64 + j]); pc++; }
            }
        }
    }
    return 0;
}
