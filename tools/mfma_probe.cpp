// tools/mfma_probe.cpp -- diagnostic micro-benchmarks (not part of the product): cycles per
// v_mfma_f64_16x16x4_f64 (independent / dependent chains), per v_fma_f64, and the shader clock held
// while every SIMD runs them.  hipcc -O3 --offload-arch=gfx950 -o tools/mfma_probe tools/mfma_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(64) probe(double *out, unsigned long long *stamps, int iters, double seed)
{
    const int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = seed * 0.5 - lane * 1e-3;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f0 = a, f1 = b, f2 = a + 1, f3 = b + 1, f4 = a + 2, f5 = b + 2, f6 = a + 3, f7 = b + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // 4 independent accumulators, 8 MFMAs / iter
            c0 = MFMA(a, b, c0); c1 = MFMA(a, b, c1); c2 = MFMA(a, b, c2); c3 = MFMA(a, b, c3);
            c0 = MFMA(b, a, c0); c1 = MFMA(b, a, c1); c2 = MFMA(b, a, c2); c3 = MFMA(b, a, c3);
        } else if (MODE == 1) {   // one accumulator chain (C dependence), 8 MFMAs / iter
            c0 = MFMA(a, b, c0); c0 = MFMA(b, a, c0); c0 = MFMA(a, b, c0); c0 = MFMA(b, a, c0);
            c0 = MFMA(a, b, c0); c0 = MFMA(b, a, c0); c0 = MFMA(a, b, c0); c0 = MFMA(b, a, c0);
        } else if (MODE == 2) {   // result feeds the next MFMA's B operand, 8 MFMAs / iter
            c0 = MFMA(a, c0.x, c0); c0 = MFMA(a, c0.y, c0); c0 = MFMA(a, c0.z, c0); c0 = MFMA(a, c0.w, c0);
            c0 = MFMA(a, c0.x, c0); c0 = MFMA(a, c0.y, c0); c0 = MFMA(a, c0.z, c0); c0 = MFMA(a, c0.w, c0);
        } else if (MODE == 3) {   // 8 independent v_fma_f64 / iter
            f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
        } else if (MODE == 4) {   // 8 dependent v_fma_f64 / iter
            f0 = __builtin_fma(f0, a, b); f0 = __builtin_fma(f0, a, b); f0 = __builtin_fma(f0, a, b); f0 = __builtin_fma(f0, a, b);
            f0 = __builtin_fma(f0, a, b); f0 = __builtin_fma(f0, a, b); f0 = __builtin_fma(f0, a, b); f0 = __builtin_fma(f0, a, b);
        } else if (MODE == 5) {   // 4 MFMA + 8 independent fma interleaved
            c0 = MFMA(a, b, c0); f0 = __builtin_fma(f0, a, b); f1 = __builtin_fma(f1, a, b);
            c1 = MFMA(a, b, c1); f2 = __builtin_fma(f2, a, b); f3 = __builtin_fma(f3, a, b);
            c2 = MFMA(a, b, c2); f4 = __builtin_fma(f4, a, b); f5 = __builtin_fma(f5, a, b);
            c3 = MFMA(a, b, c3); f6 = __builtin_fma(f6, a, b); f7 = __builtin_fma(f7, a, b);
        } else if (MODE == 7) {   // 8 independent v_mfma_f64_4x4x4_4b (four 4x4x4 blocks each; one double per lane in, one out)
            f0 = MFMA4(a, b, f0); f1 = MFMA4(a, b, f1); f2 = MFMA4(a, b, f2); f3 = MFMA4(a, b, f3);
            f4 = MFMA4(b, a, f4); f5 = MFMA4(b, a, f5); f6 = MFMA4(b, a, f6); f7 = MFMA4(b, a, f7);
        } else if (MODE == 8) {   // 8 dependent 4x4x4 (accumulator chain)
            f0 = MFMA4(a, b, f0); f0 = MFMA4(b, a, f0); f0 = MFMA4(a, b, f0); f0 = MFMA4(b, a, f0);
            f0 = MFMA4(a, b, f0); f0 = MFMA4(b, a, f0); f0 = MFMA4(a, b, f0); f0 = MFMA4(b, a, f0);
        } else if (MODE == 9) {   // 8 dependent 4x4x4, result feeds the next B operand
            f0 = MFMA4(a, f0, f1); f0 = MFMA4(a, f0, f1); f0 = MFMA4(a, f0, f1); f0 = MFMA4(a, f0, f1);
            f0 = MFMA4(a, f0, f1); f0 = MFMA4(a, f0, f1); f0 = MFMA4(a, f0, f1); f0 = MFMA4(a, f0, f1);
        } else if (MODE == 6) {   // 8 dependent rcp-free divisions emulation: v_rcp_f64 chain
            f0 = __builtin_amdgcn_rcp(f0 + 1.5); f0 = __builtin_amdgcn_rcp(f0 + 1.5); f0 = __builtin_amdgcn_rcp(f0 + 1.5); f0 = __builtin_amdgcn_rcp(f0 + 1.5);
            f0 = __builtin_amdgcn_rcp(f0 + 1.5); f0 = __builtin_amdgcn_rcp(f0 + 1.5); f0 = __builtin_amdgcn_rcp(f0 + 1.5); f0 = __builtin_amdgcn_rcp(f0 + 1.5);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + lane] = c0.x + c1.y + c2.z + c3.w + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (lane == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char *name, int blocks, int per_iter, double *out, unsigned long long *st)
{
    const int iters = 20000;
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, iters, 0.37);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, iters, 0.37);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= blocks; rt /= blocks;
    printf("%-34s blocks=%5d : %7.2f cycles/op  clock %.3f GHz  (kernel %.3f ms, %.1f ns/op)\n", name, blocks,
           cyc / ((double)iters * per_iter), cyc / rt * 0.1, ms, ms * 1e6 / ((double)iters * per_iter));
}

int main()
{
    double *out; unsigned long long *st;
    hipMalloc(&out, 4096 * 64 * 8); hipMalloc(&st, 4096 * 16);
    for (int blocks : {1, 1024, 2048}) {
        run<0>("mfma_f64_16x16x4 indep x4", blocks, 8, out, st);
        run<1>("mfma_f64_16x16x4 acc chain", blocks, 8, out, st);
        run<2>("mfma_f64_16x16x4 D->B operand chain", blocks, 8, out, st);
        run<3>("v_fma_f64 independent", blocks, 8, out, st);
        run<4>("v_fma_f64 dependent", blocks, 8, out, st);
        run<5>("4 mfma + 8 fma interleaved (per 12)", blocks, 12, out, st);
        run<6>("v_rcp_f64+add dependent (per pair)", blocks, 8, out, st);
        run<7>("mfma_f64_4x4x4_4b independent", blocks, 8, out, st);
        run<8>("mfma_f64_4x4x4_4b acc chain", blocks, 8, out, st);
        run<9>("mfma_f64_4x4x4_4b D->B operand chain", blocks, 8, out, st);
    }
    return 0;
}
