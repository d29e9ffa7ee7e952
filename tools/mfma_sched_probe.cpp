// tools/mfma_sched_probe.cpp -- diagnostic (not part of the product): how a lone wavefront should ORDER FP64 MFMAs.
// Cycles per v_mfma_f64_16x16x4_f64 for accumulator chains of different lengths interleaved in different ways, and for a
// dependent chain with VALU transitions with and without independent MFMAs placed at the transitions.
// hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o tools/mfma_sched_probe tools/mfma_sched_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void __launch_bounds__(64) probe(double *out, unsigned long long *stamps, int iters, double seed)
{
    const int lane = threadIdx.x;
    double a = seed + lane * 1e-3, b = seed * 0.5 - lane * 1e-3;
    d4 A = {0, 0, 0, 0}, B = A, C = A, D = A;
    double f = a, g = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {          // AAAA AAAA: one chain
            A = MFMA(a, b, A); A = MFMA(b, a, A); A = MFMA(a, b, A); A = MFMA(b, a, A);
            A = MFMA(a, b, A); A = MFMA(b, a, A); A = MFMA(a, b, A); A = MFMA(b, a, A);
        } else if (MODE == 1) {   // AAAA BBBB: two chains, switch every four
            A = MFMA(a, b, A); A = MFMA(b, a, A); A = MFMA(a, b, A); A = MFMA(b, a, A); SB();
            B = MFMA(a, b, B); B = MFMA(b, a, B); B = MFMA(a, b, B); B = MFMA(b, a, B); SB();
        } else if (MODE == 2) {   // ABAB ABAB: two chains alternating
            A = MFMA(a, b, A); SB(); B = MFMA(a, b, B); SB(); A = MFMA(b, a, A); SB(); B = MFMA(b, a, B); SB();
            A = MFMA(a, b, A); SB(); B = MFMA(a, b, B); SB(); A = MFMA(b, a, A); SB(); B = MFMA(b, a, B); SB();
        } else if (MODE == 3) {   // AABB AABB
            A = MFMA(a, b, A); A = MFMA(b, a, A); SB(); B = MFMA(a, b, B); B = MFMA(b, a, B); SB();
            A = MFMA(a, b, A); A = MFMA(b, a, A); SB(); B = MFMA(a, b, B); B = MFMA(b, a, B); SB();
        } else if (MODE == 4) {   // ABCD ABCD: four chains alternating
            A = MFMA(a, b, A); SB(); B = MFMA(a, b, B); SB(); C = MFMA(a, b, C); SB(); D = MFMA(a, b, D); SB();
            A = MFMA(b, a, A); SB(); B = MFMA(b, a, B); SB(); C = MFMA(b, a, C); SB(); D = MFMA(b, a, D); SB();
        } else if (MODE == 5) {   // dependent pairs with a VALU transition: (AA -> valu -> BB -> valu) x2 ; 8 MFMAs
            A = MFMA(f, b, A); A = MFMA(b, f, A); SB(); g = A.x * 0.5 + 1.0; SB();
            B = MFMA(g, b, B); B = MFMA(b, g, B); SB(); f = B.x * 0.5 + 1.0; SB();
            A = MFMA(f, b, A); A = MFMA(b, f, A); SB(); g = A.x * 0.5 + 1.0; SB();
            B = MFMA(g, b, B); B = MFMA(b, g, B); SB(); f = B.x * 0.5 + 1.0; SB();
        } else if (MODE == 6) {   // the same with one independent MFMA (chain C) issued at every transition; 12 MFMAs
            A = MFMA(f, b, A); A = MFMA(b, f, A); SB(); C = MFMA(a, b, C); SB(); g = A.x * 0.5 + 1.0; SB();
            B = MFMA(g, b, B); B = MFMA(b, g, B); SB(); C = MFMA(a, b, C); SB(); f = B.x * 0.5 + 1.0; SB();
            A = MFMA(f, b, A); A = MFMA(b, f, A); SB(); C = MFMA(a, b, C); SB(); g = A.x * 0.5 + 1.0; SB();
            B = MFMA(g, b, B); B = MFMA(b, g, B); SB(); C = MFMA(a, b, C); SB(); f = B.x * 0.5 + 1.0; SB();
        } else if (MODE == 7) {   // operand chains of four: AAAA -> (B uses A.x) BBBB -> ...; 8 MFMAs
            A = MFMA(f, b, A); A = MFMA(b, f, A); A = MFMA(f, b, A); A = MFMA(b, f, A); SB();
            B = MFMA(A.x, b, B); B = MFMA(b, A.y, B); B = MFMA(A.z, b, B); B = MFMA(b, A.w, B); SB();
            f = B.x;
        } else if (MODE == 8) {   // the same with an independent chain C interleaved one-for-one; 16 MFMAs
            A = MFMA(f, b, A); SB(); C = MFMA(a, b, C); SB(); A = MFMA(b, f, A); SB(); C = MFMA(a, b, C); SB();
            A = MFMA(f, b, A); SB(); C = MFMA(a, b, C); SB(); A = MFMA(b, f, A); SB(); C = MFMA(a, b, C); SB();
            B = MFMA(A.x, b, B); SB(); C = MFMA(a, b, C); SB(); B = MFMA(b, A.y, B); SB(); C = MFMA(a, b, C); SB();
            B = MFMA(A.z, b, B); SB(); C = MFMA(a, b, C); SB(); B = MFMA(b, A.w, B); SB(); C = MFMA(a, b, C); SB();
            f = B.x;
        } else if (MODE == 9) {   // 8 chained MFMAs with 16 independent v_fma_f64 placed two after each MFMA
            double h0 = f, h1 = g;
#pragma unroll
            for (int k = 0; k < 8; k++) { A = MFMA(a, b, A); SB(); h0 = __builtin_fma(h0, a, b); h1 = __builtin_fma(h1, a, b); SB(); }
            f = h0; g = h1;
        } else if (MODE >= 11 && MODE <= 19) {  // 8 chained MFMAs + 4 independent instructions of one kind after each
            __shared__ double lds[64 * 8];
            unsigned u0 = (unsigned)lane + i, u1 = u0 * 3;
            double h0 = f, h1 = g;
            const double *p = out + 4096 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                A = MFMA(a, b, A); SB();
                if (MODE == 11) { asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %0" : "+v"(u0), "+v"(u1)); }
                if (MODE == 12) { asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1" ::: "s20", "s21", "scc"); }
                if (MODE == 13) { asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n v_mov_b64 %0, %1\n v_mov_b64 %1, %0" : "+v"(h0), "+v"(h1)); }
                if (MODE == 14) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(u0), "+v"(u1) :: "vcc"); }
                if (MODE == 15) { lds[lane + 64 * (k & 3)] = h0; lds[lane + 64 * (4 + (k & 3))] = h1; asm volatile("" ::: "memory"); lds[lane + 64 * ((k + 1) & 3)] = h1; lds[lane + 64 * (4 + ((k + 2) & 3))] = h0; asm volatile("" ::: "memory"); }
                if (MODE == 16) { asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7"); }
                if (MODE == 17) {   // 4 stores, never waited for
                    double *qp = out + 4096 * 64 + 64 * 64 + lane;
                    __builtin_nontemporal_store(h0, qp + 64 * (k & 7)); __builtin_nontemporal_store(h1, qp + 64 * (8 + (k & 7)));
                    __builtin_nontemporal_store(h0, qp + 64 * (16 + (k & 7))); __builtin_nontemporal_store(h1, qp + 64 * (24 + (k & 7)));
                }
                if (MODE == 18) { asm volatile("v_fma_f64 %0, %0, %0, %1\n v_fma_f64 %1, %1, %1, %0\n v_fma_f64 %0, %0, %0, %1\n v_fma_f64 %1, %1, %1, %0" : "+v"(h0), "+v"(h1)); }
                if (MODE == 19) { asm volatile("v_mul_f64 %0, %0, %0\n v_add_f64 %1, %1, %0\n v_mul_f64 %0, %0, %0\n v_add_f64 %1, %1, %0" : "+v"(h0), "+v"(h1)); }
                SB();
            }
            f = h0 + (double)u0; g = h1 + (double)u1 + lds[lane];
            (void)p;
        } else if (MODE == 20) {  // 8 chained MFMAs + 4 loads issued after each, consumed only after the 8th MFMA of the NEXT iteration
            const double *p = out + 4096 * 64 + lane;
            double l[4];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                A = MFMA(a, b, A); SB();
                if (k == 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) l[j] = __builtin_nontemporal_load(p + 64 * ((i * 4 + j) & 1023));
                }
                SB();
            }
            f += l[0] + l[1] + l[2] + l[3];
        } else if (MODE == 10) {  // 8 chained MFMAs with 4 independent global loads placed between
            const double *p = out + 4096 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 8; k++) { A = MFMA(a, b, A); SB(); if (k & 1) f += __builtin_nontemporal_load(p + 64 * ((i * 8 + k) & 1023)); SB(); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = A.x + B.y + C.z + D.w + f + g;
    if (lane == 0) stamps[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int per_iter, double *out, unsigned long long *st)
{
    const int iters = 20000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, out, st, iters, 0.37); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < blocks; i++) cyc += h[i];
    cyc /= blocks;
    printf("%-72s blocks=%5d : %7.2f cycles per MFMA, %8.1f per iteration\n", name, blocks, cyc / ((double)iters * per_iter), cyc / iters);
}

int main()
{
    double *out; unsigned long long *st;
    hipMalloc(&out, (4096 * 64 + 1024 * 64 + 64) * 8); hipMemset(out, 0, (4096 * 64 + 1024 * 64 + 64) * 8); hipMalloc(&st, 4096 * 8);
    for (int blocks : {1024}) {
        run<0>("AAAAAAAA one accumulator chain", blocks, 8, out, st);
        run<1>("AAAA BBBB two chains, switch every 4", blocks, 8, out, st);
        run<3>("AABB AABB two chains, switch every 2", blocks, 8, out, st);
        run<2>("ABAB ABAB two chains alternating", blocks, 8, out, st);
        run<4>("ABCD ABCD four chains alternating", blocks, 8, out, st);
        run<5>("AA valu BB valu (each pair needs the VALU result of the one before)", blocks, 8, out, st);
        run<6>("the same + 1 independent MFMA at every transition (12 per iter)", blocks, 12, out, st);
        run<7>("AAAA -> BBBB operand-dependent", blocks, 8, out, st);
        run<8>("the same + independent chain interleaved 1:1 (16 per iter)", blocks, 16, out, st);
        run<9>("8 chained MFMAs + 2 independent v_fma_f64 after each", blocks, 8, out, st);
        run<11>("8 chained MFMAs + 4 v_add_u32 after each", blocks, 8, out, st);
        run<14>("8 chained MFMAs + 4 v_cndmask_b32 after each", blocks, 8, out, st);
        run<13>("8 chained MFMAs + 4 v_mov_b64 after each", blocks, 8, out, st);
        run<18>("8 chained MFMAs + 4 dependent v_fma_f64 after each", blocks, 8, out, st);
        run<19>("8 chained MFMAs + 2 (v_mul_f64, v_add_f64) after each", blocks, 8, out, st);
        run<12>("8 chained MFMAs + 4 s_add_u32 after each", blocks, 8, out, st);
        run<16>("8 chained MFMAs + 4 s_nop 7 after each", blocks, 8, out, st);
        run<15>("8 chained MFMAs + 4 ds_write_b64 after each", blocks, 8, out, st);
        run<17>("8 chained MFMAs + 4 global stores after each", blocks, 8, out, st);
        run<20>("8 chained MFMAs + 4 global loads per 8 (consumed at the end)", blocks, 8, out, st);
    }
    return 0;
}
