// mfma_common.h -- device helpers shared by the FP64 MFMA kernels (riccati_mfma.hip, forward_mfma.hip,
// tiled_mfma.hip, fused_mfma.hip): the tile type and primitive, bounds-checked buffer loads, a ~1 ulp reciprocal,
// and the slow path of the backward pass (Eigen's pivoted LDLT + explicit inverse).
#pragma once
#include "common.h"

namespace kpilqr {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// D = A*B + C on the FP64 matrix core, 16x16 output, 4 rows of the contraction per instruction.
#define KP_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
// byte offset that lies outside every buffer descriptor used here: the load returns 0, the store is dropped
#define KP_OOB 0x7ffffff0

__device__ __forceinline__ double kp_bld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t kp_rsrc(const void *p, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}
// 1/x to ~1 ulp: v_rcp_f64 seed + two Newton steps (the pivots are O(lambda)..O(1): no scaling needed; a
// non-positive or non-finite pivot is caught by the callers' `pos` test and takes the slow path).
__device__ __forceinline__ double kp_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// acc + Y'X over NC 4-row chunks: the one primitive of these kernels (tiles in the accumulator layout are their own
// transpose as the A operand and the next B operand as they are).
// One-tile shapes (n + 1 <= 16) that have their own backward / fused kernel instantiations: Panda reaching and push_soft
// (14,7), Acrobot and piston (4,1), hopper and floating cube (12,3), pentabot (10,3) -- the joint / actuator counts of the
// reference's TaskConfigs.  Everything else goes to the tiled kernels (any n + 2 <= 64, num_ctrl <= 8).
#define KP_T1_SHAPES(X) X(14, 7) X(4, 1) X(12, 3) X(10, 3)
static inline bool kp_t1_shape(int n, int m)
{
#define KP_X(NN, MM) if (n == NN && m == MM) return true;
    KP_T1_SHAPES(KP_X)
#undef KP_X
    return false;
}

template <int NC>
__device__ __forceinline__ d4 kp_P(const d4 &Y, const d4 &X, d4 acc)
{
    acc = KP_MFMA(Y.x, X.x, acc);
    if (NC > 1) acc = KP_MFMA(Y.y, X.y, acc);
    if (NC > 2) acc = KP_MFMA(Y.z, X.z, acc);
    if (NC > 3) acc = KP_MFMA(Y.w, X.w, acc);
    return acc;
}

// Running inverse of Q = Quu + lambda I (m x m, rows/cols < 4*NCU of a tile): Newton-Schulz steps
// Xinv <- Xinv + Xinv (I - Q Xinv) on the matrix core, 4 MFMAs each, quadratic convergence.  The number of steps is
// chosen from the measured residual so that the last one ends below 1e-15 (bound on the row sums of |I - Q Xinv|:
// m * max entry).  Returns false, leaving Xinv untouched, when the residual is too large to converge in four
// steps (the caller then factorises).  Q and Xinv are symmetric up to rounding, so Q'X = QX and X'R = XR.
template <int NCU>
__device__ __forceinline__ bool kp_inverse_refresh(const d4 &Qr, const d4 &Iu, d4 &Xinv, int m)
{
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 R = Iu - kp_P<NCU>(Qr, Xinv, zero);
    double rmax = fabs(R.x);
    if (NCU > 1) rmax = fmax(rmax, fabs(R.y));
    if (NCU > 2) rmax = fmax(rmax, fabs(R.z));
    if (NCU > 3) rmax = fmax(rmax, fabs(R.w));
    const double e = (double)m * rmax;
    if (__builtin_amdgcn_ballot_w64(!(e < 0.11)) != 0) return false;
    const int iters = (__builtin_amdgcn_ballot_w64(e >= 1.3e-2) != 0) ? 4
                    : (__builtin_amdgcn_ballot_w64(e >= 1.7e-4) != 0) ? 3
                    : (__builtin_amdgcn_ballot_w64(e >= 3.0e-8) != 0) ? 2 : 1;
    Xinv = kp_P<NCU>(Xinv, R, Xinv);
    if (iters > 1) {
        R = Iu - kp_P<NCU>(Qr, Xinv, zero); Xinv = kp_P<NCU>(Xinv, R, Xinv);
        if (iters > 2) {
            R = Iu - kp_P<NCU>(Qr, Xinv, zero); Xinv = kp_P<NCU>(Xinv, R, Xinv);
            if (iters > 3) { R = Iu - kp_P<NCU>(Qr, Xinv, zero); Xinv = kp_P<NCU>(Xinv, R, Xinv); }
        }
    }
    return true;
}

// The same with (1) a first guess extrapolated from the last two inverses, 2 Xinv - Xprev: between key-points the
// inverse moves smoothly from step to step, so the residual of the extrapolation is the SECOND difference (Panda bench
// workload: median 8e-7 against 4e-5 for the plain previous inverse; at a key-point, where the slopes of A and B change,
// it falls back to first order), and (2) a third-order step issued BEFORE the residual is looked at:
//     Y = X + X R,  X <- X + Y R   ( = X (I + R + R^2), error e^3; Y is symmetric like X: X R = X - X Q X )
// 6 MFMAs in a straight line and one branch when e < 2e-5 (e^3 < 1e-14); beyond that, second-order steps continue
// from the third-order result (residual e^3).  Shorter dependent chain than two second-order steps (3 transitions
// against 4, one residual test against four, 34.4 MFMAs per step against 36): backward sweep of one trajectory
// 3.30 -> 3.11 ms (consumer wave of the triple, where the chain is the critical path), B = 1024 one-wave 5.28 -> 5.19 ms.
template <int NCU>
__device__ __forceinline__ bool kp_inverse_refresh_p(const d4 &Qr, const d4 &Iu, d4 &Xinv, d4 &Xprev, int m)
{
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 X0 = Xinv;
    X0.x = __builtin_fma(2.0, Xinv.x, -Xprev.x);
    if (NCU > 1) X0.y = __builtin_fma(2.0, Xinv.y, -Xprev.y);
    if (NCU > 2) X0.z = __builtin_fma(2.0, Xinv.z, -Xprev.z);
    if (NCU > 3) X0.w = __builtin_fma(2.0, Xinv.w, -Xprev.w);
    d4 R = Iu - kp_P<NCU>(Qr, X0, zero);
    d4 Y = kp_P<NCU>(X0, R, X0);
    Y = kp_P<NCU>(Y, R, X0);
    double rmax = fabs(R.x);
    if (NCU > 1) rmax = fmax(rmax, fabs(R.y));
    if (NCU > 2) rmax = fmax(rmax, fabs(R.z));
    if (NCU > 3) rmax = fmax(rmax, fabs(R.w));
    const double e = (double)m * rmax;
    if (__builtin_amdgcn_ballot_w64(!(e < 2.0e-5)) != 0) {
        if (__builtin_amdgcn_ballot_w64(!(e < 0.11)) != 0) return false;
        const int iters = (__builtin_amdgcn_ballot_w64(e >= 1.3e-2) != 0) ? 3 : (__builtin_amdgcn_ballot_w64(e >= 1.7e-4) != 0) ? 2 : 1;
        R = Iu - kp_P<NCU>(Qr, Y, zero); Y = kp_P<NCU>(Y, R, Y);
        if (iters > 1) {
            R = Iu - kp_P<NCU>(Qr, Y, zero); Y = kp_P<NCU>(Y, R, Y);
            if (iters > 2) { R = Iu - kp_P<NCU>(Qr, Y, zero); Y = kp_P<NCU>(Y, R, Y); }
        }
    }
    Xprev = Xinv;
    Xinv = Y;
    return true;
}

// The same on the NEGATED inverse N = -(Quu + lambda I)^-1 (what the fused backward sweep carries): the residual
// R = I - Q X = I + Q N is the MFMA accumulator started at I -- no VALU between the products of the chain -- and the gains
// K = -X = N Quz come out with their sign.  Everything else as kp_inverse_refresh_p (N (I + R + R^2), extrapolated start).
// steps (optional): the number of second-order steps that followed the third-order one (0..3; -1 when it gave up).
// KINK (the step right below a key-point, known statically in the segment-loop forms: the peeled step): the slopes of A and B
// change there, the extrapolated first guess is off by a FIRST difference again (residual median ~4e-5 instead of ~8e-7) and the
// third-order step alone is not enough on most such steps (round 3: 18 % of all steps ran a second-order step behind it: 4 MFMAs,
// a residual test and a chain transition).  One more term of the series instead -- N0 (I + R + R^2 + R^3), error e^4, two more
// MFMAs in the same straight line -- is enough up to e < 1.7e-4 (e^4 < 1e-15).
#ifndef KP_SERIES4
#define KP_SERIES4 1                // 0: a second-order step wherever the third-order series does not reach 1e-15 (round 4 before the last day; A/B builds)
#endif
// SER4 (the consumer waves of the wave pairs / triple): see the fourth term below
template <int NCU, bool KINK = false, bool SER4 = false>
__device__ __forceinline__ bool kp_inverse_refresh_n(const d4 &Qr, const d4 &Iu, d4 &Ninv, d4 &Nprev, int m, int *steps = nullptr)
{
    if (steps) *steps = 0;
    d4 N0 = Ninv;
    N0.x = __builtin_fma(2.0, Ninv.x, -Nprev.x);
    if (NCU > 1) N0.y = __builtin_fma(2.0, Ninv.y, -Nprev.y);
    if (NCU > 2) N0.z = __builtin_fma(2.0, Ninv.z, -Nprev.z);
    if (NCU > 3) N0.w = __builtin_fma(2.0, Ninv.w, -Nprev.w);
    d4 R = kp_P<NCU>(Qr, N0, Iu);                     // I + Q N0
    d4 Y = kp_P<NCU>(N0, R, N0);
    Y = kp_P<NCU>(Y, R, N0);
    if constexpr (KINK) Y = kp_P<NCU>(Y, R, N0);      // N0 (I + R + R^2 + R^3)
    double rmax = fabs(R.x);
    if (NCU > 1) rmax = fmax(rmax, fabs(R.y));
    if (NCU > 2) rmax = fmax(rmax, fabs(R.z));
    if (NCU > 3) rmax = fmax(rmax, fabs(R.w));
    const double e = (double)m * rmax;
    if (__builtin_amdgcn_ballot_w64(!(e < (KINK ? 1.7e-4 : 2.0e-5))) != 0) {
        if (__builtin_amdgcn_ballot_w64(!(e < 0.11)) != 0) { if (steps) *steps = -1; return false; }
        if (!KINK && SER4 && KP_SERIES4 && __builtin_amdgcn_ballot_w64(e >= 1.7e-4) == 0) {
            // 2e-5 <= e < 1.7e-4 in a consumer wave, which does not know at compile time that it is on the step below a key-point:
            // the fourth term of the series, two products, instead of a second-order step, four -- the measured residual says it is
            // enough (e^4 < 1e-15).  2.92 against 2.96 ms at 512 trajectories.  (In the one-wave sweeps -- the general form meets
            // such steps at every DoF's key-points -- it measured -0.03 ... +0.10 ms: left as it was there.)
            if (steps) *steps = 1;
            Y = kp_P<NCU>(Y, R, N0);
        } else {
        // second-order steps behind the series: its residual is e^3 (e^4), squared by every step, to end below 1e-15
        const int iters = KINK ? ((__builtin_amdgcn_ballot_w64(e >= 1.3e-2) != 0) ? 2 : 1)
                               : (__builtin_amdgcn_ballot_w64(e >= 1.3e-2) != 0) ? 3 : (__builtin_amdgcn_ballot_w64(e >= 1.7e-4) != 0) ? 2 : 1;
        if (steps) *steps = iters;
        R = kp_P<NCU>(Qr, Y, Iu); Y = kp_P<NCU>(Y, R, Y);
        if (iters > 1) {
            R = kp_P<NCU>(Qr, Y, Iu); Y = kp_P<NCU>(Y, R, Y);
            if (iters > 2) { R = kp_P<NCU>(Qr, Y, Iu); Y = kp_P<NCU>(Y, R, Y); }
        }
        }
    }
    Nprev = Ninv;
    Ninv = Y;
    return true;
}

// Unpivoted LDL' of an m x m SPD matrix, done redundantly by every lane from a broadcast image (`qel(i,j)` returns
// element (i,j)): L (unit lower, strictly lower part stored) and the reciprocals of D.  Returns false when a pivot is
// not positive -- the callers then report the PD failure (checked steps) or take the pivoted slow path.
template <int M, class QEl>
__device__ __forceinline__ bool kp_ldl_factor(QEl qel, double (&Lm)[M][M], double (&rd)[M])
{
    double dd[M];
    bool pos = true;
#pragma unroll
    for (int j = 0; j < M; j++) {
        double w[M];
        double dj = qel(j, j);
#pragma unroll
        for (int kk = 0; kk < j; kk++) { w[kk] = Lm[j][kk] * dd[kk]; dj -= Lm[j][kk] * w[kk]; }
        dd[j] = dj;
        pos = pos && (dj > 0.0);
        const double rj = kp_rcp(dj);
        rd[j] = rj;
#pragma unroll
        for (int i = j + 1; i < M; i++) {
            double v = qel(i, j);
#pragma unroll
            for (int kk = 0; kk < j; kk++) v -= Lm[i][kk] * w[kk];
            Lm[i][j] = v * rj;
        }
    }
    return pos;
}
// x <- (L D L')^-1 x
template <int M>
__device__ __forceinline__ void kp_ldl_solve(const double (&Lm)[M][M], const double (&rd)[M], double (&x)[M])
{
#pragma unroll
    for (int j = 0; j < M; j++) {            // L y = z
#pragma unroll
        for (int i = j + 1; i < M; i++) x[i] -= Lm[i][j] * x[j];
    }
#pragma unroll
    for (int i = 0; i < M; i++) x[i] *= rd[i];   // D
#pragma unroll
    for (int j = M - 1; j >= 0; j--) {       // L' x = y
#pragma unroll
        for (int i = 0; i < j; i++) x[i] -= Lm[j][i] * x[j];
    }
}

// Eigen's LDLT (symmetric pivoting, in place on the lower triangle) followed by solve(Identity), restated line
// for line as the reference uses it when Q_uu + lambda I is not PD on an unchecked step
// (src/Optimiser/iLQR.cpp:597-604); identical to generic.hip's dev_ldlt_inverse and oracle/orc_ldlt_inverse.
// M: m x m row-major with row stride ms; a, x: m*m work/result (column-major); temp: m; tr: m ints.
// Run by ONE lane; noinline with a run-time m so that the rare path costs no registers in the hot loop.
__device__ static __attribute__((noinline)) void kp_slow_ldlt_inverse(int m, const double *M, int ms, double *a, double *x, double *temp, int *tr)
{
#define AA(i, j) a[(i) + (j) * m]
#define XX(i, j) x[(i) + (j) * m]
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) AA(i, j) = M[i * ms + j];
    for (int k = 0; k < m; k++) {
        int big = k; double bv = fabs(AA(k, k));
        for (int i = k + 1; i < m; i++) if (fabs(AA(i, i)) > bv) { bv = fabs(AA(i, i)); big = i; }
        tr[k] = big;
        if (big != k) {
            for (int j = 0; j < k; j++) { double t = AA(k, j); AA(k, j) = AA(big, j); AA(big, j) = t; }
            for (int i = big + 1; i < m; i++) { double t = AA(i, k); AA(i, k) = AA(i, big); AA(i, big) = t; }
            { double t = AA(k, k); AA(k, k) = AA(big, big); AA(big, big) = t; }
            for (int i = k + 1; i < big; i++) { double t = AA(i, k); AA(i, k) = AA(big, i); AA(big, i) = t; }
        }
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = AA(j, j) * AA(k, j);
            double dot = 0.0;
            for (int j = 0; j < k; j++) dot += AA(k, j) * temp[j];
            AA(k, k) -= dot;
            for (int i = k + 1; i < m; i++) {
                double d2 = 0.0;
                for (int j = 0; j < k; j++) d2 += AA(i, j) * temp[j];
                AA(i, k) -= d2;
            }
        }
        const double akk = AA(k, k);
        const bool valid = fabs(akk) > 0.0;
        if (k == 0 && !valid) { for (int j = 0; j < m; j++) tr[j] = j; break; }
        if (valid) for (int i = k + 1; i < m; i++) AA(i, k) /= akk;
    }
    for (int c = 0; c < m; c++) for (int r = 0; r < m; r++) XX(r, c) = (r == c) ? 1.0 : 0.0;
    for (int k = 0; k < m; k++)
        if (tr[k] != k) for (int c = 0; c < m; c++) { double t = XX(k, c); XX(k, c) = XX(tr[k], c); XX(tr[k], c) = t; }
    for (int c = 0; c < m; c++)
        for (int k = 0; k < m; k++) {
            const double b = XX(k, c);
            for (int i = k + 1; i < m; i++) XX(i, c) -= b * AA(i, k);
        }
    for (int i = 0; i < m; i++) {
        const double d = AA(i, i);
        for (int c = 0; c < m; c++) {
            if (fabs(d) > 2.2250738585072014e-308) XX(i, c) /= d; else XX(i, c) = 0.0;
        }
    }
    for (int c = 0; c < m; c++)
        for (int k = m - 1; k >= 0; k--) {
            const double b = XX(k, c);
            for (int i = 0; i < k; i++) XX(i, c) -= b * AA(k, i);
        }
    for (int k = m - 1; k >= 0; k--)
        if (tr[k] != k) for (int c = 0; c < m; c++) { double t = XX(k, c); XX(k, c) = XX(tr[k], c); XX(tr[k], c) = t; }
#undef AA
#undef XX
}

}  // namespace kpilqr
