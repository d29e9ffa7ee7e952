// generic.hip -- dimension-generic backward (a7) and forward (a8) kernels.  One wavefront per
// trajectory (per (trajectory, alpha) for the forward pass), every per-step matrix in LDS, and
// every expression in the reference's association and summation order, so that (compiled with
// -ffp-contract=off) the results match the CPU oracle bit for bit.  These kernels serve
//   * dimensions the MFMA kernels do not cover (n+1 > 16), and
//   * as the on-device cross-check of the MFMA kernels (KPILQR_FLAG_GENERIC_KERNELS).
// They are latency-bound by design; the fast path is riccati_mfma.hip / forward_mfma.hip.
//
// Reference: iLQR::BackwardsPassQuuRegularisation + CheckMatrixPD, src/Optimiser/iLQR.cpp:535-670;
// control law / clamp of iLQR::ForwardsPassParallel, src/Optimiser/iLQR.cpp:876-890.
#include "common.h"

namespace kpilqr {

// ---- Eigen::LLT info()==Success on the lower triangle (Cholesky/LLT.h, unblocked) ---------------
__device__ static bool dev_llt_is_pd(int m, const double *M /*col-major*/, double *Lw)
{
    for (int i = 0; i < m * m; i++) Lw[i] = M[i];
    for (int k = 0; k < m; k++) {
        double x = Lw[k + k * m];
        for (int j = 0; j < k; j++) x -= Lw[k + j * m] * Lw[k + j * m];
        if (x <= 0.0) return false;
        x = sqrt(x);
        Lw[k + k * m] = x;
        for (int i = k + 1; i < m; i++) {
            double v = Lw[i + k * m];
            for (int j = 0; j < k; j++) v -= Lw[i + j * m] * Lw[k + j * m];
            Lw[i + k * m] = v / x;
        }
    }
    return true;
}

// ---- M.ldlt().solve(Identity): diagonal-pivoted LDL^T (Cholesky/LDLT.h) -------------------------
__device__ static void dev_ldlt_inverse(int m, const double *M, double *a, double *x, double *temp, int *tr)
{
#define AA(i, j) a[(i) + (j) * m]
#define XX(i, j) x[(i) + (j) * m]
    for (int i = 0; i < m * m; i++) a[i] = M[i];
    bool zero = false;
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
        if (k == 0 && !valid) { for (int j = 0; j < m; j++) tr[j] = j; zero = true; break; }
        if (valid) for (int i = k + 1; i < m; i++) AA(i, k) /= akk;
    }
    (void)zero;
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

size_t backward_generic_lds_bytes(int n, int m)
{
    // Vxx, AtV (n*n each); BtV, Qux, Kt, G (m*n each); Quu, Qreg, inv, a, x (m*m each);
    // Vx, Qx (n); Qu, kt, g, temp (m); tr (m ints, padded); flag
    size_t d = 2 * (size_t)n * n + 4 * (size_t)m * n + 5 * (size_t)m * m + 2 * n + 4 * m + m + 2;
    return d * sizeof(double);
}

__global__ void __launch_bounds__(64)
k_backward_generic(RecLayout L, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                   int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                   double *__restrict__ delta_J, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int n = L.n, m = L.m, tid = threadIdx.x, NT = blockDim.x;
    const int b = blockIdx.x;
    double *Vxx = sh, *AtV = Vxx + n * n, *BtV = AtV + n * n, *Qux = BtV + m * n, *Kt = Qux + m * n,
           *G = Kt + m * n, *Quu = G + m * n, *Qreg = Quu + m * m, *inv = Qreg + m * m, *wa = inv + m * m,
           *wx = wa + m * m, *Vx = wx + m * m, *Qx = Vx + n, *Qu = Qx + n, *kt = Qu + m, *g = kt + m,
           *temp = g + m;
    int *tr = (int *)(temp + m);
    int *flag = (int *)(temp + m + m);

    const double *R0 = rec + (size_t)b * T * L.stride;
    const double lam = lambda[b];
    // V_x = l_x[T-1]; V_xx = l_xx[T-1]     (iLQR.cpp:537-539); LDS matrices are (i,j) -> i + j*rows
    {
        const double *R = R0 + (size_t)(T - 1) * L.stride;
        for (int e = tid; e < n * n; e += NT) { const int i = e % n, j = e / n; Vxx[i + j * n] = R[L.off_lxx + i * n + j]; }
        for (int e = tid; e < n; e += NT) Vx[e] = R[L.off_lx + e];
    }
    if (tid == 0) *flag = 0;
    __syncthreads();

    int pd_counter = 0;
    double dJ = 0.0;
    int fail = 0;
    for (int t = T - 1; t >= 0; t--) {
        const double *R = R0 + (size_t)t * L.stride;
        const double *A = R + L.off_A, *B = R + L.off_B;       // column-major: A(r,c) = A[c*n+r], B(r,c) = B[c*n+r]
        pd_counter++;
        // -- Q_x, Q_u, A'V_xx, B'V_xx --------------------------------------------------  :570-579
        for (int e = tid; e < n * n; e += NT) {
            const int i = e % n, j = e / n;
            double s = 0.0;
            for (int p = 0; p < n; p++) s += A[i * n + p] * Vxx[p + j * n];
            AtV[i + j * n] = s;
        }
        for (int e = tid; e < m * n; e += NT) {
            const int i = e % m, j = e / m;
            double s = 0.0;
            for (int p = 0; p < n; p++) s += B[i * n + p] * Vxx[p + j * n];
            BtV[i + j * m] = s;
        }
        for (int i = tid; i < n; i += NT) {
            double s = 0.0;
            for (int p = 0; p < n; p++) s += A[i * n + p] * Vx[p];
            Qx[i] = R[L.off_lx + i] + s;
        }
        for (int i = tid; i < m; i += NT) {
            double s = 0.0;
            for (int p = 0; p < n; p++) s += B[i * n + p] * Vx[p];
            Qu[i] = R[L.off_lu + i] + s;
        }
        __syncthreads();
        for (int e = tid; e < m * m; e += NT) {
            const int i = e % m, j = e / m;
            double s = 0.0;
            for (int p = 0; p < n; p++) s += BtV[i + p * m] * B[j * n + p];
            Quu[i + j * m] = R[L.off_luu + i * m + j] + s;
        }
        for (int e = tid; e < m * n; e += NT) {
            const int i = e % m, j = e / m;
            double s = 0.0;
            for (int p = 0; p < n; p++) s += BtV[i + p * m] * A[j * n + p];
            Qux[i + j * m] = s;
        }
        __syncthreads();
        // -- regularise, PD test every pd_stride steps, explicit inverse -----------------  :581-600
        if (tid == 0) {
            for (int e = 0; e < m * m; e++) Qreg[e] = Quu[e];
            for (int i = 0; i < m; i++) Qreg[i + i * m] += lam;
            bool ok = true;
            if (pd_counter >= pd_stride) ok = dev_llt_is_pd(m, Qreg, wa);
            if (!ok) *flag = t + 1;
            else dev_ldlt_inverse(m, Qreg, wa, wx, temp, tr);
            if (ok) for (int e = 0; e < m * m; e++) inv[e] = wx[e];
        }
        __syncthreads();
        fail = *flag;
        if (fail) break;
        if (pd_counter >= pd_stride) pd_counter = 0;
        // -- k = -inv Q_u ; K = -inv Q_ux ---------------------------------------------------  :603-604
        for (int i = tid; i < m; i += NT) {
            double s = 0.0;
            for (int p = 0; p < m; p++) s += (-inv[i + p * m]) * Qu[p];
            kt[i] = s;
            kout[((size_t)b * T + t) * m + i] = s;
        }
        for (int e = tid; e < m * n; e += NT) {
            const int i = e % m, j = e / m;
            double s = 0.0;
            for (int p = 0; p < m; p++) s += (-inv[i + p * m]) * Qux[p + j * m];
            Kt[i + j * m] = s;
            Kout[((size_t)b * T + t) * m * n + e] = s;
        }
        __syncthreads();
        // -- Q_uu k, Q_uu K ---------------------------------------------------------------------
        for (int i = tid; i < m; i += NT) {
            double s = 0.0;
            for (int p = 0; p < m; p++) s += Quu[i + p * m] * kt[p];
            g[i] = s;
        }
        for (int e = tid; e < m * n; e += NT) {
            const int i = e % m, j = e / m;
            double s = 0.0;
            for (int p = 0; p < m; p++) s += Quu[i + p * m] * Kt[p + j * m];
            G[i + j * m] = s;
        }
        __syncthreads();
        // -- V_x, V_xx (:606-607), delta_J (:612-613) ------------------------------------------------
        for (int e = tid; e < n * n; e += NT) {
            const int i = e % n, j = e / n;
            double q = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
            for (int p = 0; p < n; p++) q += AtV[i + p * n] * A[j * n + p];
            q = R[L.off_lxx + i * n + j] + q;
            for (int p = 0; p < m; p++) s2 += Kt[p + i * m] * G[p + j * m];
            for (int p = 0; p < m; p++) s3 += Kt[p + i * m] * Qux[p + j * m];
            for (int p = 0; p < m; p++) s4 += Qux[p + i * m] * Kt[p + j * m];
            Vxx[i + j * n] = ((q + s2) + s3) + s4;
        }
        for (int i = tid; i < n; i += NT) {   // nothing reads V_x in this phase: update in place
            double s2 = 0.0, s3 = 0.0, s4 = 0.0;
            for (int p = 0; p < m; p++) s2 += Kt[p + i * m] * g[p];
            for (int p = 0; p < m; p++) s3 += Kt[p + i * m] * Qu[p];
            for (int p = 0; p < m; p++) s4 += Qux[p + i * m] * kt[p];
            Vx[i] = ((Qx[i] + s2) + s3) + s4;
        }
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < m; i++) s += kt[i] * Qu[i];
            dJ += s;
            s = 0.0;
            for (int j = 0; j < m; j++) {
                double tj = 0.0;
                for (int i = 0; i < m; i++) tj += kt[i] * Quu[i + j * m];
                temp[j] = tj;
            }
            for (int j = 0; j < m; j++) s += temp[j] * kt[j];
            dJ += s;
        }
        __syncthreads();
        // -- V_xx = (V_xx + V_xx')/2 evaluated IN PLACE column by column, as Eigen executes the
        //    aliased expression of iLQR.cpp:610 (see oracle/kpilqr_oracle.c): lower triangle gets the
        //    average, upper triangle the average of itself and that average.
        for (int e = tid; e < n * n; e += NT) {
            const int i = e % n, j = e / n;
            if (i > j) {
                const double lo = Vxx[i + j * n], up = Vxx[j + i * n];
                const double s = (lo + up) / 2;
                Vxx[i + j * n] = s;
                Vxx[j + i * n] = (up + s) / 2;
            } else if (i == j) {
                const double v = Vxx[i + j * n];
                Vxx[i + j * n] = (v + v) / 2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) { delta_J[b] = dJ; status[b] = fail; }
}

hipError_t launch_backward_generic(Ctx *c, int pd_stride)
{
    const size_t lds = backward_generic_lds_bytes(c->n, c->d.m);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void *)k_backward_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_backward_generic, dim3(c->d.batch), dim3(64), lds, c->stream, c->L, c->d.T, c->rec,
                       c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// a8, generic: one wavefront per (trajectory, alpha).
__global__ void __launch_bounds__(64)
k_forward_generic(RecLayout L, int T, int n_alpha, const double *__restrict__ rec,
                  const double *__restrict__ Kin, const double *__restrict__ kin,
                  const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                  const double *__restrict__ alphas, double *__restrict__ cost_pred,
                  double *__restrict__ U_alpha)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int n = L.n, m = L.m, tid = threadIdx.x, NT = blockDim.x;
    const int b = blockIdx.x, a = blockIdx.y;
    double *dx = sh, *dxn = dx + n, *tv = dxn + n, *du = tv + n, *fb = du + m, *tu = fb + m;
    const double alpha = alphas[a];
    for (int i = tid; i < n; i += NT) dx[i] = 0.0;
    __syncthreads();
    double cost = 0.0;
    for (int t = 0; t < T; t++) {
        const double *R = rec + ((size_t)b * T + t) * L.stride;
        const double *A = R + L.off_A, *B = R + L.off_B;
        const double *Kt = Kin + ((size_t)b * T + t) * m * n, *kt = kin + ((size_t)b * T + t) * m;
        const double *un = u_nom + ((size_t)b * T + t) * m;
        for (int i = tid; i < m; i += NT) {
            double s = 0.0;
            for (int p = 0; p < n; p++) s += Kt[i + p * m] * dx[p];          // K[t]*state_feedback  :876
            double u = (un[i] + (alpha * kt[i])) + s;                          // :879
            if (u > ctrl_lim[2 * i + 1]) u = ctrl_lim[2 * i + 1];              // :883-889
            if (u < ctrl_lim[2 * i]) u = ctrl_lim[2 * i];
            if (U_alpha) U_alpha[(((size_t)b * n_alpha + a) * T + t) * m + i] = u;
            du[i] = u - un[i];
        }
        for (int i = tid; i < n; i += NT) {
            double s = 0.0;
            for (int p = 0; p < n; p++) s += R[L.off_lxx + i * n + p] * dx[p];
            tv[i] = s;
            double s2 = 0.0;
            for (int p = 0; p < n; p++) s2 += A[p * n + i] * dx[p];
            dxn[i] = s2;
        }
        __syncthreads();
        for (int i = tid; i < m; i += NT) {
            double s = 0.0;
            for (int p = 0; p < m; p++) s += R[L.off_luu + i * m + p] * du[p];
            tu[i] = s;
        }
        for (int i = tid; i < n; i += NT) {
            double s = 0.0;
            for (int p = 0; p < m; p++) s += B[p * n + i] * du[p];
            dxn[i] = dxn[i] + s;
        }
        __syncthreads();
        if (tid == 0) {
            double c = 0.0, q = 0.0;
            for (int i = 0; i < n; i++) c += R[L.off_lx + i] * dx[i];
            for (int i = 0; i < n; i++) q += dx[i] * tv[i];
            c += 0.5 * q;
            for (int i = 0; i < m; i++) c += R[L.off_lu + i] * du[i];
            q = 0.0;
            for (int i = 0; i < m; i++) q += du[i] * tu[i];
            c += 0.5 * q;
            cost += c;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) dx[i] = dxn[i];
        __syncthreads();
    }
    if (tid == 0) cost_pred[(size_t)b * n_alpha + a] = cost;
}

hipError_t launch_forward_generic(Ctx *c, double *U_alpha_dev)
{
    const size_t lds = sizeof(double) * (3 * c->n + 3 * c->d.m);
    hipLaunchKernelGGL(k_forward_generic, dim3(c->d.batch, c->d.n_alpha), dim3(64), lds, c->stream, c->L, c->d.T,
                       c->d.n_alpha, c->rec, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred,
                       U_alpha_dev);
    return hipGetLastError();
}

}  // namespace kpilqr
