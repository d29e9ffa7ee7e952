/*
 * kpilqr_oracle.c -- see kpilqr_oracle.h.  TEST INFRASTRUCTURE, not product code.
 * Compile with -ffp-contract=off: the reference is built for baseline x86-64 (-O3, no -march,
 * CMakeLists.txt:4-6), i.e. without FMA contraction, and Keypoints_Test.cpp:273-289 pins the
 * interpolation bitwise.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Dense products use plain k-ordered dot products; Eigen's GEMM/GEMV kernels may sum in a
 * different order (Eigen version unpinned, CMakeLists.txt:21), which moves results by a few
 * ulp only -- the tolerance for K is 1e-6 relative (BASELINE.json north_star).
 */
#include "kpilqr_oracle.h"

#define _POSIX_C_SOURCE 200809L
#include <float.h>
#include <pthread.h>
#include <time.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* a2: src/Differentiator/Differentiator.cpp:166-222 (ctrl), 286-321 (vel), 386-423 (pos),
 *     scatter into A/B :441-457.                                                         */
void orc_fd_difference(int n, int m, int njobs,
                       const int *job_t, const int *job_col, const unsigned char *job_mode,
                       const int *job_nom,
                       const double *xplus, const double *xminus, const double *xnom,
                       double eps, double *A, double *B)
{
    for (int j = 0; j < njobs; j++) {
        const int t = job_t[j], col = job_col[j];
        const double *xp = xplus + (size_t)j * n;
        const double *xm = xminus + (size_t)j * n;
        const double *x0 = xnom ? xnom + (size_t)job_nom[j] * n : 0;
        double *dst = (col < n) ? A + (size_t)t * n * n + (size_t)col * n
                                : B + (size_t)t * n * m + (size_t)(col - n) * n;
        for (int r = 0; r < n; r++) {
            double v;
            if (job_mode[j] == 0)      v = (xp[r] - xm[r]) / (2 * eps);   /* :170-178 */
            else if (job_mode[j] == 1) v = (xp[r] - x0[r]) / (eps);       /* :189-197 */
            else                       v = (x0[r] - xm[r]) / (eps);       /* :207-215 */
            dst[r] = v;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* a3: key-point generators                                                              */

/* src/KeyPointGenerator/KeyPointGenerator.cpp:319-339 */
int orc_kp_set_interval(int dof, int T, int min_N, int *offs, int *cols)
{
    int cnt = 0;
    for (int t = 0; t < T - 1; t++) {
        offs[t] = cnt;
        if (t % min_N == 0)
            for (int i = 0; i < dof; i++) cols[cnt++] = i;
    }
    offs[T - 1] = cnt;                       /* "Always push the last row" :337-338 */
    for (int i = 0; i < dof; i++) cols[cnt++] = i;
    offs[T] = cnt;
    return cnt;
}

/* GenerateAccellerationProfile :772-795 + GenerateKeyPointsAdaptive :341-382 (dispatch :98-101).  The profile is the
 * SIGNED velocity difference of consecutive steps (no division by dt, no abs), horizon-1 entries; the placement compares it
 * with jerk_thresholds (GenerateKeyPointsAdaptive reads that member whatever the profile, :360). */
int orc_kp_adaptive_accel(int dof, int T, int min_N, int max_N, const double *thr,
                          const double *X, int *offs, int *cols)
{
    const int n = 2 * dof;
    double *acc = (double *)calloc((size_t)T * dof, sizeof(double));
    for (int t = 0; t < T - 1; t++) {                                /* :783 */
        const double *s1 = X + (size_t)t * n, *s2 = s1 + n;
        for (int j = 0; j < dof; j++) acc[(size_t)t * dof + j] = s2[j + dof] - s1[j + dof];   /* :785-788 */
    }
    int cnt = 0;
    int *last = (int *)calloc((size_t)dof, sizeof(int));
    offs[0] = 0;
    for (int i = 0; i < dof; i++) cols[cnt++] = i;                   /* :347 */
    for (int t = 1; t < T - 1; t++) {
        offs[t] = cnt;
        for (int j = 0; j < dof; j++) {
            if ((t - last[j]) >= min_N) {                            /* :359 */
                if (acc[(size_t)t * dof + j] > thr[j]) { cols[cnt++] = j; last[j] = t; }
            }
            if ((t - last[j]) >= max_N) { cols[cnt++] = j; last[j] = t; }   /* :365 */
        }
    }
    offs[T - 1] = cnt;
    for (int i = 0; i < dof; i++) cols[cnt++] = i;                   /* :381 */
    offs[T] = cnt;
    free(acc); free(last);
    return cnt;
}

/* GenerateJerkProfile :730-770 + GenerateKeyPointsAdaptive :341-382 */
int orc_kp_adaptive_jerk(int dof, int T, int min_N, int max_N, const double *thr,
                         double dt, const double *X, int *offs, int *cols)
{
    const int n = 2 * dof;
    double *jerk = (double *)calloc((size_t)T * dof, sizeof(double));
    for (int t = 0; t < T - 2; t++) {
        const double *s1 = X + (size_t)t * n, *s2 = s1 + n, *s3 = s2 + n;
        for (int j = 0; j < dof; j++) {
            double a1 = (s2[j + dof] - s1[j + dof]) / dt;            /* :748 */
            double a2 = (s3[j + dof] - s2[j + dof]) / dt;            /* :749 */
            jerk[(size_t)t * dof + j] = fabs((a2 - a1) / dt);        /* :752 */
        }
    }
    /* last two rows stay zero (:757-761) */
    int cnt = 0;
    int *last = (int *)calloc((size_t)dof, sizeof(int));
    offs[0] = 0;
    for (int i = 0; i < dof; i++) cols[cnt++] = i;                   /* :347 */
    for (int t = 1; t < T - 1; t++) {
        offs[t] = cnt;
        for (int j = 0; j < dof; j++) {
            if ((t - last[j]) >= min_N) {                            /* :359 */
                if (jerk[(size_t)t * dof + j] > thr[j]) { cols[cnt++] = j; last[j] = t; }
            }
            if ((t - last[j]) >= max_N) { cols[cnt++] = j; last[j] = t; }   /* :365 */
        }
    }
    offs[T - 1] = cnt;
    for (int i = 0; i < dof; i++) cols[cnt++] = i;                   /* :381 */
    offs[T] = cnt;
    free(jerk); free(last);
    return cnt;
}

/* GenerateVelocityProfile :797-808 + GenerateKeyPointsVelocityChange :642-728 */
int orc_kp_velocity_change(int dof, int T, int min_N, int max_N, const double *thr,
                           const double *X, int *offs, int *cols)
{
    const int n = 2 * dof;
    int cnt = 0;
    int *counter = (int *)calloc((size_t)dof, sizeof(int));
    double *last_val = (double *)calloc((size_t)dof, sizeof(double));
    double *last_dir = (double *)calloc((size_t)dof, sizeof(double));
    offs[0] = 0;
    for (int i = 0; i < dof; i++) cols[cnt++] = i;
    for (int t = 1; t < T; t++) {
        offs[t] = cnt;
        for (int i = 0; i < dof; i++) {
            counter[i]++;
            double v = X[(size_t)t * n + dof + i], vp = X[(size_t)(t - 1) * n + dof + i];
            double dir = v - vp;                                     /* :671 */
            last_val[i] += fabs(v);                                  /* :673 */
            if (counter[i] >= min_N) {                               /* :676 */
                if (fabs(last_val[i]) > thr[i]) {
                    cols[cnt++] = i; last_val[i] = 0.0; counter[i] = 0; continue;
                }
            }
            if (counter[i] >= min_N) {                               /* :687 */
                if (dir * last_dir[i] < 0) {
                    cols[cnt++] = i; last_val[i] = 0.0; counter[i] = 0; continue;
                }
            } else {
                last_dir[i] = dir;                                   /* :697-699 */
            }
            if (counter[i] >= max_N) {                               /* :702 */
                cols[cnt++] = i; last_val[i] = 0.0; counter[i] = 0; continue;
            }
        }
    }
    /* "Enforce last keypoint for all dofs" -- appended to row T-1, duplicates and all (:724-727) */
    for (int i = 0; i < dof; i++) cols[cnt++] = i;
    offs[T] = cnt;
    free(counter); free(last_val); free(last_dir);
    return cnt;
}

/* CheckDOFColumnError :550-640 on a dense A sequence. computed: [dof][T] flags. */
static int ie_check(int dof, int T, int min_N, double threshold, const double *A,
                    unsigned char *computed, int s, int e, int d)
{
    const int n = 2 * dof;
    int mid = (s + e) / 2;
    if ((e - s) <= min_N) return 1;                                  /* :562-564 */
    computed[(size_t)d * T + s] = 1;                                 /* :588-604 */
    computed[(size_t)d * T + mid] = 1;
    computed[(size_t)d * T + e] = 1;
    double err = 0.0; int counter = 0;
    const int colidx[2] = { d, d + dof };
    for (int i = 0; i < 2; i++) {
        const double *cs = A + (size_t)s * n * n + (size_t)colidx[i] * n;
        const double *ce = A + (size_t)e * n * n + (size_t)colidx[i] * n;
        const double *cm = A + (size_t)mid * n * n + (size_t)colidx[i] * n;
        for (int j = dof; j < n; j++) {
            double approx = (cs[j] + ce[j]) / 2;                     /* :606-607 */
            double diff = cm[j] - approx;
            err += diff * diff;                                      /* pow(x,2) :615 */
            counter++;
        }
    }
    double avg = counter > 0 ? err / counter : 0.0;
    return avg < threshold;                                          /* :636-639 */
}

/* GenerateKeyPointsIteratively :449-548 */
int orc_kp_iterative_error(int dof, int T, int min_N, double threshold, const double *A,
                           int *offs, int *cols)
{
    unsigned char *computed = (unsigned char *)calloc((size_t)dof * T, 1);
    int cap = 2 * T + 4;
    int *cur = (int *)malloc(sizeof(int) * 2 * cap), *nxt = (int *)malloc(sizeof(int) * 2 * cap);
    for (int d = 0; d < dof; d++) {
        int ncur = 1; cur[0] = 0; cur[1] = T - 1;
        for (;;) {
            int all_ok = 1, nn = 0;
            for (int j = 0; j < ncur; j++) {
                int s = cur[2 * j], e = cur[2 * j + 1], mid = (s + e) / 2;
                if (!ie_check(dof, T, min_N, threshold, A, computed, s, e, d)) {
                    all_ok = 0;
                    if (nn + 2 > cap) { cap *= 2; nxt = (int *)realloc(nxt, sizeof(int) * 2 * cap);
                                        cur = (int *)realloc(cur, sizeof(int) * 2 * cap); }
                    nxt[2 * nn] = s;   nxt[2 * nn + 1] = mid; nn++;
                    nxt[2 * nn] = mid; nxt[2 * nn + 1] = e;   nn++;
                }
            }
            if (all_ok) break;
            int *tmp = cur; cur = nxt; nxt = tmp; ncur = nn;
        }
    }
    int cnt = 0;
    for (int t = 0; t < T; t++) {                                    /* :528-545 */
        offs[t] = cnt;
        for (int d = 0; d < dof; d++)
            if (computed[(size_t)d * T + t]) cols[cnt++] = d;
    }
    offs[T] = cnt;
    free(computed); free(cur); free(nxt);
    return cnt;
}

/* ComputePercentageDerivatives :810-838 */
void orc_kp_percentages(int dof, int T, const int *offs, const int *cols, double *pct)
{
    for (int d = 0; d < dof; d++) pct[d] = 0.0;
    for (int t = 0; t < T; t++)
        for (int e = offs[t]; e < offs[t + 1]; e++)
            if (cols[e] >= 0 && cols[e] < dof) pct[cols[e]] += 1.0;
    for (int d = 0; d < dof; d++) pct[d] = (pct[d] / (double)T) * 100;
}

/* ------------------------------------------------------------------------------------ */
/* a4: src/KeyPointGenerator/KeyPointGenerator.cpp:840-954, loop structure kept literally
 * (including: B column i follows DoF i's key-points only when i < num_ctrl, :927-931).   */
void orc_interpolate(int dof, int m, int T, const int *offs, const int *cols,
                     double *A, double *B)
{
    const int n = 2 * dof;
    int *start = (int *)calloc((size_t)dof, sizeof(int));            /* :874-877 */
    double *add1 = (double *)malloc(sizeof(double) * n);
    double *add2 = (double *)malloc(sizeof(double) * n);
    double *addB = (double *)malloc(sizeof(double) * n);
    for (int t = 1; t < T; t++) {                                    /* :881 */
        for (int i = 0; i < dof; i++) {
            for (int e = offs[t]; e < offs[t + 1]; e++) {
                if (cols[e] != i) continue;                          /* :896 */
                const int s = start[i];
                const double span = (double)(t - s);
                double *As = A + (size_t)s * n * n, *At = A + (size_t)t * n * n;
                const double *s1 = As + (size_t)i * n, *e1 = At + (size_t)i * n;
                const double *s2 = As + (size_t)(i + dof) * n, *e2 = At + (size_t)(i + dof) * n;
                for (int r = 0; r < n; r++) {
                    add1[r] = (e1[r] - s1[r]) / span;                /* :900 */
                    add2[r] = (e2[r] - s2[r]) / span;                /* :905 */
                }
                const double *sB = 0;
                if (i < m) {                                         /* :927-931 */
                    sB = B + (size_t)s * n * m + (size_t)i * n;
                    const double *eB = B + (size_t)t * n * m + (size_t)i * n;
                    for (int r = 0; r < n; r++) addB[r] = (eB[r] - sB[r]) / span;
                }
                for (int k = s + 1; k < t; k++) {                    /* :933-948 */
                    const double f = (double)(k - s);
                    double *Ak = A + (size_t)k * n * n;
                    for (int r = 0; r < n; r++) {
                        Ak[(size_t)i * n + r] = s1[r] + (f * add1[r]);
                        Ak[(size_t)(i + dof) * n + r] = s2[r] + (f * add2[r]);
                    }
                    if (i < m) {
                        double *Bk = B + (size_t)k * n * m + (size_t)i * n;
                        for (int r = 0; r < n; r++) Bk[r] = sB[r] + (f * addB[r]);
                    }
                }
                start[i] = t;                                        /* :949 */
            }
        }
    }
    free(start); free(add1); free(add2); free(addB);
}

/* ------------------------------------------------------------------------------------ */
/* a6 */
/* src/ModelTranslator/ModelTranslator.cpp:314-327 */
double orc_cost_function(int nr, const double *r, const double *w)
{
    double cost = 0.0;
    for (int i = 0; i < nr; i++) cost += w[i] * (r[i] * r[i]);      /* w * pow(r,2) */
    return cost;
}

/* src/ModelTranslator/ModelTranslator.cpp:552-583 for one time-step */
static void cost_derivs_step(int n, int m, int nr, const double *r, const double *rx,
                             const double *ru, const double *w,
                             double *l_x, double *l_xx, double *l_u, double *l_uu)
{
    memset(l_x, 0, sizeof(double) * n);
    memset(l_xx, 0, sizeof(double) * n * n);
    memset(l_u, 0, sizeof(double) * m);
    memset(l_uu, 0, sizeof(double) * m * m);
    for (int i = 0; i < nr; i++) {
        const double w2 = w[i] * 2;                                  /* weight_term * 2 */
        const double s = w2 * r[i];                                  /* ... * residuals(i) */
        const double *rxi = rx + (size_t)i * n, *rui = ru + (size_t)i * m;
        for (int a = 0; a < n; a++) l_x[a] += s * rxi[a];            /* :570 */
        for (int b = 0; b < n; b++)                                  /* :573 outer product */
            for (int a = 0; a < n; a++) l_xx[a + (size_t)b * n] += (w2 * rxi[a]) * rxi[b];
        for (int a = 0; a < m; a++) l_u[a] += s * rui[a];            /* :576 */
        for (int b = 0; b < m; b++)                                  /* :578 */
            for (int a = 0; a < m; a++) l_uu[a + (size_t)b * m] += (w2 * rui[a]) * rui[b];
    }
}

/* loop of src/Optimiser/Optimiser.cpp:202-211 */
void orc_cost_derivs(int n, int m, int nr, int T,
                     const double *r, const double *r_x, const double *r_u,
                     const double *w_run, const double *w_term,
                     double *l_x, double *l_xx, double *l_u, double *l_uu)
{
    for (int t = 0; t < T; t++)
        cost_derivs_step(n, m, nr, r + (size_t)t * nr, r_x + (size_t)t * nr * n,
                         r_u + (size_t)t * nr * m, w_run,
                         l_x + (size_t)t * n, l_xx + (size_t)t * n * n,
                         l_u + (size_t)t * m, l_uu + (size_t)t * m * m);
    const int t = T - 1;                                             /* :208-211 terminal */
    cost_derivs_step(n, m, nr, r + (size_t)t * nr, r_x + (size_t)t * nr * n,
                     r_u + (size_t)t * nr * m, w_term,
                     l_x + (size_t)t * n, l_xx + (size_t)t * n * n,
                     l_u + (size_t)t * m, l_uu + (size_t)t * m * m);
}

/* ------------------------------------------------------------------------------------ */
/* Eigen pieces (Eigen itself is not vendored in the reference; algorithm restated from the
 * published Eigen 3.3/3.4 sources: Cholesky/LLT.h llt_inplace<Lower>::unblocked and
 * Cholesky/LDLT.h ldlt_inplace<Lower>::unblocked + LDLT::_solve_impl).                    */

/* Eigen::LLT<MatrixXd>(M).info() == Success, as used by iLQR::CheckMatrixPD
 * (src/Optimiser/iLQR.cpp:659-670).  Reads the lower triangle only. */
int orc_llt_is_pd(int m, const double *M)
{
    double *L = (double *)malloc(sizeof(double) * m * m);
    memcpy(L, M, sizeof(double) * m * m);
    int ok = 1;
    for (int k = 0; k < m && ok; k++) {
        double x = L[k + (size_t)k * m];
        for (int j = 0; j < k; j++) x -= L[k + (size_t)j * m] * L[k + (size_t)j * m];
        if (x <= 0.0) { ok = 0; break; }
        x = sqrt(x);
        L[k + (size_t)k * m] = x;
        for (int i = k + 1; i < m; i++) {
            double v = L[i + (size_t)k * m];
            for (int j = 0; j < k; j++) v -= L[i + (size_t)j * m] * L[k + (size_t)j * m];
            L[i + (size_t)k * m] = v / x;
        }
    }
    free(L);
    return ok;
}

/* (M).ldlt().solve(Identity) as in src/Optimiser/iLQR.cpp:597-600: diagonal-pivoted LDL^T on
 * the lower triangle, then P^T L^-T D^-1 L^-1 P applied to I. */
void orc_ldlt_inverse(int m, const double *M, double *Minv)
{
    double *a = (double *)malloc(sizeof(double) * m * m);
    double *temp = (double *)malloc(sizeof(double) * m);
    int *tr = (int *)malloc(sizeof(int) * m);
    memcpy(a, M, sizeof(double) * m * m);
#define AA(i, j) a[(i) + (size_t)(j) * m]
    for (int k = 0; k < m; k++) {
        /* largest |diagonal| in the trailing corner */
        int big = k; double bv = fabs(AA(k, k));
        for (int i = k + 1; i < m; i++) if (fabs(AA(i, i)) > bv) { bv = fabs(AA(i, i)); big = i; }
        tr[k] = big;
        if (big != k) {
            for (int j = 0; j < k; j++) { double t = AA(k, j); AA(k, j) = AA(big, j); AA(big, j) = t; }
            for (int i = big + 1; i < m; i++) { double t = AA(i, k); AA(i, k) = AA(i, big); AA(i, big) = t; }
            { double t = AA(k, k); AA(k, k) = AA(big, big); AA(big, big) = t; }
            for (int i = k + 1; i < big; i++) { double t = AA(i, k); AA(i, k) = AA(big, i); AA(big, i) = t; }
        }
        const int rs = m - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = AA(j, j) * AA(k, j);      /* D * A10^T */
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
        const int pivot_valid = fabs(akk) > 0.0;
        if (k == 0 && !pivot_valid) {             /* matrix is all zero */
            for (int j = 0; j < m; j++) tr[j] = j;
            break;
        }
        if (rs > 0 && pivot_valid)
            for (int i = k + 1; i < m; i++) AA(i, k) /= akk;
    }
    /* solve for the identity */
    double *x = Minv;
    for (int c = 0; c < m; c++) for (int r = 0; r < m; r++) x[r + (size_t)c * m] = (r == c) ? 1.0 : 0.0;
#define XX(i, j) x[(i) + (size_t)(j) * m]
    for (int k = 0; k < m; k++)                                       /* dst = P * rhs */
        if (tr[k] != k) for (int c = 0; c < m; c++) { double t = XX(k, c); XX(k, c) = XX(tr[k], c); XX(tr[k], c) = t; }
    for (int c = 0; c < m; c++)                                       /* L (unit lower) */
        for (int k = 0; k < m; k++) {
            const double b = XX(k, c);
            for (int i = k + 1; i < m; i++) XX(i, c) -= b * AA(i, k);
        }
    const double tol = DBL_MIN;                                       /* (numeric_limits::min)() */
    for (int i = 0; i < m; i++) {
        const double d = AA(i, i);
        for (int c = 0; c < m; c++) {
            if (fabs(d) > tol) XX(i, c) /= d; else XX(i, c) = 0.0;
        }
    }
    for (int c = 0; c < m; c++)                                       /* L^T (unit upper) */
        for (int k = m - 1; k >= 0; k--) {
            const double b = XX(k, c);
            for (int i = 0; i < k; i++) XX(i, c) -= b * AA(k, i);
        }
    for (int k = m - 1; k >= 0; k--)                                  /* dst = P^T * dst */
        if (tr[k] != k) for (int c = 0; c < m; c++) { double t = XX(k, c); XX(k, c) = XX(tr[k], c); XX(tr[k], c) = t; }
#undef XX
#undef AA
    free(a); free(temp); free(tr);
}

/* C(r x c) = op(A)(r x kk) * B(kk x c); column-major, k-ordered dot products. */
static void mm(int r, int kk, int c, const double *A, int lda, int transA,
               const double *B, int ldb, double *C)
{
    for (int j = 0; j < c; j++)
        for (int i = 0; i < r; i++) {
            double s = 0.0;
            for (int p = 0; p < kk; p++) {
                const double a = transA ? A[p + (size_t)i * lda] : A[i + (size_t)p * lda];
                s += a * B[p + (size_t)j * ldb];
            }
            C[i + (size_t)j * r] = s;
        }
}

/* a7: src/Optimiser/iLQR.cpp:535-634 */
int orc_backward(int n, int m, int T,
                 const double *A, const double *B,
                 const double *l_x, const double *l_xx, const double *l_u, const double *l_uu,
                 double lambda, int pd_stride, double *K, double *k, double *delta_J)
{
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm_ = (size_t)m * m;
    double *V_x = (double *)malloc(sizeof(double) * n), *V_xx = (double *)malloc(sizeof(double) * nn);
    double *Q_x = (double *)malloc(sizeof(double) * n), *Q_u = (double *)malloc(sizeof(double) * m);
    double *Q_xx = (double *)malloc(sizeof(double) * nn), *Q_uu = (double *)malloc(sizeof(double) * mm_);
    double *Q_ux = (double *)malloc(sizeof(double) * nm), *Q_uu_reg = (double *)malloc(sizeof(double) * mm_);
    double *inv = (double *)malloc(sizeof(double) * mm_);
    double *AtV = (double *)malloc(sizeof(double) * nn), *BtV = (double *)malloc(sizeof(double) * nm);
    double *t1 = (double *)malloc(sizeof(double) * nn), *t2 = (double *)malloc(sizeof(double) * nn);
    double *t3 = (double *)malloc(sizeof(double) * nn), *t4 = (double *)malloc(sizeof(double) * nn);
    int ret = 0;

    memcpy(V_x, l_x + (size_t)(T - 1) * n, sizeof(double) * n);        /* :537 */
    memcpy(V_xx, l_xx + (size_t)(T - 1) * nn, sizeof(double) * nn);    /* :539 */
    int pd_counter = 0;
    double dJ = 0.0;                                                   /* :555 */

    for (int t = T - 1; t >= 0; t--) {                                 /* :560 */
        const double *At = A + (size_t)t * nn, *Bt = B + (size_t)t * nm;
        double *Kt = K + (size_t)t * nm, *kt = k + (size_t)t * m;
        pd_counter++;                                                  /* :565 */

        mm(n, n, 1, At, n, 1, V_x, n, t1);                             /* A' V_x  :570 */
        for (int i = 0; i < n; i++) Q_x[i] = l_x[(size_t)t * n + i] + t1[i];
        mm(m, n, 1, Bt, n, 1, V_x, n, t1);                             /* B' V_x  :572 */
        for (int i = 0; i < m; i++) Q_u[i] = l_u[(size_t)t * m + i] + t1[i];

        mm(n, n, n, At, n, 1, V_xx, n, AtV);                           /* (A' V_xx) A  :575 */
        mm(n, n, n, AtV, n, 0, At, n, t1);
        for (size_t i = 0; i < nn; i++) Q_xx[i] = l_xx[(size_t)t * nn + i] + t1[i];

        mm(m, n, n, Bt, n, 1, V_xx, n, BtV);                           /* (B' V_xx) B  :577 */
        mm(m, n, m, BtV, m, 0, Bt, n, t1);
        for (size_t i = 0; i < mm_; i++) Q_uu[i] = l_uu[(size_t)t * mm_ + i] + t1[i];

        mm(m, n, n, BtV, m, 0, At, n, Q_ux);                           /* (B' V_xx) A  :579 */

        memcpy(Q_uu_reg, Q_uu, sizeof(double) * mm_);                  /* :581-585 */
        for (int i = 0; i < m; i++) Q_uu_reg[i + (size_t)i * m] += lambda;

        if (pd_counter >= pd_stride) {                                 /* :587-595 */
            if (!orc_llt_is_pd(m, Q_uu_reg)) { ret = t + 1; break; }
            pd_counter = 0;
        }

        orc_ldlt_inverse(m, Q_uu_reg, inv);                            /* :597-600 */

        for (size_t i = 0; i < mm_; i++) t2[i] = -inv[i];              /* -Q_uu_inv */
        mm(m, m, 1, t2, m, 0, Q_u, m, kt);                             /* :603 */
        mm(m, m, n, t2, m, 0, Q_ux, m, Kt);                            /* :604 */

        /* V_x = Q_x + K'(Q_uu k) + K'Q_u + Q_ux' k   :606 */
        mm(m, m, 1, Q_uu, m, 0, kt, m, t1);
        mm(n, m, 1, Kt, m, 1, t1, m, t2);
        mm(n, m, 1, Kt, m, 1, Q_u, m, t3);
        mm(n, m, 1, Q_ux, m, 1, kt, m, t4);
        for (int i = 0; i < n; i++) V_x[i] = ((Q_x[i] + t2[i]) + t3[i]) + t4[i];

        /* V_xx = Q_xx + K'(Q_uu K) + K'Q_ux + Q_ux' K   :607 */
        mm(m, m, n, Q_uu, m, 0, Kt, m, t1);
        mm(n, m, n, Kt, m, 1, t1, m, t2);
        mm(n, m, n, Kt, m, 1, Q_ux, m, t3);
        mm(n, m, n, Q_ux, m, 1, Kt, m, t4);
        for (size_t i = 0; i < nn; i++) V_xx[i] = ((Q_xx[i] + t2[i]) + t3[i]) + t4[i];

        /* V_xx = (V_xx + V_xx')/2   :610.  The reference assigns this expression to V_xx itself
         * without .eval(); Eigen evaluates it coefficient by coefficient, column by column, IN
         * PLACE, so entries above the diagonal see the already-averaged entry below it.
         * Restated as executed (the two forms differ by O(eps) * asymmetry only). */
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++)
                V_xx[i + (size_t)j * n] = (V_xx[i + (size_t)j * n] + V_xx[j + (size_t)i * n]) / 2;

        /* delta_J += k'Q_u ; delta_J += (k'Q_uu) k   :612-613 */
        double s = 0.0;
        for (int i = 0; i < m; i++) s += kt[i] * Q_u[i];
        dJ += s;
        mm(1, m, m, kt, m, 1, Q_uu, m, t1);
        s = 0.0;
        for (int i = 0; i < m; i++) s += t1[i] * kt[i];
        dJ += s;
    }
    *delta_J = dJ;
    free(V_x); free(V_xx); free(Q_x); free(Q_u); free(Q_xx); free(Q_uu); free(Q_ux); free(Q_uu_reg);
    free(inv); free(AtV); free(BtV); free(t1); free(t2); free(t3); free(t4);
    return ret;
}

/* ------------------------------------------------------------------------------------ */
/* a8 */
void orc_alphas(int n_alpha, double *alphas)                          /* iLQR.cpp:466-470 */
{
    for (int i = 1; i < n_alpha + 1; i++) {
        double lin = (double)i / (n_alpha);
        alphas[i - 1] = lin * lin;
    }
}

void orc_forward_linear(int n, int m, int T, int n_alpha, const double *alphas,
                        const double *A, const double *B, const double *K, const double *k,
                        const double *l_x, const double *l_xx, const double *l_u, const double *l_uu,
                        const double *u_nom, const double *ctrl_lim,
                        double *cost_pred, double *U_alpha)
{
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm_ = (size_t)m * m;
    double *dx = (double *)malloc(sizeof(double) * n), *dxn = (double *)malloc(sizeof(double) * n);
    double *du = (double *)malloc(sizeof(double) * m), *fb = (double *)malloc(sizeof(double) * m);
    double *tv = (double *)malloc(sizeof(double) * (n > m ? n : m));
    for (int a = 0; a < n_alpha; a++) {
        const double alpha = alphas[a];
        double cost = 0.0;
        for (int i = 0; i < n; i++) dx[i] = 0.0;
        for (int t = 0; t < T; t++) {
            const double *At = A + t * nn, *Bt = B + t * nm, *Kt = K + t * nm, *kt = k + (size_t)t * m;
            const double *un = u_nom + (size_t)t * m;
            mm(m, n, 1, Kt, m, 0, dx, n, fb);                         /* K[t]*state_feedback :876 */
            for (int i = 0; i < m; i++) {
                double u = (un[i] + (alpha * kt[i])) + fb[i];         /* :879 */
                if (u > ctrl_lim[2 * i + 1]) u = ctrl_lim[2 * i + 1]; /* :883-889 */
                if (u < ctrl_lim[2 * i]) u = ctrl_lim[2 * i];
                if (U_alpha) U_alpha[((size_t)a * T + t) * m + i] = u;
                du[i] = u - un[i];
            }
            double c = 0.0, q;
            for (int i = 0; i < n; i++) c += l_x[(size_t)t * n + i] * dx[i];
            mm(n, n, 1, l_xx + t * nn, n, 0, dx, n, tv);
            q = 0.0; for (int i = 0; i < n; i++) q += dx[i] * tv[i];
            c += 0.5 * q;
            for (int i = 0; i < m; i++) c += l_u[(size_t)t * m + i] * du[i];
            mm(m, m, 1, l_uu + t * mm_, m, 0, du, m, tv);
            q = 0.0; for (int i = 0; i < m; i++) q += du[i] * tv[i];
            c += 0.5 * q;
            cost += c;
            mm(n, n, 1, At, n, 0, dx, n, dxn);
            mm(n, m, 1, Bt, n, 0, du, m, tv);
            for (int i = 0; i < n; i++) dx[i] = dxn[i] + tv[i];
        }
        cost_pred[a] = cost;
    }
    free(dx); free(dxn); free(du); free(fb); free(tv);
}

/* ------------------------------------------------------------------------------------ */
/* a9 */
int orc_update_lambda(double *lambda, int valid, double factor, double min_lambda, double max_lambda)
{                                                                     /* iLQR.cpp:636-657 */
    int lambda_exit = 0;
    if (!valid) *lambda *= factor; else *lambda /= factor;
    if (*lambda > max_lambda) { *lambda = max_lambda; lambda_exit = 1; }
    if (*lambda < min_lambda) *lambda = min_lambda;
    return lambda_exit;
}

int orc_check_convergence(double old_cost, double new_cost, double eps_converge)
{                                                                     /* Optimiser.cpp:30-37 */
    double g = (old_cost - new_cost) / new_cost;
    return g < eps_converge;
}

/* ------------------------------------------------------------------------------------ */
/* optional pieces either side of the hot path */

/* Optimiser::FilterDynamicsMatrices (src/Optimiser/Optimiser.cpp:340-406): rows dof..2dof-1 of A, every column,
 * filtered along time.  method 0 = low_pass (FilterIndValLowPass :372-388, coefs[0] = lowPassACoefficient),
 * method 1 = FIR (FilterIndValFIRFilter :390-406).  A: [T][n*n] column-major. */
void orc_filter_dynamics(int dof, int T, int method, const double *coefs, int ncoef, double *A)
{
    const int n = 2 * dof;
    double *u = (double *)malloc(sizeof(double) * (size_t)T), *f = (double *)malloc(sizeof(double) * (size_t)T);
    for (int i = dof; i < 2 * dof; i++)
        for (int j = 0; j < n; j++) {
            for (int k = 0; k < T; k++) u[k] = A[(size_t)k * n * n + i + (size_t)j * n];
            if (method == 0) {
                const double a = coefs[0];
                double yn1 = u[0], xn1 = u[0];
                for (int k = 0; k < T; k++) {
                    double xn = u[k];
                    double yn = ((1 - a) * yn1) + a * ((xn + xn1) / 2);       /* :380 */
                    xn1 = xn; yn1 = yn;
                    f[k] = yn;
                }
            } else {
                for (int k = 0; k < T; k++) f[k] = 0;
                for (int k = 0; k < T; k++)
                    for (int c = 0; c < ncoef; c++)
                        if (k - c >= 0) f[k] += u[k - c] * coefs[c];          /* :398-402 */
            }
            for (int k = 0; k < T; k++) A[(size_t)k * n * n + i + (size_t)j * n] = f[k];
        }
    free(u); free(f);
}

/* iLQR_SVR::LeastImportantDofs, "sampling and summing" branch (src/Optimiser/iLQR_SVR.cpp:952-968):
 * sums[i] = (sum over t = 0, s, 2s, ... and controls j of |K[t](j,i)| + |K[t](j,i+dof)|) / T.  K: [T][m*n]. */
void orc_dof_importance(int dof, int m, int T, int sampling, const double *K, double *sums)
{
    for (int i = 0; i < dof; i++) sums[i] = 0.0;
    for (int t = 0; t < T; t += sampling)
        for (int i = 0; i < dof; i++)
            for (int j = 0; j < m; j++) {
                sums[i] += fabs(K[(size_t)t * m * 2 * dof + j + (size_t)i * m]);
                sums[i] += fabs(K[(size_t)t * m * 2 * dof + j + (size_t)(i + dof) * m]);
            }
    for (int i = 0; i < dof; i++) sums[i] /= T;
}

/* iLQR_SVR line-search set (src/Optimiser/iLQR_SVR.cpp:469-471): 1 - i/n, i = 0..n-1 */
void orc_alphas_svr(int n_alpha, double *alphas)
{
    for (int i = 0; i < n_alpha; i++) alphas[i] = 1.0 - (double)i / n_alpha;
}

int orc_linesearch_accept(int n_alpha, const double *costs, double old_cost,
                          double *new_cost, int *accepted,
                          double *lambda, double lambda_factor, double max_lambda)
{                                                                     /* iLQR.cpp:490-528 */
    int best = 0;
    for (int i = 1; i < n_alpha; i++) if (costs[i] < costs[best]) best = i;  /* std::min_element */
    if (costs[best] < old_cost) { *new_cost = costs[best]; *accepted = 1; }
    else {
        *new_cost = old_cost; *accepted = 0;
        *lambda *= lambda_factor; *lambda *= lambda_factor;           /* :525-526 */
        if (*lambda > max_lambda) *lambda = max_lambda;
    }
    return best;
}

/* ------------------------------------------------------------------------------------ */
/* whole iteration for one trajectory + pthread batch driver                               */
int orc_iteration(const orc_problem *p, double *K, double *k, double *delta_J, double *cost_pred)
{
    const int n = 2 * p->dof, m = p->m, T = p->T;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm_ = (size_t)m * m;
    double *A = (double *)calloc((size_t)T * nn, sizeof(double)), *B = (double *)calloc((size_t)T * nm, sizeof(double));
    double *l_x = (double *)malloc(sizeof(double) * T * n), *l_xx = (double *)malloc(sizeof(double) * T * nn);
    double *l_u = (double *)malloc(sizeof(double) * T * m), *l_uu = (double *)malloc(sizeof(double) * T * mm_);
    double alphas[16];
    orc_fd_difference(n, m, p->njobs, p->job_t, p->job_col, p->job_mode, p->job_nom, p->xplus, p->xminus, p->xnom,
                      p->eps, A, B);
    orc_interpolate(p->dof, m, T, p->kp_offs, p->kp_cols, A, B);
    orc_cost_derivs(n, m, p->nr, T, p->r, p->r_x, p->r_u, p->w_run, p->w_term, l_x, l_xx, l_u, l_uu);
    const int st = orc_backward(n, m, T, A, B, l_x, l_xx, l_u, l_uu, p->lambda, p->pd_stride, K, k, delta_J);
    if (st == 0) {
        orc_alphas(p->n_alpha, alphas);
        orc_forward_linear(n, m, T, p->n_alpha, alphas, A, B, K, k, l_x, l_xx, l_u, l_uu, p->u_nom, p->ctrl_lim,
                           cost_pred, 0);
    }
    free(A); free(B); free(l_x); free(l_xx); free(l_u); free(l_uu);
    return st;
}

typedef struct { const orc_problem *p; int reps; } orc_worker_arg;

static void *orc_worker(void *arg_)
{
    const orc_worker_arg *arg = (const orc_worker_arg *)arg_;
    const orc_problem *p = arg->p;
    const int n = 2 * p->dof;
    double *K = (double *)malloc(sizeof(double) * (size_t)p->T * n * p->m), *k = (double *)malloc(sizeof(double) * (size_t)p->T * p->m);
    double dJ, cost[16];
    for (int i = 0; i < arg->reps; i++) orc_iteration(p, K, k, &dJ, cost);
    free(K); free(k);
    return 0;
}

double orc_iteration_batch(const orc_problem *p, int nthreads, int reps)
{
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    orc_worker_arg arg = { p, reps };
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], 0, orc_worker, &arg);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
