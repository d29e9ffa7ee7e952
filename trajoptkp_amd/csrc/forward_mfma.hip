// forward_mfma.hip -- linearised forward rollout over the line-search alphas (a8) for n+2 <= 16:
// ONE wavefront per trajectory, the n_alpha candidate steps are the COLUMNS of a 16x16 FP64 MFMA
// tile, so the per-step mat-vecs of all alphas become one small contraction
// (v_mfma_f64_16x16x4_f64).
//
// Reference control law + clamp: iLQR::ForwardsPassParallel, src/Optimiser/iLQR.cpp:876-890; alphas
// :466-470.  Dynamics and cost are the first/second-order models (declared semantic change,
// DESIGN.md section 2):  dx+ = A dx + B du,  cost += l_x'dx + dx'l_xx dx/2 + l_u'du + du'l_uu du/2.
//
// Tiles in the MFMA "D" layout (lane (c,q), register r <-> element (4r+q, c)); primitive
// P(Y,X) = Y'X as in riccati_mfma.hip.  State tile Z: rows 0..n-1 = dx, row n = alpha, row n+1 = 1,
// column a = line-search candidate a.  Then with
//     Yk = [K' ; k' ; u_nom']  (rows = state index / alpha / one, cols = control index)
//     U  = P(Yk, Z) = K dx + alpha k + u_nom            (:879, association differs by rounding)
//     dU = clamp(U) - u_nom                              (:883-889)
//     Z+ = P(Ya, Z) + P(Yb, dU),  Ya = [A' 0 0; 0 1 0; 0 0 1], Yb = B'
//     cost: lane-local  sum_r Z_r * P(Lc, Z)_r / 2 + dU_r * (P(Luu, dU)_r / 2 + l_u)   with
//           Lc = [l_xx 0 l_x; 0 0 0; l_x' 0 0]; per-lane partial sums over t, one cross-lane
//           reduction at the end.
#include "common.h"

namespace kpilqr {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

template <int NC>
__device__ __forceinline__ d4 PF(const d4 &Y, const d4 &X, d4 acc)
{
    acc = MFMA(Y.x, X.x, acc);
    if (NC > 1) acc = MFMA(Y.y, X.y, acc);
    if (NC > 2) acc = MFMA(Y.z, X.z, acc);
    if (NC > 3) acc = MFMA(Y.w, X.w, acc);
    return acc;
}

__device__ __forceinline__ double ldz(const double *R, int off)
{
    const double v = R[off < 0 ? 0 : off];
    return off < 0 ? 0.0 : v;
}

template <int NCZ, int NCU>
__global__ void __launch_bounds__(64)
k_forward_mfma(RecLayout L, int T, int n_alpha, const double *__restrict__ rec,
               const double *__restrict__ Kin, const double *__restrict__ kin,
               const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
               const double *__restrict__ alphas, double *__restrict__ cost_pred,
               double *__restrict__ U_alpha)
{
    const int n = L.n, m = L.m;
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;

    // per-lane source offsets (doubles); -1 = structural zero
    int oK[4], ok_[4], oun[4], oA[4], oB[4], oLc[4], oLuu[4], olu[4], oub[4];
    double oneA[4], lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        oK[r] = (row < n && c < m) ? row * m + c : -1;            // K(i=c, p=row) at c + row*m
        ok_[r] = (row == n && c < m) ? c : -1;                     // k(i=c)
        oun[r] = (row == n + 1 && c < m) ? c : -1;                 // u_nom(i=c)
        oA[r] = (row < n && c < n) ? L.off_A + c * n + row : -1;   // A(i=c, p=row)
        oneA[r] = ((row == n && c == n) || (row == n + 1 && c == n + 1)) ? 1.0 : 0.0;
        oB[r] = (row < m && c < n) ? L.off_B + c * m + row : -1;   // B(i=c, p=row)
        oLc[r] = (row < n && c < n) ? L.off_lxx + row * n + c
               : (row == n + 1 && c < n) ? L.off_lx + c
               : (c == n + 1 && row < n) ? L.off_lx + row : -1;
        oLuu[r] = (row < m && c < m) ? L.off_luu + row * m + c : -1;
        olu[r] = (row < m) ? L.off_lu + row : -1;
        oub[r] = (row < m) ? row : -1;
        lo[r] = (row < m) ? ctrl_lim[2 * row] : -1.0e300;
        hi[r] = (row < m) ? ctrl_lim[2 * row + 1] : 1.0e300;
    }
    const double my_alpha = (c < n_alpha) ? alphas[c] : 0.0;
    d4 Z = {0.0, 0.0, 0.0, 0.0};
    {
        double zr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + q;
            zr[r] = (row == n) ? my_alpha : (row == n + 1) ? 1.0 : 0.0;
        }
        Z.x = zr[0]; Z.y = zr[1]; Z.z = zr[2]; Z.w = zr[3];
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double partial = 0.0;

    struct Tiles { d4 Yk, Ya, Yb, Lc, Luu, lu, ub; };
    auto load_tiles = [&](int t, Tiles &s) {
        const double *R = rec + ((size_t)b * T + t) * L.stride;
        const double *Kt = Kin + ((size_t)b * T + t) * m * n;
        const double *kt = kin + ((size_t)b * T + t) * m;
        const double *un = u_nom + ((size_t)b * T + t) * m;
        s.Yk.x = ldz(Kt, oK[0]) + ldz(kt, ok_[0]) + ldz(un, oun[0]);
        s.Yk.y = ldz(Kt, oK[1]) + ldz(kt, ok_[1]) + ldz(un, oun[1]);
        s.Yk.z = ldz(Kt, oK[2]) + ldz(kt, ok_[2]) + ldz(un, oun[2]);
        s.Yk.w = ldz(Kt, oK[3]) + ldz(kt, ok_[3]) + ldz(un, oun[3]);
        s.Ya.x = ldz(R, oA[0]) + oneA[0]; s.Ya.y = ldz(R, oA[1]) + oneA[1];
        s.Ya.z = ldz(R, oA[2]) + oneA[2]; s.Ya.w = ldz(R, oA[3]) + oneA[3];
        s.Yb.x = ldz(R, oB[0]); s.Yb.y = ldz(R, oB[1]); s.Yb.z = ldz(R, oB[2]); s.Yb.w = ldz(R, oB[3]);
        s.Lc.x = ldz(R, oLc[0]); s.Lc.y = ldz(R, oLc[1]); s.Lc.z = ldz(R, oLc[2]); s.Lc.w = ldz(R, oLc[3]);
        s.Luu.x = ldz(R, oLuu[0]); s.Luu.y = ldz(R, oLuu[1]); s.Luu.z = ldz(R, oLuu[2]); s.Luu.w = ldz(R, oLuu[3]);
        s.lu.x = ldz(R, olu[0]); s.lu.y = ldz(R, olu[1]); s.lu.z = ldz(R, olu[2]); s.lu.w = ldz(R, olu[3]);
        s.ub.x = ldz(un, oub[0]); s.ub.y = ldz(un, oub[1]); s.ub.z = ldz(un, oub[2]); s.ub.w = ldz(un, oub[3]);
    };
    Tiles cur, nxt;
    load_tiles(0, cur);
    nxt = cur;

    for (int t = 0; t < T; t++) {
        if (t + 1 < T) load_tiles(t + 1, nxt);      // prefetch one step ahead
        const d4 Yk = cur.Yk, Ya = cur.Ya, Yb = cur.Yb, Lc = cur.Lc, Luu = cur.Luu, lu = cur.lu, ub = cur.ub;

        // control law + clamp
        d4 U = PF<NCZ>(Yk, Z, zero);
        d4 dU;
        {
            double u;
            u = U.x; if (u > hi[0]) u = hi[0]; if (u < lo[0]) u = lo[0]; U.x = u; dU.x = u - ub.x;
            u = U.y; if (u > hi[1]) u = hi[1]; if (u < lo[1]) u = lo[1]; U.y = u; dU.y = u - ub.y;
            u = U.z; if (u > hi[2]) u = hi[2]; if (u < lo[2]) u = lo[2]; U.z = u; dU.z = u - ub.z;
            u = U.w; if (u > hi[3]) u = hi[3]; if (u < lo[3]) u = lo[3]; U.w = u; dU.w = u - ub.w;
        }
        if (U_alpha && c < n_alpha) {
            double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
            const double uv[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) { const int row = 4 * r + q; if (row < m) Ua[row] = uv[r]; }
        }
        // cost of this step on the quadratic model (lane-local partial sums)
        d4 Wz = PF<NCZ>(Lc, Z, zero);
        d4 Wu = PF<NCU>(Luu, dU, zero);
        partial += 0.5 * (Z.x * Wz.x + Z.y * Wz.y + Z.z * Wz.z + Z.w * Wz.w);
        partial += dU.x * (0.5 * Wu.x + lu.x) + dU.y * (0.5 * Wu.y + lu.y)
                 + dU.z * (0.5 * Wu.z + lu.z) + dU.w * (0.5 * Wu.w + lu.w);
        // linearised dynamics
        d4 Zn = PF<NCZ>(Ya, Z, zero);
        Zn = PF<NCU>(Yb, dU, Zn);
        Z = Zn;
        cur = nxt;
    }
    // column sums: lanes c, c+16, c+32, c+48
    partial += __shfl_xor(partial, 16);
    partial += __shfl_xor(partial, 32);
    if (q == 0 && c < n_alpha) cost_pred[(size_t)b * n_alpha + c] = partial;
}

bool forward_mfma_supported(int n, int m, int n_alpha)
{
    return (n + 2 <= 16) && (m <= 8) && (n_alpha <= 16) && n >= 2;
}

hipError_t launch_forward_mfma(Ctx *c, double *U_alpha_dev)
{
    const int n = c->n, m = c->d.m;
    const int ncz = (n + 2 + 3) / 4, ncu = (m + 3) / 4;
    dim3 grid(c->d.batch), block(64);
#define LAUNCH(NCZ, NCU)                                                                               \
    hipLaunchKernelGGL((k_forward_mfma<NCZ, NCU>), grid, block, 0, c->stream, c->L, c->d.T, c->d.n_alpha, \
                       c->rec, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev)
    if (ncu <= 1) { if (ncz <= 2) LAUNCH(2, 1); else if (ncz == 3) LAUNCH(3, 1); else LAUNCH(4, 1); }
    else          { if (ncz <= 2) LAUNCH(2, 2); else if (ncz == 3) LAUNCH(3, 2); else LAUNCH(4, 2); }
#undef LAUNCH
    return hipGetLastError();
}

}  // namespace kpilqr
