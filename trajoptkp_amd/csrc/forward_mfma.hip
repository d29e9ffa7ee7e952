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
//     Yk = [K' ; k']  (rows = state index / alpha, cols = control index)
//     U  = u_nom + P(Yk, Z) = u_nom + K dx + alpha k    (:879, association differs by rounding)
//     dU = clamp(U) - u_nom                              (:883-889)
//     Z+ = P(Ya, Z) + P(Yb, dU),  Ya = [A' 0 0; 0 1 0; 0 0 1], Yb = B'
//     cost: lane-local  sum_r Z_r * P(Lc, Z)_r / 2 + dU_r * (P(Luu, dU)_r / 2 + l_u)   with
//           Lc = [l_xx 0 l_x; 0 0 0; l_x' 0 0]; per-lane partial sums over t, one cross-lane
//           reduction at the end.
#include "mfma_common.h"

namespace kpilqr {

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

// Bounds-checked buffer loads: a lane whose tile element is a structural zero carries an out-of-range
// offset and gets 0 from the hardware (no select on the loaded value -> the loads of step t+1 stay in
// flight behind the whole of step t; see riccati_mfma.hip).
#define OOB 0x7ffffff0
__device__ __forceinline__ double bld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const double *p, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}

template <int NCZ, int NCU>
__device__ __forceinline__ void forward_body(RecLayout L, int T, int n_alpha, const double *__restrict__ rec,
               const double *__restrict__ Kin, const double *__restrict__ kin,
               const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
               const double *__restrict__ alphas, double *__restrict__ cost_pred,
               double *__restrict__ U_alpha)
{
    const int n = L.n, m = L.m;
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;

    // per-lane source BYTE offsets; OOB = structural zero
    int oK[4], ok_[4], oA[4], oB[4], oLc[4], oLuu[4], olu[4], oub[4];
    double oneA[4], lo[4], hi[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        oK[r] = (row < n && c < m) ? 8 * (row * m + c) : OOB;            // K(i=c, p=row) at c + row*m
        ok_[r] = (row == n && c < m) ? 8 * c : OOB;                      // k(i=c)
        oA[r] = (row < n && c < n) ? 8 * L.a(c, row) : OOB;  // A(i=c, p=row)
        oneA[r] = ((row == n && c == n) || (row == n + 1 && c == n + 1)) ? 1.0 : 0.0;
        oB[r] = (row < m && c < n) ? 8 * L.b(c, row) : OOB;  // B(i=c, p=row)
        oLc[r] = (row < n && c < n) ? 8 * (L.off_lxx + row * n + c)
               : (row == n + 1 && c < n) ? 8 * (L.off_lx + c)
               : (c == n + 1 && row < n) ? 8 * (L.off_lx + row) : OOB;
        oLuu[r] = (row < m && c < m) ? 8 * (L.off_luu + row * m + c) : OOB;
        olu[r] = (row < m) ? 8 * (L.off_lu + row) : OOB;
        oub[r] = (row < m) ? 8 * row : OOB;
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

    struct Tiles { d4 YkK, Ykk, Ya, Yb, Lc, Luu, lu, ub; };
    const int rec_bytes = L.rec * 8;
    auto load_tiles = [&](int t, Tiles &s) {
        __amdgpu_buffer_rsrc_t rR = rsrc(rec + ((size_t)b * T + t) * L.stride, rec_bytes);
        __amdgpu_buffer_rsrc_t rK = rsrc(Kin + ((size_t)b * T + t) * m * n, m * n * 8);
        __amdgpu_buffer_rsrc_t rk = rsrc(kin + ((size_t)b * T + t) * m, m * 8);
        __amdgpu_buffer_rsrc_t ru = rsrc(u_nom + ((size_t)b * T + t) * m, m * 8);
        s.YkK.x = bld(rK, oK[0]); s.YkK.y = bld(rK, oK[1]); s.YkK.z = bld(rK, oK[2]); s.YkK.w = bld(rK, oK[3]);
        s.Ykk.x = bld(rk, ok_[0]); s.Ykk.y = bld(rk, ok_[1]); s.Ykk.z = bld(rk, ok_[2]); s.Ykk.w = bld(rk, ok_[3]);
        s.Ya.x = bld(rR, oA[0]); s.Ya.y = bld(rR, oA[1]); s.Ya.z = bld(rR, oA[2]); s.Ya.w = bld(rR, oA[3]);
        s.Yb.x = bld(rR, oB[0]); s.Yb.y = NCU > 1 ? bld(rR, oB[1]) : 0.0;
        s.Yb.z = NCU > 2 ? bld(rR, oB[2]) : 0.0; s.Yb.w = NCU > 3 ? bld(rR, oB[3]) : 0.0;
        s.Lc.x = bld(rR, oLc[0]); s.Lc.y = bld(rR, oLc[1]); s.Lc.z = bld(rR, oLc[2]); s.Lc.w = bld(rR, oLc[3]);
        s.Luu.x = bld(rR, oLuu[0]); s.Luu.y = NCU > 1 ? bld(rR, oLuu[1]) : 0.0;
        s.Luu.z = NCU > 2 ? bld(rR, oLuu[2]) : 0.0; s.Luu.w = NCU > 3 ? bld(rR, oLuu[3]) : 0.0;
        s.lu.x = bld(rR, olu[0]); s.lu.y = NCU > 1 ? bld(rR, olu[1]) : 0.0;
        s.lu.z = NCU > 2 ? bld(rR, olu[2]) : 0.0; s.lu.w = NCU > 3 ? bld(rR, olu[3]) : 0.0;
        s.ub.x = bld(ru, oub[0]); s.ub.y = NCU > 1 ? bld(ru, oub[1]) : 0.0;
        s.ub.z = NCU > 2 ? bld(ru, oub[2]) : 0.0; s.ub.w = NCU > 3 ? bld(ru, oub[3]) : 0.0;
    };
    // Two tile sets (even / odd steps), each re-requested for step t+2 as soon as step t has taken its copy; the time loop is a
    // body of two steps without conditions (a request behind `if (t + 1 < T)` makes the waits that follow conservative), the
    // odd last step peeled.  24 loads per step: a third set would not fit the 6-bit counter.
    Tiles S[2];
    load_tiles(0, S[0]);
    load_tiles(T > 1 ? 1 : 0, S[1]);
    auto step = [&](int t, Tiles &set) __attribute__((always_inline)) {
        const Tiles cur = set;
        const d4 Yk = cur.YkK + cur.Ykk;
        d4 Ya = cur.Ya;
        Ya.x += oneA[0]; Ya.y += oneA[1]; Ya.z += oneA[2]; Ya.w += oneA[3];
        const d4 Yb = cur.Yb, Lc = cur.Lc, Luu = cur.Luu, lu = cur.lu, ub = cur.ub;
        load_tiles(t + 2 < T ? t + 2 : T - 1, set);           // (behind the last steps: a valid record, never used)
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch AHEAD of this step's compute

        // Order of the step: every product is a dependent MFMA chain whose result is usable ~100 cycles after its last
        // issue, so an independent chain follows each one before its consumer:
        //   U chain | A dx chain | clamp (U ready) | B du | l_uu du | Lc Z | cost
        d4 U = PF<NCZ>(Yk, Z, ub);        // u_nom + K dx + alpha k
        d4 Zn = PF<NCZ>(Ya, Z, zero);     // A dx (does not need the controls)
        d4 dU;
        {
            double u;
            u = U.x; if (u > hi[0]) u = hi[0]; if (u < lo[0]) u = lo[0]; U.x = u; dU.x = u - ub.x;
            u = U.y; if (u > hi[1]) u = hi[1]; if (u < lo[1]) u = lo[1]; U.y = u; dU.y = u - ub.y;
            u = U.z; if (u > hi[2]) u = hi[2]; if (u < lo[2]) u = lo[2]; U.z = u; dU.z = u - ub.z;
            u = U.w; if (u > hi[3]) u = hi[3]; if (u < lo[3]) u = lo[3]; U.w = u; dU.w = u - ub.w;
        }
        Zn = PF<NCU>(Yb, dU, Zn);         // + B du: the next state
        d4 Wu = PF<NCU>(Luu, dU, zero);
        d4 Wz = PF<NCZ>(Lc, Z, zero);
        if (U_alpha && c < n_alpha) {
            double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
            const double uv[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) { const int row = 4 * r + q; if (row < m) Ua[row] = uv[r]; }
        }
        // cost of this step on the quadratic model (lane-local partial sums)
        partial += dU.x * (0.5 * Wu.x + lu.x) + dU.y * (0.5 * Wu.y + lu.y)
                 + dU.z * (0.5 * Wu.z + lu.z) + dU.w * (0.5 * Wu.w + lu.w);
        partial += 0.5 * (Z.x * Wz.x + Z.y * Wz.y + Z.z * Wz.z + Z.w * Wz.w);
        Z = Zn;
    };
    int t = 0;
    for (; t + 2 <= T; t += 2) { step(t, S[0]); step(t + 1, S[1]); }
    if (t < T) step(t, S[0]);
    // column sums: lanes c, c+16, c+32, c+48
    partial += __shfl_xor(partial, 16);
    partial += __shfl_xor(partial, 32);
    if (q == 0 && c < n_alpha) cost_pred[(size_t)b * n_alpha + c] = partial;
}

// Shared / exclusive-SIMD entry points (see riccati_mfma.hip: one wave per SIMD when the batch fits).
template <int NCZ, int NCU>
__global__ void __launch_bounds__(64)
k_forward_mfma(RecLayout L, int T, int n_alpha, const double *__restrict__ rec, const double *__restrict__ Kin,
               const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
               const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha)
{
    forward_body<NCZ, NCU>(L, T, n_alpha, rec, Kin, kin, u_nom, ctrl_lim, alphas, cost_pred, U_alpha);
}
template <int NCZ, int NCU>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_forward_mfma_excl(RecLayout L, int T, int n_alpha, const double *__restrict__ rec, const double *__restrict__ Kin,
                    const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                    const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha)
{
    forward_body<NCZ, NCU>(L, T, n_alpha, rec, Kin, kin, u_nom, ctrl_lim, alphas, cost_pred, U_alpha);
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
    const bool excl = c->d.batch <= c->n_simd;
#define LAUNCH(NCZ, NCU)                                                                                         \
    do {                                                                                                         \
        if (excl)                                                                                                \
            hipLaunchKernelGGL((k_forward_mfma_excl<NCZ, NCU>), grid, block, 0, c->stream, c->L, c->d.T,         \
                               c->d.n_alpha, c->rec, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, \
                               U_alpha_dev);                                                                     \
        else                                                                                                     \
            hipLaunchKernelGGL((k_forward_mfma<NCZ, NCU>), grid, block, 0, c->stream, c->L, c->d.T,              \
                               c->d.n_alpha, c->rec, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, \
                               U_alpha_dev);                                                                     \
    } while (0)
    if (ncu <= 1) { if (ncz <= 2) LAUNCH(2, 1); else if (ncz == 3) LAUNCH(3, 1); else LAUNCH(4, 1); }
    else          { if (ncz <= 2) LAUNCH(2, 2); else if (ncz == 3) LAUNCH(3, 2); else LAUNCH(4, 2); }
#undef LAUNCH
    return hipGetLastError();
}

}  // namespace kpilqr
