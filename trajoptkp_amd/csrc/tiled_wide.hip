// tiled_wide.hip -- the tiled backward (a7) and forward (a8) sweeps for control dimensions beyond what tiled_mfma.hip takes:
// 8 < num_ctrl <= 32 in the backward pass (its per-lane LDL' needs the m x m system in registers), 16 < num_ctrl <= 32 in the
// forward pass.  Shapes of the reference's locomotion tasks: the humanoid (TaskConfigs/locomotion/humanoid.yaml:14-15, 21
// actuators, a 27-DoF state with the free root: n = 54) ran on the VALU / LDS kernels of generic.hip before.
//
// Same formulation, tile layout and wave organisation as tiled_mfma.hip (homogeneous coordinate z = [dx; 1], P(Y,X) = Y'X on
// v_mfma_f64_16x16x4_f64, NT wavefronts per trajectory, wave w owning column tile w of the state), with the control block
// spanning MT = 1 or 2 tiles: Fu is NT x MT tiles, Quu MT x MT, Quz and X MT x NT.  What changes in substance:
//   * Quu = l_uu + Fu' (V Fu): every wave leaves its row tile of Tu = V Fu in LDS and the MT^2 output tiles are dealt to the
//     waves (no per-wave partial images: they would not fit beside V and Fz at four state tiles);
//   * (Quu + lambda I)^-1 is carried along the sweep as MT x MT tiles and refreshed by Newton-Schulz steps on the matrix core
//     like the narrow kernels do; the factorisation behind it -- first step, every checked step (CheckMatrixPD,
//     iLQR.cpp:587-595, 659-670), re-seeds -- is a COOPERATIVE unpivoted LDL' of the dense m x m image in LDS by one
//     wavefront (the lanes share the rank-1 updates), the explicit inverse column by column (the reference forms the
//     explicit inverse too, :597-600), handed to the other waves through LDS;
//   * an indefinite Quu + lambda I on an unchecked step goes through Eigen's pivoted LDLT restated (kp_slow_ldlt_inverse), as
//     everywhere else.
// A, B and the cost derivatives come from the step records (no a4 / a6 fusion in this form).
//
// Reference: iLQR::BackwardsPassQuuRegularisation + CheckMatrixPD, src/Optimiser/iLQR.cpp:535-670; control law / clamp of
// iLQR::ForwardsPassParallel, :876-890.
#include <cstdlib>
#include "mfma_common.h"

#ifndef KP_WIDE_SPREAD
#define KP_WIDE_SPREAD 1            // 0: the wide-control sweeps' requests as one block per step: A/B builds
#endif

namespace kpilqr {

typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
#define WMFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define OOBW 0x7ffffff0
#define WTILE 256
#define WTPAD 272                                    // 16 x 17: padded row-major tile
#define WDS 33                                       // row stride of the dense m x m images (m <= 32)

__device__ __forceinline__ double wbld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ d4 wlds(const double *t, int lane)
{
    d4 v; v.x = t[lane]; v.y = t[64 + lane]; v.z = t[128 + lane]; v.w = t[192 + lane];
    return v;
}
__device__ __forceinline__ void wsts(double *t, int lane, const d4 &v)
{
    t[lane] = v.x; t[64 + lane] = v.y; t[128 + lane] = v.z; t[192 + lane] = v.w;
}
__device__ __forceinline__ d4 WPn(const d4 &Y, const d4 &X, d4 acc, int nc)       // acc + Y'X over nc 4-row chunks (wave-uniform)
{
    acc = WMFMA(Y.x, X.x, acc);
    if (nc > 1) acc = WMFMA(Y.y, X.y, acc);
    if (nc > 2) acc = WMFMA(Y.z, X.z, acc);
    if (nc > 3) acc = WMFMA(Y.w, X.w, acc);
    return acc;
}
// LDS address of element (i, j) of an MT x MT (or any) grid of accumulator-layout tiles, tile (a, b) at (a * MTc + b) * 256
__device__ __forceinline__ int wtile_addr(int i, int j, int MTc)
{
    return ((i >> 4) * MTc + (j >> 4)) * WTILE + ((i & 15) >> 2) * 64 + (j & 15) + 16 * (i & 3);
}
__device__ __forceinline__ void wave_lds_sync()
{
    // one wavefront: its LDS instructions execute in order; the compiler must not move accesses across, and the data of the
    // earlier ones must have landed before a later one of ANOTHER lane reads it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cooperative unpivoted LDL' of the m x m matrix whose tiles sit at qt (MT x MT accumulator-layout tiles) and its explicit
// inverse written as tiles to invt, by ONE wavefront.  A: dense work image [m][WDS].  Returns false when a pivot is not
// positive (the caller then reports the PD failure or takes the pivoted slow path; invt is not meaningful).
__device__ static __attribute__((noinline)) bool coop_ldl_inverse(const double *qt, double *invt, double *A, int m, int MTc, int lane)
{
    for (int idx = lane; idx < m * m; idx += 64) {
        const int i = idx / m, j = idx - i * m;
        A[i * WDS + j] = qt[wtile_addr(i, j, MTc)];
    }
    wave_lds_sync();
    bool pos = true;
    for (int j = 0; j < m; j++) {
        const double dj = A[j * WDS + j];
        pos = pos && (dj > 0.0);
        const double rj = kp_rcp(dj);
        const int cnt = m - j - 1;
        // trailing update with the unscaled column: A(i,k) -= A(i,j) A(k,j) / d_j for j < k <= i
        for (int p = lane; p < cnt * cnt; p += 64) {
            const int ii = p / cnt, kk = p - ii * cnt;
            if (kk <= ii) {
                const int i = j + 1 + ii, k = j + 1 + kk;
                A[i * WDS + k] -= (A[i * WDS + j] * rj) * A[k * WDS + j];
            }
        }
        wave_lds_sync();
        if (lane < cnt) A[(j + 1 + lane) * WDS + j] *= rj;           // L(i,j)
        wave_lds_sync();
    }
    if (!pos) return false;
    // inverse, one column per lane (c < m): L y = e_c, y /= D, L' x = y; the column lives in the tile image itself
    if (lane < m) {
        const int cc = lane;
        for (int i = 0; i < m; i++) {
            double s = (i == cc) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= A[i * WDS + k] * invt[wtile_addr(k, cc, MTc)];
            invt[wtile_addr(i, cc, MTc)] = s;
        }
        for (int i = 0; i < m; i++) invt[wtile_addr(i, cc, MTc)] *= kp_rcp(A[i * WDS + i]);
        for (int i = m - 1; i >= 0; i--) {
            double s = invt[wtile_addr(i, cc, MTc)];
            for (int k = i + 1; k < m; k++) s -= A[k * WDS + i] * invt[wtile_addr(k, cc, MTc)];
            invt[wtile_addr(i, cc, MTc)] = s;
        }
    }
    wave_lds_sync();
    return true;
}

struct WideSrc { int n, m; int off_A, off_B, off_lxx, off_lx, off_luu, off_lu; };

__device__ __forceinline__ d4 wld_Lzz(__amdgpu_buffer_rsrc_t rs, const WideSrc &S, int ti, int tj, int q, int c)
{
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 16 * ti + 4 * r + q, col = 16 * tj + c, n = S.n;
        const int off = (row < n && col < n) ? 8 * (S.off_lxx + row * n + col)
                      : (col == n && row < n) ? 8 * (S.off_lx + row)
                      : (row == n && col < n) ? 8 * (S.off_lx + col) : OOBW;
        v[r] = wbld(rs, off);
    }
    d4 o = {v[0], v[1], v[2], v[3]};
    return o;
}
__device__ __forceinline__ d4 wld_Fz(__amdgpu_buffer_rsrc_t rs, const WideSrc &S, int ti, int tj, int q, int c)
{
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 16 * ti + 4 * r + q, col = 16 * tj + c, n = S.n;
        v[r] = wbld(rs, (row < n && col < n) ? 8 * (S.off_A + col * n + row) : OOBW);
    }
    d4 o = {v[0], v[1], v[2], v[3]};
    return o;
}
// B(row tile ti, control tile tj): element (16 ti + 4r+q, 16 tj + c)
__device__ __forceinline__ d4 wld_Fu(__amdgpu_buffer_rsrc_t rs, const WideSrc &S, int ti, int tj, int q, int c)
{
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 16 * ti + 4 * r + q, col = 16 * tj + c;
        v[r] = wbld(rs, (row < S.n && col < S.m) ? 8 * (S.off_B + col * S.n + row) : OOBW);
    }
    d4 o = {v[0], v[1], v[2], v[3]};
    return o;
}
// Luz(control tile tj, state tile tw): l_u in column n
__device__ __forceinline__ d4 wld_Luz(__amdgpu_buffer_rsrc_t rs, const WideSrc &S, int tj, int tw, int q, int c)
{
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 16 * tj + 4 * r + q, col = 16 * tw + c;
        v[r] = wbld(rs, (row < S.m && col == S.n) ? 8 * (S.off_lu + row) : OOBW);
    }
    d4 o = {v[0], v[1], v[2], v[3]};
    return o;
}
__device__ __forceinline__ d4 wld_Luu(__amdgpu_buffer_rsrc_t rs, const WideSrc &S, int ta, int tb, int q, int c)
{
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 16 * ta + 4 * r + q, col = 16 * tb + c;
        v[r] = wbld(rs, (row < S.m && col < S.m) ? 8 * (S.off_luu + row * S.m + col) : OOBW);
    }
    d4 o = {v[0], v[1], v[2], v[3]};
    return o;
}

// LDS map (doubles) of the backward kernel
template <int NT, int MT> struct WideLds {
    static constexpr int NZZ = NT * NT;
    static constexpr int V = 0;
    static constexpr int F = V + NZZ * WTILE;
    static constexpr int TneedA = NZZ * WTPAD;                                   // the transposing scratch of phases E / F
    static constexpr int TneedB = (NT * MT + 2 * MT * MT) * WTILE;               // Tu, Quu + lambda I, inverse tiles (phases BC .. D)
    static constexpr int Tsz = TneedA > TneedB ? TneedA : TneedB;
    static constexpr int T = F + NZZ * WTILE;
    static constexpr int Fu = T + Tsz;
    static constexpr int X = Fu + NT * MT * WTILE;
    static constexpr int Flags = X + MT * NT * WTILE;
    // dense work images of the factorisation / slow path: A [32][33], a [32*32], x [32*32], temp [32], tr [32 ints]
    static constexpr int DenseSz = 32 * WDS + 2 * 1024 + 64;
    static constexpr bool DenseInF = NZZ * WTILE >= DenseSz;                     // four state tiles: Fz is free in phase D
    static constexpr int Dense = DenseInF ? F : Flags + 8;
    static constexpr int Total = DenseInF ? Flags + 8 : Dense + DenseSz;
};

template <int NT, int MT>
__global__ void __launch_bounds__(64 * NT)
k_backward_tiled_wide(RecLayout L, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                      int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                      double *__restrict__ delta_J, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    using M_ = WideLds<NT, MT>;
    const int n = L.n, m = L.m, nz = n + 1;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const double lam = lambda[b];
    double *bufV = sh + M_::V, *bufF = sh + M_::F, *bufT = sh + M_::T, *bufFu = sh + M_::Fu, *bufX = sh + M_::X;
    double *bufTu = bufT, *bufQ = bufTu + NT * MT * WTILE, *bufInv = bufQ + MT * MT * WTILE;      // alias the scratch of phases E / F
    int *flags = (int *)(sh + M_::Flags);
    double *dense = sh + M_::Dense;
    auto nchunk = [&](int kt) { const int rows = nz - 16 * kt; return rows >= 16 ? 4 : (rows + 3) / 4; };
    const int ncl = nchunk(NT - 1);
    auto mch = [&](int j) { const int rows = m - 16 * j; return rows >= 16 ? 4 : rows <= 0 ? 0 : (rows + 3) / 4; };
    WideSrc S = {n, m, L.off_A, L.off_B, L.off_lxx, L.off_lx, L.off_luu, L.off_lu};
    const double *R0 = rec + (size_t)b * T * L.stride;
    const int rec_bytes = L.rec * 8;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const int tn = n >> 4, cn = n & 15;
    const bool lane_nn = (c == cn) && (q == (cn & 3));
    const int reg_nn = cn >> 2;
    d4 nn_keep;
    nn_keep.x = (lane_nn && reg_nn == 0) ? 0.0 : 1.0; nn_keep.y = (lane_nn && reg_nn == 1) ? 0.0 : 1.0;
    nn_keep.z = (lane_nn && reg_nn == 2) ? 0.0 : 1.0; nn_keep.w = (lane_nn && reg_nn == 3) ? 0.0 : 1.0;
    const d4 nn_one = 1.0 - nn_keep;
    // identity of the control block, per diagonal tile a: element (16a + 4r+q, 16a + c) with 4r+q == c and 16a + c < m
    auto eye = [&](int a) {
        d4 e;
        e.x = (q == c && 16 * a + c < m) ? 1.0 : 0.0; e.y = (4 + q == c && 16 * a + c < m) ? 1.0 : 0.0;
        e.z = (8 + q == c && 16 * a + c < m) ? 1.0 : 0.0; e.w = (12 + q == c && 16 * a + c < m) ? 1.0 : 0.0;
        return e;
    };
    auto rsrc_of = [&](int t) { return __builtin_amdgcn_make_buffer_rsrc((void *)(R0 + (size_t)t * L.stride), 0, rec_bytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t rnone = __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000);

    // source tiles of the current step: Fz(k,w), Lzz(k,w), Fu(w,j), Luz(j,w); Luu(j1,j2) of the Quu tiles this wave forms
    d4 pF[NT], pL[NT], pFu[MT], pLuz[MT], pLuu[(MT * MT + NT - 1) / NT];
    auto load_src = [&](__amdgpu_buffer_rsrc_t rs) {
#pragma unroll
        for (int k = 0; k < NT; k++) { pF[k] = wld_Fz(rs, S, k, w, q, c); pL[k] = wld_Lzz(rs, S, k, w, q, c); }
#pragma unroll
        for (int j = 0; j < MT; j++) { pFu[j] = wld_Fu(rs, S, w, j, q, c); pLuz[j] = wld_Luz(rs, S, j, w, q, c); }
#pragma unroll
        for (int s_ = 0; s_ < (MT * MT + NT - 1) / NT; s_++) {
            const int tq = w + s_ * NT;
            pLuu[s_] = (tq < MT * MT) ? wld_Luu(rs, S, tq / MT, tq % MT, q, c) : zero;
        }
    };
    load_src(rsrc_of(T - 1));
    // V' <- Lzz(T-1)   (iLQR.cpp:537-539)
#pragma unroll
    for (int k = 0; k < NT; k++) wsts(bufV + (k * NT + w) * WTILE, lane, pL[k]);
    if (threadIdx.x == 0) flags[0] = 0;

    int pd_counter = 0, fail = 0;
    double dJ = 0.0;
    d4 Xinv[MT * MT];
#pragma unroll
    for (int i = 0; i < MT * MT; i++) Xinv[i] = zero;
    bool haveX = false;

    for (int t = T - 1; t >= 0; t--) {
        pd_counter++;
        const bool check_pd = pd_counter >= pd_stride;
        const bool more = t > 0;
        // ---- A: stage Fz(:,w) (+ the homogeneous 1) and Fu(w,:) ---------------------------------------------------------
        d4 cL[NT], cLuz[MT], cLuu[(MT * MT + NT - 1) / NT];
#pragma unroll
        for (int k = 0; k < NT; k++) {
            d4 f = pF[k];
            if (k == tn && w == tn) f = f + nn_one;                    // Fz(n,n) = 1
            wsts(bufF + (k * NT + w) * WTILE, lane, f);
            cL[k] = pL[k];
        }
#pragma unroll
        for (int j = 0; j < MT; j++) { wsts(bufFu + (w * MT + j) * WTILE, lane, pFu[j]); cLuz[j] = pLuz[j]; }
#pragma unroll
        for (int s_ = 0; s_ < (MT * MT + NT - 1) / NT; s_++) cLuu[s_] = pLuu[s_];
        __builtin_amdgcn_sched_barrier(0);
        // (round 5, late: the ~50 requests of a step go out in groups behind the product groups of phase BC instead of as one block
        // in front of this barrier -- the waves of a trajectory share one address unit)
        const __amdgpu_buffer_rsrc_t rn = more ? rsrc_of(t - 1) : rnone;
        if constexpr (!KP_WIDE_SPREAD) load_src(rn);                   // single-buffered: requested right behind their last use
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // ---- BC: Tz(:,w), Tu(w,:), Quz(:,w), Qzz(:,w) ---------------------------------------------------------------------
        d4 Fc[NT], Tz[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) Fc[k] = wlds(bufF + (k * NT + w) * WTILE, lane);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            d4 acc = zero;
#pragma unroll
            for (int k = 0; k < NT; k++) acc = WPn(wlds(bufV + (k * NT + i) * WTILE, lane), Fc[k], acc, k < NT - 1 ? 4 : ncl);
            Tz[i] = acc;
            if constexpr (KP_WIDE_SPREAD) { pF[i] = wld_Fz(rn, S, i, w, q, c); pL[i] = wld_Lzz(rn, S, i, w, q, c); }
        }
#pragma unroll
        for (int j = 0; j < MT; j++) {
            d4 Tu = zero;
#pragma unroll
            for (int k = 0; k < NT; k++) Tu = WPn(wlds(bufV + (k * NT + w) * WTILE, lane), wlds(bufFu + (k * MT + j) * WTILE, lane), Tu, k < NT - 1 ? 4 : ncl);
            wsts(bufTu + (w * MT + j) * WTILE, lane, Tu);
            if constexpr (KP_WIDE_SPREAD) { pFu[j] = wld_Fu(rn, S, w, j, q, c); pLuz[j] = wld_Luz(rn, S, j, w, q, c); }
        }
        d4 Quzw[MT];
#pragma unroll
        for (int j = 0; j < MT; j++) {
            d4 acc = cLuz[j];
#pragma unroll
            for (int k = 0; k < NT; k++) acc = WPn(wlds(bufFu + (k * MT + j) * WTILE, lane), Tz[k], acc, k < NT - 1 ? 4 : ncl);
            Quzw[j] = acc;
        }
        if constexpr (KP_WIDE_SPREAD) {
#pragma unroll
            for (int s_ = 0; s_ < (MT * MT + NT - 1) / NT; s_++) {
                const int tq = w + s_ * NT;
                pLuu[s_] = (tq < MT * MT) ? wld_Luu(rn, S, tq / MT, tq % MT, q, c) : zero;
            }
        }
        d4 Qzz[NT];
#pragma unroll
        for (int i = 0; i < NT; i++) {
            d4 acc = cL[i];
#pragma unroll
            for (int k = 0; k < NT; k++) acc = WPn(wlds(bufF + (k * NT + i) * WTILE, lane), Tz[k], acc, k < NT - 1 ? 4 : ncl);
            Qzz[i] = acc;
        }
        __syncthreads();
        // ---- D1: the Quu tiles dealt to this wave: l_uu + sum_k Fu(k,j1)' Tu(k,j2), + lambda I ---------------------------------
#pragma unroll
        for (int s_ = 0; s_ < (MT * MT + NT - 1) / NT; s_++) {
            const int tq = w + s_ * NT;
            if (tq < MT * MT) {
                const int j1 = tq / MT, j2 = tq % MT;
                d4 acc = cLuu[s_];
#pragma unroll
                for (int k = 0; k < NT; k++) acc = WPn(wlds(bufFu + (k * MT + j1) * WTILE, lane), wlds(bufTu + (k * MT + j2) * WTILE, lane), acc, k < NT - 1 ? 4 : ncl);
                if (j1 == j2) acc = acc + lam * eye(j1);
                wsts(bufQ + tq * WTILE, lane, acc);
            }
        }
        __syncthreads();
        // ---- D2: (Quu + lambda I)^-1 by every wave (identical inputs: block-uniform decisions), X, K, k, G ----------------------
        d4 Qr[MT * MT];
#pragma unroll
        for (int i = 0; i < MT * MT; i++) Qr[i] = wlds(bufQ + i * WTILE, lane);
        bool done = false;
        if (haveX && !check_pd) {
            // Newton-Schulz: R = I - Q X, X <- X + X R; the count chosen from the measured residual (kp_inverse_refresh)
            d4 R[MT * MT];
            auto residual = [&]() -> double {
                double rmax = 0.0;
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int bb = 0; bb < MT; bb++) {
                        d4 acc = zero;
#pragma unroll
                        for (int j = 0; j < MT; j++) acc = WPn(Qr[j * MT + a], Xinv[j * MT + bb], acc, mch(j));
                        d4 r_ = (a == bb ? eye(a) : zero) - acc;
                        R[a * MT + bb] = r_;
                        rmax = fmax(rmax, fmax(fmax(fabs(r_.x), fabs(r_.y)), fmax(fabs(r_.z), fabs(r_.w))));
                    }
                return rmax;
            };
            auto update = [&]() {
                d4 Xn[MT * MT];
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int bb = 0; bb < MT; bb++) {
                        d4 acc = Xinv[a * MT + bb];
#pragma unroll
                        for (int j = 0; j < MT; j++) acc = WPn(Xinv[j * MT + a], R[j * MT + bb], acc, mch(j));
                        Xn[a * MT + bb] = acc;
                    }
#pragma unroll
                for (int i = 0; i < MT * MT; i++) Xinv[i] = Xn[i];
            };
            const double e = (double)m * residual();
            if (__builtin_amdgcn_ballot_w64(!(e < 0.11)) == 0) {
                const int iters = (__builtin_amdgcn_ballot_w64(e >= 1.3e-2) != 0) ? 4
                                : (__builtin_amdgcn_ballot_w64(e >= 1.7e-4) != 0) ? 3
                                : (__builtin_amdgcn_ballot_w64(e >= 3.0e-8) != 0) ? 2 : 1;
                update();
                for (int it = 1; it < iters; it++) { (void)residual(); update(); }
                done = true;
            }
        }
        if (!done) {
            // factorisation by wave 0 (every wave takes this branch: the decision above is block-uniform)
            if (w == 0) {
                const bool pos = coop_ldl_inverse(bufQ, bufInv, dense, m, MT, lane);
                if (lane == 0) flags[1] = pos ? 0 : 1;
            }
            __syncthreads();
            const bool pos = flags[1] == 0;
            if (check_pd) {                       // CheckMatrixPD every pd_stride steps   :587-595
                if (!pos) { fail = t + 1; break; }
                pd_counter = 0;
            }
            if (!pos) {
                // indefinite on an unchecked step: Eigen's pivoted LDLT + explicit inverse (:597-604), by one lane
                if (threadIdx.x == 0) {
                    double *Am = dense, *wa = dense + 32 * WDS, *wx = wa + 1024, *wt = wx + 1024;
                    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Am[i * WDS + j] = bufQ[wtile_addr(i, j, MT)];
                    kp_slow_ldlt_inverse(m, Am, WDS, wa, wx, wt, (int *)(wt + 32));
                    for (int i = 0; i < MT * 16; i++) for (int j = 0; j < MT * 16; j++)
                        bufInv[wtile_addr(i, j, MT)] = (i < m && j < m) ? wx[i + j * m] : 0.0;
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < MT * MT; i++) Xinv[i] = wlds(bufInv + i * WTILE, lane);
            if (pos) {
                // entries outside the m x m block of the image were never written by the cooperative inverse: mask them
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int bb = 0; bb < MT; bb++) {
                        d4 &x_ = Xinv[a * MT + bb];
                        const bool cv = 16 * bb + c < m;
                        x_.x = (cv && 16 * a + q < m) ? x_.x : 0.0; x_.y = (cv && 16 * a + 4 + q < m) ? x_.y : 0.0;
                        x_.z = (cv && 16 * a + 8 + q < m) ? x_.z : 0.0; x_.w = (cv && 16 * a + 12 + q < m) ? x_.w : 0.0;
                    }
            }
            haveX = pos;
        }
        d4 X[MT], G[MT];
#pragma unroll
        for (int j = 0; j < MT; j++) {
            d4 acc = zero;
#pragma unroll
            for (int j2 = 0; j2 < MT; j2++) acc = WPn(Xinv[j2 * MT + j], Quzw[j2], acc, mch(j2));
            X[j] = acc;
        }
        {
            const int col = 16 * w + c;
            __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void *)(Kout + ((size_t)b * T + t) * m * n), 0, m * n * 8, 0x00020000);
            __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(kout + ((size_t)b * T + t) * m), 0, m * 8, 0x00020000);
#pragma unroll
            for (int j = 0; j < MT; j++) {
                const double xv[4] = {X[j].x, X[j].y, X[j].z, X[j].w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * j + 4 * r + q;
                    const double kv = -xv[r];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2w, kv), rK, (row < m && col < n) ? 8 * (row + col * m) : OOBW, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2w, kv), rk, (row < m && col == n) ? 8 * row : OOBW, 0, 0);
                    if (col == n) dJ -= lam * (xv[r] * xv[r]);               // delta_J -= lambda k'k (:612-613)
                }
                // G = (Quu + 2 lambda I) K' = -(Quz + lambda X) because (Quu + lambda I) X = Quz
                G[j].x = -__builtin_fma(lam, X[j].x, Quzw[j].x); G[j].y = -__builtin_fma(lam, X[j].y, Quzw[j].y);
                G[j].z = -__builtin_fma(lam, X[j].z, Quzw[j].z); G[j].w = -__builtin_fma(lam, X[j].w, Quzw[j].w);
                wsts(bufX + (j * NT + w) * WTILE, lane, X[j]);
            }
        }
        __syncthreads();
        // ---- E: acc(i,w) = Qzz(i,w) + sum_j X(j,i)' G(j,w) -> bufT --------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NT; i++) {
#pragma unroll
            for (int j = 0; j < MT; j++) Qzz[i] = WPn(wlds(bufX + (j * NT + i) * WTILE, lane), G[j], Qzz[i], mch(j));
            double *pw = bufT + (i * NT + w) * WTPAD + q * 17 + c;
            pw[0] = Qzz[i].x; pw[4 * 17] = Qzz[i].y; pw[8 * 17] = Qzz[i].z; pw[12 * 17] = Qzz[i].w;
        }
        __syncthreads();
        // ---- F: V'(i,w) = (acc(i,w) + acc(w,i)')/2   (:610) ---------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const double *pt = bufT + (w * NT + i) * WTPAD + c * 17 + q;
            d4 at;
            at.x = pt[0]; at.y = pt[4]; at.z = pt[8]; at.w = pt[12];
            d4 na = 0.5 * (Qzz[i] + at);
            if (i == tn && w == tn) na = na * nn_keep;
            wsts(bufV + (i * NT + w) * WTILE, lane, na);
        }
        __syncthreads();                                   // bufT is Tu / Quu / inverse again in the next step's phase BC
    }
    dJ += __shfl_xor(dJ, 16);
    dJ += __shfl_xor(dJ, 32);
    if (w == tn && lane_nn) delta_J[b] = dJ;
    if (threadIdx.x == 0) status[b] = fail;
}

static int wide_nt(int n, int nt_min)
{
    int nt = (n + 1 + 15) / 16;
    if (nt < 2) nt = 2;
    if (nt_min > nt) nt = nt_min;
    return nt;
}

bool backward_wide_supported(int n, int m, int nt_min)
{
    const int nt = wide_nt(n, nt_min);
    return nt >= 2 && nt <= 4 && m > 8 && m <= 32;
}

template <int NT, int MT>
static hipError_t launch_bw(Ctx *c, int pd_stride)
{
    const size_t lds = sizeof(double) * (size_t)WideLds<NT, MT>::Total;
    hipError_t e = hipFuncSetAttribute((const void *)k_backward_tiled_wide<NT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_backward_tiled_wide<NT, MT>), dim3(c->d.batch), dim3(64 * NT), lds, c->stream, c->L, c->d.T, c->rec, c->lambda,
                       pd_stride, c->K, c->k, c->delta_J, c->status);
    return hipGetLastError();
}

hipError_t launch_backward_wide(Ctx *c, int pd_stride)
{
    const int nt = wide_nt(c->n, c->tune.tiled_nt_min), mt = c->d.m > 16 ? 2 : 1;
    if (mt == 1) { if (nt == 2) return launch_bw<2, 1>(c, pd_stride); if (nt == 3) return launch_bw<3, 1>(c, pd_stride); if (nt == 4) return launch_bw<4, 1>(c, pd_stride); }
    else { if (nt == 2) return launch_bw<2, 2>(c, pd_stride); if (nt == 3) return launch_bw<3, 2>(c, pd_stride); if (nt == 4) return launch_bw<4, 2>(c, pd_stride); }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Forward pass for 16 < num_ctrl <= 32: k_forward_tiled's plain form with the control block over two tiles.  Wave wi owns row
// tile wi of Z+ and of Lc Z and forms its slice of the control law for BOTH control tiles; every wave clamps the same U.
template <int NT, int MT>
__global__ void __launch_bounds__(64 * NT)
k_forward_tiled_wide(RecLayout L, int T, int n_alpha, const double *__restrict__ rec, const double *__restrict__ Kin,
                     const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                     const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha)
{
    extern __shared__ __attribute__((aligned(16))) double fsh[];
    double (*zbuf)[NT * WTILE] = (double (*)[NT * WTILE])fsh;               // [2][NT * TILE]
    double *upart = fsh + 2 * NT * WTILE;                                    // [NT][MT] tiles: per-wave partials of K dx + alpha k
    double *red = upart + NT * MT * WTILE;                                   // [NT * 64]
    const int n = L.n, m = L.m, nz2 = n + 2;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    auto nchunk = [&](int kt) { const int rows = nz2 - 16 * kt; return rows >= 16 ? 4 : (rows + 3) / 4; };
    const int ncl = nchunk(NT - 1);
    const int ncw = nchunk(wi);
    auto mch = [&](int j) { const int rows = m - 16 * j; return rows >= 16 ? 4 : rows <= 0 ? 0 : (rows + 3) / 4; };

    int oKw[MT][4], okw[MT][4], oA[NT][4], oLc[NT][4], oB[MT][4], oLuu[MT][MT][4], olu[MT][4], oub[MT][4];
    double oneT[4], lo[MT][4], hi[MT][4];
    const int tnz = n >> 4;
    const int o = 16 * wi + c;
#pragma unroll
    for (int k = 0; k < NT; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pp = 16 * k + 4 * r + q;
            oA[k][r] = (pp < n && o < n) ? 8 * L.a(o, pp) : OOBW;
            oLc[k][r] = (pp < n && o < n) ? 8 * (L.off_lxx + pp * n + o)
                      : (pp == n + 1 && o < n) ? 8 * (L.off_lx + o)
                      : (o == n + 1 && pp < n) ? 8 * (L.off_lx + pp) : OOBW;
        }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        const int pt = 16 * tnz + row;
        oneT[r] = ((pt == n && o == n) || (pt == n + 1 && o == n + 1)) ? 1.0 : 0.0;
        const int pw = 16 * wi + row;                                   // this wave's slice of the state index
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const int cu = 16 * j + c, ru = 16 * j + row;               // control index as a column / as a row
            oKw[j][r] = (pw < n && cu < m) ? 8 * (pw * m + cu) : OOBW;
            okw[j][r] = (pw == n && cu < m) ? 8 * cu : OOBW;
            oB[j][r] = (ru < m && o < n) ? 8 * L.b(o, ru) : OOBW;
            olu[j][r] = (ru < m) ? 8 * (L.off_lu + ru) : OOBW;
            oub[j][r] = (ru < m) ? 8 * ru : OOBW;
            lo[j][r] = (ru < m) ? ctrl_lim[2 * ru] : -1.0e300;
            hi[j][r] = (ru < m) ? ctrl_lim[2 * ru + 1] : 1.0e300;
#pragma unroll
            for (int j2 = 0; j2 < MT; j2++) {
                const int cu2 = 16 * j2 + c;
                oLuu[j][j2][r] = (ru < m && cu2 < m) ? 8 * (L.off_luu + ru * m + cu2) : OOBW;     // l_uu(ru, cu2): tile (j, j2)
            }
        }
    }
    const double my_alpha = (c < n_alpha) ? alphas[c] : 0.0;
    d4 Zi;
    {
        double zr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wi + 4 * r + q;
            zr[r] = (row == n) ? my_alpha : (row == n + 1) ? 1.0 : 0.0;
        }
        Zi.x = zr[0]; Zi.y = zr[1]; Zi.z = zr[2]; Zi.w = zr[3];
        wsts(zbuf[0] + wi * WTILE, lane, Zi);
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double partial = 0.0;
    struct Tiles { d4 Ykw[MT], Ya[NT], Lc[NT], Yb[MT], Luu[MT][MT], lu[MT], ub[MT]; };
    const int rec_bytes = L.rec * 8;
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, const int *off) -> d4 {
        d4 v; v.x = wbld(rs, off[0]); v.y = wbld(rs, off[1]); v.z = wbld(rs, off[2]); v.w = wbld(rs, off[3]);
        return v;
    };
    auto rs_of = [&](const double *base, size_t step_elems, int t, int bytes) {
        const bool ok = t < T;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + ((size_t)b * T + (ok ? t : 0)) * step_elems), 0, ok ? bytes : 0, 0x00020000);
    };
    Tiles cur;
    // Requests (round 5, late; as in tiled_mfma.hip's k_forward_tiled): every group of tiles is re-requested for step t+1 right behind
    // the products that read it instead of in one block of ~90 requests at the end of the step (the waves of a trajectory share one
    // address unit), k rides in ONE request per control tile (the lanes of row n) and joins the gain operand at its use -- summed
    // behind the request it made every step wait for the request -- and l_uu, l_u are requested by the wave that scores the controls.
    const bool k_here = tnz == wi;
    int okn[MT];
    double kmask[4], kv[MT];
#pragma unroll
    for (int j = 0; j < MT; j++) { okn[j] = (k_here && q == (n & 3) && 16 * j + c < m) ? 8 * (16 * j + c) : OOBW; kv[j] = 0.0; }
#pragma unroll
    for (int r = 0; r < 4; r++) kmask[r] = (r == ((n & 15) >> 2)) ? 1.0 : 0.0;
    auto load_all = [&](int t) {
        const __amdgpu_buffer_rsrc_t rR = rs_of(rec, L.stride, t, rec_bytes), rK = rs_of(Kin, (size_t)m * n, t, m * n * 8);
        const __amdgpu_buffer_rsrc_t rk = rs_of(kin, m, t, m * 8), ru = rs_of(u_nom, m, t, m * 8);
#pragma unroll
        for (int j = 0; j < MT; j++) {
            if constexpr (KP_WIDE_SPREAD) { cur.Ykw[j] = ld4(rK, oKw[j]); kv[j] = wbld(rk, okn[j]); }
            else cur.Ykw[j] = ld4(rK, oKw[j]) + ld4(rk, okw[j]);
            cur.Yb[j] = ld4(rR, oB[j]); cur.lu[j] = ld4(rR, olu[j]); cur.ub[j] = ld4(ru, oub[j]);
#pragma unroll
            for (int j2 = 0; j2 < MT; j2++) cur.Luu[j][j2] = ld4(rR, oLuu[j][j2]);
        }
#pragma unroll
        for (int k = 0; k < NT; k++) { cur.Ya[k] = ld4(rR, oA[k]); cur.Lc[k] = ld4(rR, oLc[k]); }
    };
    load_all(0);
    __syncthreads();

    for (int t = 0; t < T; t++) {
        const double *zc = zbuf[t & 1];
        double *zn = zbuf[(t + 1) & 1];
        const __amdgpu_buffer_rsrc_t rRn = rs_of(rec, L.stride, t + 1, rec_bytes), rKn = rs_of(Kin, (size_t)m * n, t + 1, m * n * 8);
        const __amdgpu_buffer_rsrc_t rkn = rs_of(kin, m, t + 1, m * 8), run = rs_of(u_nom, m, t + 1, m * 8);
        // ---- this wave's slice of K dx + alpha k, both control tiles ---------------------------------------------------------
#pragma unroll
        for (int j = 0; j < MT; j++) {
            if constexpr (KP_WIDE_SPREAD) {
                d4 Yk = cur.Ykw[j];
                Yk.x = __builtin_fma(kmask[0], kv[j], Yk.x); Yk.y = __builtin_fma(kmask[1], kv[j], Yk.y);
                Yk.z = __builtin_fma(kmask[2], kv[j], Yk.z); Yk.w = __builtin_fma(kmask[3], kv[j], Yk.w);
                wsts(upart + (wi * MT + j) * WTILE, lane, WPn(Yk, Zi, zero, ncw));
                cur.Ykw[j] = ld4(rKn, oKw[j]); kv[j] = wbld(rkn, okn[j]);
            } else
            wsts(upart + (wi * MT + j) * WTILE, lane, WPn(cur.Ykw[j], Zi, zero, ncw));
        }
        __syncthreads();
        // ---- control law + clamp (every wave; :876-890) ----------------------------------------------------------------------
        d4 U[MT], dU[MT];
        d4 Zk[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) Zk[k] = wlds(zc + k * WTILE, lane);
#pragma unroll
        for (int j = 0; j < MT; j++) {
            d4 u = cur.ub[j];
#pragma unroll
            for (int k = 0; k < NT; k++) u = u + wlds(upart + (k * MT + j) * WTILE, lane);
            double v;
            v = u.x; if (v > hi[j][0]) v = hi[j][0]; if (v < lo[j][0]) v = lo[j][0]; u.x = v;
            v = u.y; if (v > hi[j][1]) v = hi[j][1]; if (v < lo[j][1]) v = lo[j][1]; u.y = v;
            v = u.z; if (v > hi[j][2]) v = hi[j][2]; if (v < lo[j][2]) v = lo[j][2]; u.z = v;
            v = u.w; if (v > hi[j][3]) v = hi[j][3]; if (v < lo[j][3]) v = lo[j][3]; u.w = v;
            U[j] = u; dU[j] = u - cur.ub[j];
            if constexpr (KP_WIDE_SPREAD) cur.ub[j] = ld4(run, oub[j]);
        }
        if (wi == NT - 1) {                        // control cost and U_alpha by the LAST wave (its row tile is the shortest)
            if (U_alpha && c < n_alpha) {
                double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
#pragma unroll
                for (int j = 0; j < MT; j++) {
                    const double uv[4] = {U[j].x, U[j].y, U[j].z, U[j].w};
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int row = 16 * j + 4 * r + q; if (row < m) Ua[row] = uv[r]; }
                }
            }
#pragma unroll
            for (int j = 0; j < MT; j++) {         // du' (l_uu du / 2 + l_u): rows of control tile j
                d4 Wu = zero;
#pragma unroll
                for (int j2 = 0; j2 < MT; j2++) Wu = WPn(cur.Luu[j2][j], dU[j2], Wu, mch(j2));     // (l_uu(j2,j))' du(j2) = l_uu(j,j2) du(j2)
                partial += dU[j].x * (0.5 * Wu.x + cur.lu[j].x) + dU[j].y * (0.5 * Wu.y + cur.lu[j].y)
                         + dU[j].z * (0.5 * Wu.z + cur.lu[j].z) + dU[j].w * (0.5 * Wu.w + cur.lu[j].w);
            }
            if constexpr (KP_WIDE_SPREAD) {
#pragma unroll
                for (int j = 0; j < MT; j++) {
                    cur.lu[j] = ld4(rRn, olu[j]);
#pragma unroll
                    for (int j2 = 0; j2 < MT; j2++) cur.Luu[j][j2] = ld4(rRn, oLuu[j][j2]);
                }
            }
        }
        // ---- state cost rows of this tile, then the linearised dynamics for this tile ----------------------------------------------
        d4 Wz = zero, Zn = zero;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            Wz = WPn(cur.Lc[k], Zk[k], Wz, k < NT - 1 ? 4 : ncl);
            if constexpr (KP_WIDE_SPREAD) cur.Lc[k] = ld4(rRn, oLc[k]);
        }
        partial += 0.5 * (Zi.x * Wz.x + Zi.y * Wz.y + Zi.z * Wz.z + Zi.w * Wz.w);
#pragma unroll
        for (int k = 0; k < NT; k++) {
            d4 Ya = cur.Ya[k];
            if (k == tnz) { Ya.x += oneT[0]; Ya.y += oneT[1]; Ya.z += oneT[2]; Ya.w += oneT[3]; }
            Zn = WPn(Ya, Zk[k], Zn, k < NT - 1 ? 4 : ncl);
            if constexpr (KP_WIDE_SPREAD) cur.Ya[k] = ld4(rRn, oA[k]);
        }
#pragma unroll
        for (int j = 0; j < MT; j++) {
            Zn = WPn(cur.Yb[j], dU[j], Zn, mch(j));
            if constexpr (KP_WIDE_SPREAD) cur.Yb[j] = ld4(rRn, oB[j]);
        }
        if constexpr (!KP_WIDE_SPREAD) {
        __builtin_amdgcn_sched_barrier(0);
        load_all(t + 1);                           // single-buffered: everything of step t has been consumed
        __builtin_amdgcn_sched_barrier(0);
        }
        Zi = Zn;
        wsts(zn + wi * WTILE, lane, Zn);
        __syncthreads();
    }
    partial += __shfl_xor(partial, 16);
    partial += __shfl_xor(partial, 32);
    red[wi * 64 + lane] = partial;
    __syncthreads();
    if (wi == 0 && q == 0 && c < n_alpha) {
        double sum = 0.0;
        for (int w = 0; w < NT; w++) sum += red[w * 64 + lane];
        cost_pred[(size_t)b * n_alpha + c] = sum;
    }
}

bool forward_wide_supported(int n, int m, int n_alpha, int nt_min)
{
    const int nt = wide_nt(n, nt_min);
    return nt >= 2 && nt <= 4 && m > 16 && m <= 32 && n_alpha <= 16;
}

template <int NT>
static hipError_t launch_fw(Ctx *c, double *U_alpha_dev)
{
    const size_t lds = sizeof(double) * ((size_t)(2 * NT + NT * 2) * WTILE + NT * 64);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_forward_tiled_wide<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_forward_tiled_wide<NT, 2>), dim3(c->d.batch), dim3(64 * NT), lds, c->stream, c->L, c->d.T, c->d.n_alpha, c->rec,
                       c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev);
    return hipGetLastError();
}

hipError_t launch_forward_wide(Ctx *c, double *U_alpha_dev)
{
    const int nt = wide_nt(c->n, c->tune.tiled_nt_min);
    if (nt == 2) return launch_fw<2>(c, U_alpha_dev);
    if (nt == 3) return launch_fw<3>(c, U_alpha_dev);
    if (nt == 4) return launch_fw<4>(c, U_alpha_dev);
    return hipErrorInvalidValue;
}

}  // namespace kpilqr
