// tiled_mfma.hip -- backward (a7) and forward (a8) passes for state dimensions beyond one MFMA tile:
// n+2 <= 16*NT, NT = 2..4 (Panda pushing n=20, low/moderate/heavy clutter n=38..62), m <= 8.
// Same formulation as riccati_mfma.hip / forward_mfma.hip (homogeneous coordinate, P(Y,X) = Y'X on
// v_mfma_f64_16x16x4_f64, tiles in the accumulator layout), but the matrices are NT x NT grids of
// 16x16 tiles that live in LDS (NT wavefronts per trajectory, each owning one column of tiles; up to 158 KB of
// the CU's 160 KB at NT=4), and products loop over tiles.  This is where the per-step A'V_xxA is a real
// contraction (62^3 at the high-DoF configuration): the FP64 matrix core does 1024 FMAs per ~80-cycle issue.
//
// Reference: iLQR::BackwardsPassQuuRegularisation + CheckMatrixPD, src/Optimiser/iLQR.cpp:535-670;
// control law / clamp of iLQR::ForwardsPassParallel, src/Optimiser/iLQR.cpp:876-890.
//
// LDS tile image: 256 doubles, register r of lane l at [r*64 + l]  (element (4r + (l>>4), l&15)): a tile is
// read back as an MFMA operand with conflict-free 8-byte accesses.  The waves of a workgroup meet at s_barrier
// between the phases of a step (the barrier waits for LDS traffic only, so global loads stay in flight).
#include <cstdlib>
#include <type_traits>
#include "mfma_common.h"
#ifndef KP_COL_STAGE_D
#define KP_COL_STAGE_D 1            // 0: the column kernel stages Fz, Fu at the top of the step: A/B builds
#endif
#ifndef KP_SC_SPREAD
#define KP_SC_SPREAD 1              // 0: the state / cost forward sweep's requests as blocks behind the products: A/B builds
#endif
#ifndef KP_UW_SPREAD
#define KP_UW_SPREAD 1              // 0: the u-wave kernel's requests as blocks behind the products: A/B builds
#endif
#ifndef KP_FT_SPREAD
#define KP_FT_SPREAD 1              // 0: the one-group forward sweeps' requests as blocks behind the products (three / four tiles, every a6-inside form): A/B builds
#endif
#ifndef KP_COL_SPREAD
#define KP_COL_SPREAD 1             // 0: the column kernel's requests as blocks (top of the step, behind the products): A/B builds
#endif
#ifndef KP_NS_HOLD
#define KP_NS_HOLD 8                // factorised steps behind a Newton-Schulz refresh that gave up before the fast path is tried again
#endif
#ifndef KP_FT_SETS
#define KP_FT_SETS 2                // register sets of the two-tile forward sweep on materialised tiles (1: rounds 1-3, A/B builds)
#endif

// latency probe (-DKP_PROBE_SAMEB): every workgroup works on one of two trajectories, everything hits in cache
#ifdef KP_PROBE_SAMEB
#define KP_TILED_TRAJ ((int)(blockIdx.x & 1))
#else
#define KP_TILED_TRAJ ((int)blockIdx.x)
#endif
namespace kpilqr {

typedef unsigned int u32x2t __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define OOBT 0x7ffffff0
#define TILE 256
#define TPAD 272                                     // 16 x 17: padded row-major tile (transposed reads without bank conflicts)

__device__ __forceinline__ double tbld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ d4 lds_tile(const double *t, int lane)
{
    d4 v; v.x = t[lane]; v.y = t[64 + lane]; v.z = t[128 + lane]; v.w = t[192 + lane];
    return v;
}
__device__ __forceinline__ void lds_store(double *t, int lane, const d4 &v)
{
    t[lane] = v.x; t[64 + lane] = v.y; t[128 + lane] = v.z; t[192 + lane] = v.w;
}
// acc += Y'X over `nc` 4-row chunks (nc is wave-uniform)
__device__ __forceinline__ d4 Pn(const d4 &Y, const d4 &X, d4 acc, int nc)
{
    acc = MFMA(Y.x, X.x, acc);
    if (nc > 1) acc = MFMA(Y.y, X.y, acc);
    if (nc > 2) acc = MFMA(Y.z, X.z, acc);
    if (nc > 3) acc = MFMA(Y.w, X.w, acc);
    return acc;
}
// k-th tile of a contraction over NT row tiles: every tile but the last is full (4 chunks, straight-line code so
// the LDS reads of the following tiles are issued ahead of the MFMAs); only the last tile is cut to the chunks
// that hold rows of z (ncl, wave-uniform).
template <int NT>
__device__ __forceinline__ d4 Pk(int k, const d4 &Y, const d4 &X, d4 acc, int ncl)
{
    if (k < NT - 1) {
        acc = MFMA(Y.x, X.x, acc); acc = MFMA(Y.y, X.y, acc); acc = MFMA(Y.z, X.z, acc); acc = MFMA(Y.w, X.w, acc);
        return acc;
    }
    return Pn(Y, X, acc, ncl);
}

// ---------------------------------------------------------------------------------------------
// Backward pass.  z = [dx; 1] has nz = n+1 entries covered by NT row tiles.
// Tile loaders (bounds-checked: structural zeros come back as 0).
struct TileSrc { int n, m; int off_A, off_B, off_lxx, off_lx, off_luu, off_lu; };

// (register r of a tile in the accumulator layout: element (16 ti + 4 r + q, 16 tj + c); the *1 loaders fetch one register, so that
// a sweep can spread its requests under its products)
__device__ __forceinline__ double ld_Lzz1(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int ti, int tj, int r, int q, int c)
{
    const int row = 16 * ti + 4 * r + q, col = 16 * tj + c, n = S.n;
    const int off = (row < n && col < n) ? 8 * (S.off_lxx + row * n + col)
                  : (col == n && row < n) ? 8 * (S.off_lx + row)
                  : (row == n && col < n) ? 8 * (S.off_lx + col) : OOBT;
    return tbld(rs, off);
}
__device__ __forceinline__ double ld_Fz1(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int ti, int tj, int r, int q, int c)
{
    const int row = 16 * ti + 4 * r + q, col = 16 * tj + c, n = S.n;
    return tbld(rs, (row < n && col < n) ? 8 * (S.off_A + col * n + row) : OOBT);
}
__device__ __forceinline__ double ld_Fu1(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int ti, int r, int q, int c)
{
    const int row = 16 * ti + 4 * r + q;
    return tbld(rs, (row < S.n && c < S.m) ? 8 * (S.off_B + c * S.n + row) : OOBT);
}
__device__ __forceinline__ double ld_Luz1(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int tj, int r, int q, int c)
{
    const int row = 4 * r + q, col = 16 * tj + c;
    return tbld(rs, (row < S.m && col == S.n) ? 8 * (S.off_lu + row) : OOBT);
}
__device__ __forceinline__ double ld_Luu1(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int r, int q, int c)
{
    const int row = 4 * r + q;
    return tbld(rs, (row < S.m && c < S.m) ? 8 * (S.off_luu + row * S.m + c) : OOBT);
}
__device__ __forceinline__ d4 ld_Lzz(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int ti, int tj, int q, int c)
{
    d4 o = {ld_Lzz1(rs, S, ti, tj, 0, q, c), ld_Lzz1(rs, S, ti, tj, 1, q, c), ld_Lzz1(rs, S, ti, tj, 2, q, c), ld_Lzz1(rs, S, ti, tj, 3, q, c)};
    return o;
}
__device__ __forceinline__ d4 ld_Fz(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int ti, int tj, int q, int c)
{
    d4 o = {ld_Fz1(rs, S, ti, tj, 0, q, c), ld_Fz1(rs, S, ti, tj, 1, q, c), ld_Fz1(rs, S, ti, tj, 2, q, c), ld_Fz1(rs, S, ti, tj, 3, q, c)};
    return o;
}
__device__ __forceinline__ d4 ld_Fu(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int ti, int q, int c)
{
    d4 o = {ld_Fu1(rs, S, ti, 0, q, c), ld_Fu1(rs, S, ti, 1, q, c), ld_Fu1(rs, S, ti, 2, q, c), ld_Fu1(rs, S, ti, 3, q, c)};
    return o;
}
__device__ __forceinline__ d4 ld_Luz(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int tj, int q, int c)
{
    d4 o = {ld_Luz1(rs, S, tj, 0, q, c), ld_Luz1(rs, S, tj, 1, q, c), ld_Luz1(rs, S, tj, 2, q, c), ld_Luz1(rs, S, tj, 3, q, c)};
    return o;
}
__device__ __forceinline__ d4 ld_Luu(__amdgpu_buffer_rsrc_t rs, const TileSrc &S, int q, int c)
{
    d4 o = {ld_Luu1(rs, S, 0, q, c), ld_Luu1(rs, S, 1, q, c), ld_Luu1(rs, S, 2, q, c), ld_Luu1(rs, S, 3, q, c)};
    return o;
}
__device__ __forceinline__ void setc(d4 &v, int r, double x)
{
    if (r == 0) v.x = x; else if (r == 1) v.y = x; else if (r == 2) v.z = x; else v.w = x;
}

// a6 inside the sweeps (KPILQR_FLAG_FUSED on a tiled shape): the cost tiles are formed from the residuals and their
// Jacobians instead of being read from the records -- Lzz(k,w) = Rz_k' W Rz_w with Rz = [r_x | r] (one column tile each),
// l_uu = Ru' W Ru, l_u = Ru' W r, W = diag(2 w): ModelTranslator::CostDerivativesFromResiduals
// (ModelTranslator.cpp:552-583) as in fused_mfma.hip, so l_xx (n^2 doubles per step) is neither written nor read.
struct CostSrc { const double *r, *r_x, *r_u, *w_run, *w_term; int nr; };

// element (res = 4r+q, col = 16*tj + c) of r_x [nr][n]
__device__ __forceinline__ double ld_Rx1(__amdgpu_buffer_rsrc_t rs, int n, int nr, int tj, int r, int q, int c)
{
    const int res = 4 * r + q, col = 16 * tj + c;
    return tbld(rs, (res < nr && col < n) ? 8 * (res * n + col) : OOBT);
}
// r [nr] in column `cn` of a tile (the homogeneous column n lives in tile n>>4 at column n&15)
__device__ __forceinline__ double ld_R11(__amdgpu_buffer_rsrc_t rs, int nr, int cn, int r, int q, int c)
{
    const int res = 4 * r + q;
    return tbld(rs, (res < nr && c == cn) ? 8 * res : OOBT);
}
// element (res, c) of r_u [nr][m]
__device__ __forceinline__ double ld_Ru1(__amdgpu_buffer_rsrc_t rs, int m, int nr, int r, int q, int c)
{
    const int res = 4 * r + q;
    return tbld(rs, (res < nr && c < m) ? 8 * (res * m + c) : OOBT);
}
__device__ __forceinline__ d4 ld_Rx(__amdgpu_buffer_rsrc_t rs, int n, int nr, int tj, int q, int c)
{
    d4 o = {ld_Rx1(rs, n, nr, tj, 0, q, c), ld_Rx1(rs, n, nr, tj, 1, q, c), ld_Rx1(rs, n, nr, tj, 2, q, c), ld_Rx1(rs, n, nr, tj, 3, q, c)};
    return o;
}
__device__ __forceinline__ d4 ld_R1(__amdgpu_buffer_rsrc_t rs, int nr, int cn, int q, int c)
{
    d4 o = {ld_R11(rs, nr, cn, 0, q, c), ld_R11(rs, nr, cn, 1, q, c), ld_R11(rs, nr, cn, 2, q, c), ld_R11(rs, nr, cn, 3, q, c)};
    return o;
}
__device__ __forceinline__ d4 ld_Ru(__amdgpu_buffer_rsrc_t rs, int m, int nr, int q, int c)
{
    d4 o = {ld_Ru1(rs, m, nr, 0, q, c), ld_Ru1(rs, m, nr, 1, q, c), ld_Ru1(rs, m, nr, 2, q, c), ld_Ru1(rs, m, nr, 3, q, c)};
    return o;
}

// ---------------------------------------------------------------------------------------------
// Backward pass, NT wavefronts per trajectory, COLUMN decomposition: wave w owns column tile w of Tz, Quz,
// Qzz and V'.  Tz(:,w) and Qzz(:,w) never leave the wave's registers (Qzz(i,w) = Lzz(i,w) + sum_k Fz(k,i)'Tz(k,w)
// only needs the wave's own Tz column), Quu is summed from per-wave partials Fu(w)'Tu(w), and the only tiles
// exchanged through LDS are Fz/Fu (staged once per step), X, G and the unsymmetrised V' for the transpose:
//   A  prefetched Fz(:,w), Fu(w): registers -> LDS                                        | barrier
//   BC Tz(:,w), Tu(w), Quu partial, Quz(w), Qzz(i,w) for the OWNED tiles i     (straight-line)    | barrier
//   D  Quu = l_uu + partials; LDL' (every wave); solve / K / k / G for column tile w       | barrier
//   EF V'(i,w) = Qzz(i,w) + X_i' G_w for the owned tiles, written as tile (i,w) AND, transposed, as tile (w,i)
//                                                          (the barrier after the next A orders it)
// Ownership (round 5): Qzz, X'G = -X'(Quu + 2 lambda I)X and V' are symmetric, so every unordered pair {i, j} of tile indices is
// formed ONCE -- wave w owns tiles ((w + d) mod NT, w), d = 0 .. NT/2 (for even NT the antipodal pairs d = NT/2 belong to the
// waves w < NT/2; the others form theirs redundantly and drop it, so that all waves run the same straight-line code): 3 instead
// of 4 tiles of Qzz / Lzz per wave at four tiles (164 -> 148 products per wave and step), and the step's symmetrisation
// (V + V')/2 (iLQR.cpp:610) becomes: the owner's tile (i,w) as it is, its transpose as tile (w,i) -- exactly symmetric by
// construction -- with the average of the reference formed on the diagonal tiles only.  The transposes are wave-local (a private
// padded scratch tile), so the barrier between the old phases E and F is gone.  The off-diagonal tiles differ from the reference's
// average by the rounding-level asymmetry of Qzz (the tests hold the kernel to the oracle at 1e-9 as before).
// All source tiles are single-buffered: re-requested for step t-1 right behind their last use in step t.
// (Rounds 2-4 also carried an A4 form -- Fz, Fu interpolated in registers from the key-point columns of the records, per-lane
// trackers -- parity-green and slower: with per-DoF lists some lane of a wave crosses a key-point on practically every step, so
// every step paid a crossing's 20 loads and 20 divisions per lane; configs[4], round 5: backward 49.6 against 42.7 ms, forward 19.8
// against 12.5 ms, for 5.0 ms of k_interpolate saved.  Removed in round 5.)
// NCL > 0: the chunk count of the last row tile at compile time (see k_backward_tiled_uw) and the products of phase BC as
// interleaved chains; NCL = 0: run-time count, one chain after the other.
template <int M, int NT, bool A6, int NCL = 0>
__global__ void __launch_bounds__(64 * NT)
k_backward_tiled_col(RecLayout L, CostSrc CS, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                     int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                     double *__restrict__ delta_J, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    constexpr int NCU = (M + 3) / 4;
    constexpr int NZZ = NT * NT;
    // SPREAD: the requests for the next step's tiles go out one by one under the products of phase BC (round 5, late: as blocks of
    // twenty at the top of the step and behind the products they were 1 000 + ~700 exposed cycles of a 21 100-cycle step -- the four
    // waves of a trajectory share one address unit)
    constexpr bool SPREAD = KP_COL_SPREAD && NCL > 0;
    static_assert(!SPREAD || NT == 4, "the spread requests deal Fu over the NT row tiles: four registers");
    constexpr bool STAGE_D = SPREAD && KP_COL_STAGE_D;
    // M = 8 is the catch-all instantiation for any num_ctrl <= 8 (walker 6, hopper / pentabot 3, ...): the m x m system is
    // padded with identity rows to 8 x 8 for the (rare) per-lane LDL' steps; everything else works on tiles anyway.
    constexpr bool PAD = (M == 8);
    const int n = L.n, m = PAD ? L.m : M, nz = n + 1;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // this wave's column tile (wave-uniform)
    const int b = KP_TILED_TRAJ;
    const double lam = lambda[b];
    double *bufV = sh;                               // V' (NT x NT tiles, tile (i,j) at i*NT+j)
    double *bufF = bufV + NZZ * TILE;                // Fz
    double *bufT = bufF + NZZ * TILE;                // unsymmetrised V' for the transpose: row-major tiles, row stride 17
    double *bufFu = bufT + NZZ * TPAD;               // NT
    double *bufQuz = bufFu + NT * TILE;              // NT
    double *bufX = bufQuz + NT * TILE;               // NT
    double *bufG = bufX + NT * TILE;                 // NT
    double *bufQp = bufG + NT * TILE;                // NT: per-wave partials of Fu'Tu
    double *sQ = bufQp + NT * TILE + w * TILE;       // NT: this wave's image of Quu + lambda I
    double *sRow = bufT;                             // slow-path work area (phase D: bufT is free outside phase EF)
    // tiles this wave owns: ((w + d) mod NT, w), d < ND (see the header); `red`: the antipodal tile of a wave w >= NT/2 is formed
    // (same code in every wave) and dropped into a dummy area
    constexpr int ND = NT / 2 + 1;
    int ti[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) ti[d] = (w + d) % NT;
    const bool red_last = (NT % 2 == 0) && w >= NT / 2;
    double *scr = bufT + w * TPAD;                   // private padded scratch tile for the wave-local transposes
    double *dummy = bufT + NT * TPAD;                // two tiles nobody reads
    auto nchunk = [&](int kt) { const int rows = nz - 16 * kt; return rows >= 16 ? 4 : (rows + 3) / 4; };
    const int ncl = nchunk(NT - 1);
    const int ncw = (w < NT - 1) ? 4 : ncl;          // chunks of row tile w
    TileSrc S = {n, m, L.off_A, L.off_B, L.off_lxx, L.off_lx, L.off_luu, L.off_lu};

    const double *R0 = rec + (size_t)b * T * L.stride;
    const int rec_bytes = L.rec * 8; (void)rec_bytes;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const int tn = n >> 4, cn = n & 15;
    const bool lane_nn = (c == cn) && (q == (cn & 3));
    const int reg_nn = cn >> 2;
    d4 nn_keep;
    nn_keep.x = (lane_nn && reg_nn == 0) ? 0.0 : 1.0; nn_keep.y = (lane_nn && reg_nn == 1) ? 0.0 : 1.0;
    nn_keep.z = (lane_nn && reg_nn == 2) ? 0.0 : 1.0; nn_keep.w = (lane_nn && reg_nn == 3) ? 0.0 : 1.0;
    double lam2d[4];
#pragma unroll
    for (int r = 0; r < 4; r++) lam2d[r] = (4 * r + q == c && c < m) ? 2.0 * lam : 0.0;

    // source tiles of the current step (column w): Fz(k,w), Lzz(k,w), Fu(w), Luz(w), Luu -- with A6 the cost tiles are
    // formed from residual tiles instead: pL[k] holds Rx_k (r_x rows, column tile k), pLuz the r column, pLuu Ru
    d4 pF[NT], pL[ND], pFu, pLuz, pLuu;
    auto rsrc_of = [&](int t) { return __builtin_amdgcn_make_buffer_rsrc((void *)(R0 + (size_t)t * L.stride), 0, rec_bytes, 0x00020000); };
    const int nr = CS.nr, ncr = (nr + 3) >> 2;
    const double *rb = CS.r + (size_t)b * (T + 1) * nr, *rxb = CS.r_x + (size_t)b * (T + 1) * nr * n,
                 *rub = CS.r_u + (size_t)b * (T + 1) * nr * m;
    d4 W2run = zero, W2term = zero;
    if (A6) {
        double a[4], e[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int res = 4 * r + q;
            a[r] = (res < nr) ? 2.0 * CS.w_run[res] : 0.0;
            e[r] = (res < nr) ? 2.0 * CS.w_term[res] : 0.0;
        }
        W2run.x = a[0]; W2run.y = a[1]; W2run.z = a[2]; W2run.w = a[3];
        W2term.x = e[0]; W2term.y = e[1]; W2term.z = e[2]; W2term.w = e[3];
    }
    auto load_cost = [&](int t, bool ok) {            // the cost sources of step t (zero-size descriptors when !ok)
        if (A6) {
            const size_t tt = ok ? t : 0;
            __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)(rxb + tt * nr * n), 0, ok ? nr * n * 8 : 0, 0x00020000);
            __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void *)(rb + tt * nr), 0, ok ? nr * 8 : 0, 0x00020000);
            __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void *)(rub + tt * nr * m), 0, ok ? nr * m * 8 : 0, 0x00020000);
#pragma unroll
            for (int d = 0; d < ND; d++) pL[d] = ld_Rx(rX, n, nr, ti[d], q, c);        // r_x column tiles of the owned rows (d = 0: tile w itself)
            pLuz = ld_R1(rR, nr, cn, q, c);
            pLuu = ld_Ru(rU, m, nr, q, c);
        } else {
            __amdgpu_buffer_rsrc_t rs = ok ? rsrc_of(t) : __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000);
#pragma unroll
            for (int d = 0; d < ND; d++) pL[d] = ld_Lzz(rs, S, ti[d], w, q, c);
            pLuz = ld_Luz(rs, S, w, q, c); pLuu = ld_Luu(rs, S, q, c);
        }
    };
    // SPREAD: the same requests one register at a time -- request (k, r) of the Qzz loop: register r of pL[k] (k < ND) and, at the
    // last k, of pLuz and pLuu; `rc`: the descriptors of the step (cost_rsrc)
    __amdgpu_buffer_rsrc_t rc0 = __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000), rc1 = rc0, rc2 = rc0;
    auto cost_rsrc = [&](int t, bool ok) {
        if (A6) {
            const size_t tt = ok ? t : 0;
            rc0 = __builtin_amdgcn_make_buffer_rsrc((void *)(rxb + tt * nr * n), 0, ok ? nr * n * 8 : 0, 0x00020000);
            rc1 = __builtin_amdgcn_make_buffer_rsrc((void *)(rb + tt * nr), 0, ok ? nr * 8 : 0, 0x00020000);
            rc2 = __builtin_amdgcn_make_buffer_rsrc((void *)(rub + tt * nr * m), 0, ok ? nr * m * 8 : 0, 0x00020000);
        } else {
            rc0 = ok ? rsrc_of(t) : __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000);
        }
    };
    auto cost_req = [&](int k, int r) {
        if (A6) {
            if (k < ND) setc(pL[k], r, ld_Rx1(rc0, n, nr, ti[k], r, q, c));
            if (k == NT - 1) { setc(pLuz, r, ld_R11(rc1, nr, cn, r, q, c)); setc(pLuu, r, ld_Ru1(rc2, m, nr, r, q, c)); }
        } else {
            if (k < ND) setc(pL[k], r, ld_Lzz1(rc0, S, ti[k], w, r, q, c));
            if (k == NT - 1) { setc(pLuz, r, ld_Luz1(rc0, S, w, r, q, c)); setc(pLuu, r, ld_Luu1(rc0, S, r, q, c)); }
        }
    };
    (void)rc1; (void)rc2;
    // cost tiles of the step whose sources are in pL / pLuz / pLuu:  cL[d] = Lzz(ti[d],w), cLuz = Luz(w), cLuu
    d4 cL[ND], cLuz, cLuu;
    auto form_cost = [&](const d4 &W2) {
        if (A6) {
            const d4 R1w = (w == tn) ? pLuz : zero;
            const d4 WRz = (pL[0] + R1w) * W2;          // column tile w of W Rz
#pragma unroll
            for (int d = 0; d < ND; d++) {
                d4 Rzk = pL[d];
                if (ti[d] == tn) Rzk = Rzk + pLuz;
                cL[d] = Pn(Rzk, WRz, zero, ncr);
            }
            cLuz = (w == tn) ? Pn(pLuu, pLuz * W2, zero, ncr) : zero;     // l_u in column n
            cLuu = (w == 0) ? Pn(pLuu, pLuu * W2, zero, ncr) : zero;
        } else {
#pragma unroll
            for (int d = 0; d < ND; d++) cL[d] = pL[d];
            cLuz = pLuz; cLuu = pLuu;
        }
    };
    __amdgpu_buffer_rsrc_t rs = rsrc_of(T - 1);
#pragma unroll
    for (int k = 0; k < NT; k++) pF[k] = ld_Fz(rs, S, k, w, q, c);
    pFu = ld_Fu(rs, S, w, q, c);
    load_cost(T - 1, true);
    // V' <- Lzz(T-1)   (iLQR.cpp:537-539), terminal weights (Optimiser.cpp:208-211)
    // an owned tile of V' goes out as tile (i,w) and, transposed through the private scratch tile (same-wave LDS accesses are
    // ordered: no barrier), as tile (w,i); the diagonal tile is averaged with its own transpose (iLQR.cpp:610)
    auto publish_V = [&](int d, const d4 &acc, bool average) {
        double *pw = scr + q * 17 + c;                               // element (4r+q, c) at (4r+q)*17 + c
        pw[0] = acc.x; pw[4 * 17] = acc.y; pw[8 * 17] = acc.z; pw[12 * 17] = acc.w;
        const double *pt = scr + c * 17 + q;                         // transposed: element (c, 4r+q); <= 2-way bank conflicts
        d4 at;
        at.x = pt[0]; at.y = pt[4]; at.z = pt[8]; at.w = pt[12];
        if (d == 0) {
            d4 na = average ? 0.5 * (acc + at) : acc;
            if (average && w == tn) na = na * nn_keep;
            lds_store(bufV + (w * NT + w) * TILE, lane, na);
        } else {
            const bool red = red_last && d == NT / 2;
            lds_store(red ? dummy : bufV + (ti[d] * NT + w) * TILE, lane, acc);
            lds_store(red ? dummy + TILE : bufV + (w * NT + ti[d]) * TILE, lane, at);
        }
    };
    form_cost(W2term);
#pragma unroll
    for (int d = 0; d < ND; d++) publish_V(d, cL[d], false);

    int pd_counter = 0, fail = 0;
    double dJ = 0.0;
    d4 Xinv = zero, Iu;                          // running inverse of Quu + lambda I, identity of the u-block
    bool haveX = false;
    int ns_hold = 0;                             // factorised steps left before the fast path is seeded and tried again (phase D)
    Iu.x = (q == c && c < m) ? 1.0 : 0.0; Iu.y = (4 + q == c && c < m) ? 1.0 : 0.0;
    Iu.z = (8 + q == c && c < m) ? 1.0 : 0.0; Iu.w = (12 + q == c && c < m) ? 1.0 : 0.0;
#ifdef KP_CYC_COL
    long long cyc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cc0 = __builtin_readcyclecounter(), cc1;
#define CYK(i) { cc1 = __builtin_readcyclecounter(); cyc[i] += cc1 - cc0; cc0 = cc1; }
#else
#define CYK(i)
#endif
    // the step's tiles into LDS.  STAGE_D (with the spread requests): Fz(:,w), Fu(w) of step t-1 are staged in phase D of step t -- bufF
    // and bufFu are free behind the barrier that ends BC, the tiles were requested under BC's first half, and the writes drain
    // under phase D's per-lane factorisation instead of standing (540 cycles) in front of the step's first barrier (drawn residual
    // Jacobians 36.5 -> 35.9 ms on configs[4]; nothing where the refresh applies)
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < NT; k++) lds_store(bufF + (k * NT + w) * TILE, lane, pF[k]);
        // Fz(n,n) = 1: the loaded element is a structural zero, ONE lane of the wave that owns column n overwrites it (round 5: the
        // add-and-select over every tile of the column was 80 VALU instructions per step)
        if (w == tn && lane_nn) bufF[(tn * NT + w) * TILE + reg_nn * 64 + lane] = 1.0;
        lds_store(bufFu + w * TILE, lane, pFu);
    };
    if constexpr (STAGE_D) stage();
    for (int t = T - 1; t >= 0; t--) {
        pd_counter++;
        const bool check_pd = pd_counter >= pd_stride;
        const bool more = t > 0;
        __amdgpu_buffer_rsrc_t rn = more ? rsrc_of(t - 1) : __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000);
        // ---- A: stage Fz(:,w) (+ the homogeneous 1) and Fu(w) ----------------------------------------------
        CYK(10)
        if constexpr (!STAGE_D) stage();
        __builtin_amdgcn_sched_barrier(0);
        CYK(9)
        if constexpr (!SPREAD) {
#pragma unroll
            for (int k = 0; k < NT; k++) pF[k] = ld_Fz(rn, S, k, w, q, c);
            pFu = ld_Fu(rn, S, w, q, c);
        }
        __builtin_amdgcn_sched_barrier(0);
        CYK(0)
        __syncthreads();
        CYK(1)
        // ---- BC: Tz(:,w), Tu(w), Quu partial, Quz(w), Qzz(:,w) ------------------------------------------------
        form_cost(t == T - 1 ? W2term : W2run);        // cL, cLuz, cLuu of this step (A6: from the residual tiles)
        if constexpr (SPREAD) cost_rsrc(t - 1, more);
        d4 Fc[NT], Tz[NT], Qzz[ND];
        d4 Quzw = cLuz;                                                     // Quz(w): also kept in registers for the fast path
#pragma unroll
        for (int k = 0; k < NT; k++) Fc[k] = lds_tile(bufF + (k * NT + w) * TILE, lane);
        if constexpr (NCL > 0) {
            // independent accumulation chains issued interleaved (back-to-back MFMAs of ONE chain issue every ~80 cycles, of
            // different chains every 64): Tz(0..NT-1,w) with Tu(w); then Quz(w) with Qzz(0..NT-1,w)
            auto nck = [](int kt) { return kt < NT - 1 ? 4 : NCL; };
            auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
            d4 Tu = zero;
#pragma unroll
            for (int i = 0; i < NT; i++) Tz[i] = zero;
#pragma unroll
            for (int k = 0; k < NT; k++) {
                d4 Vk[NT];
#pragma unroll
                for (int i = 0; i < NT; i++) Vk[i] = lds_tile(bufV + (k * NT + i) * TILE, lane);
                const d4 Vw = lds_tile(bufV + (k * NT + w) * TILE, lane), Fuk = lds_tile(bufFu + k * TILE, lane);
#pragma unroll
                for (int r = 0; r < nck(k); r++) {
#pragma unroll
                    for (int i = 0; i < NT; i++) Tz[i] = MFMA(comp(Vk[i], r), comp(Fc[k], r), Tz[i]);
                    Tu = MFMA(comp(Vw, r), comp(Fuk, r), Tu);
                    if constexpr (SPREAD) {         // the next step's Fz(k,w), Fu(w): one request under every group of products
                        setc(pF[k], r, ld_Fz1(rn, S, k, w, r, q, c));
                        if (r == 0) setc(pFu, k, ld_Fu1(rn, S, w, k, q, c));
                        __builtin_amdgcn_sched_group_barrier(0x008, NT + 1, 0);
                        if (r == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if constexpr (SPREAD) {
#pragma unroll
                    for (int r = nck(k); r < 4; r++) setc(pF[k], r, 0.0);       // (chunks of the last row tile that hold no rows of z)
                }
            }
            lds_store(bufQp + w * TILE, lane, Pn(lds_tile(bufFu + w * TILE, lane), Tu, (A6 && w == 0) ? cLuu : zero, ncw));
#pragma unroll
            for (int d = 0; d < ND; d++) Qzz[d] = cL[d];
#pragma unroll
            for (int k = 0; k < NT; k++) {
                d4 Fk[ND];
#pragma unroll
                for (int d = 0; d < ND; d++) Fk[d] = lds_tile(bufF + (k * NT + ti[d]) * TILE, lane);
                const d4 Fuk = lds_tile(bufFu + k * TILE, lane);
#pragma unroll
                for (int r = 0; r < nck(k); r++) {
                    Quzw = MFMA(comp(Fuk, r), comp(Tz[k], r), Quzw);
#pragma unroll
                    for (int d = 0; d < ND; d++) Qzz[d] = MFMA(comp(Fk[d], r), comp(Tz[k], r), Qzz[d]);
                    if constexpr (SPREAD) {         // the next step's cost sources (this step's were consumed by form_cost above)
                        cost_req(k, r);
                        __builtin_amdgcn_sched_group_barrier(0x008, ND + 1, 0);
                        if (k == NT - 1) __builtin_amdgcn_sched_group_barrier(0x020, (NT - 1 < ND) ? 3 : 2, 0);
                        else if (k < ND) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
                if constexpr (SPREAD) {
#pragma unroll
                    for (int r = nck(k); r < 4; r++) cost_req(k, r);
                }
            }
            lds_store(bufQuz + w * TILE, lane, Quzw);
        } else {
#pragma unroll
        for (int i = 0; i < NT; i++) {
            d4 acc = zero;
#pragma unroll
            for (int k = 0; k < NT; k++) acc = Pk<NT>(k, lds_tile(bufV + (k * NT + i) * TILE, lane), Fc[k], acc, ncl);
            Tz[i] = acc;
        }
        {
            d4 Tu = zero;
#pragma unroll
            for (int k = 0; k < NT; k++) Tu = Pk<NT>(k, lds_tile(bufV + (k * NT + w) * TILE, lane), lds_tile(bufFu + k * TILE, lane), Tu, ncl);
            // A6: l_uu rides in wave 0's partial (the other waves never form it)
            lds_store(bufQp + w * TILE, lane, Pn(lds_tile(bufFu + w * TILE, lane), Tu, (A6 && w == 0) ? cLuu : zero, ncw));
        }
#pragma unroll
        for (int k = 0; k < NT; k++) Quzw = Pk<NT>(k, lds_tile(bufFu + k * TILE, lane), Tz[k], Quzw, ncl);
        lds_store(bufQuz + w * TILE, lane, Quzw);
#pragma unroll
        for (int d = 0; d < ND; d++) {
            d4 acc = cL[d];
#pragma unroll
            for (int k = 0; k < NT; k++) acc = Pk<NT>(k, lds_tile(bufF + (k * NT + ti[d]) * TILE, lane), Tz[k], acc, ncl);
            Qzz[d] = acc;
        }
        }
        const d4 Luu_t = cLuu;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!SPREAD) load_cost(t - 1, more);
        __builtin_amdgcn_sched_barrier(0);
        CYK(2)
        __syncthreads();
        CYK(3)
        // ---- D: Quu, LDL' (every wave: keeps the PD verdict block-uniform), solve / K / G for column tile w ----
        d4 Quu = A6 ? zero : Luu_t;
        d4 Qp_[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) Qp_[k] = lds_tile(bufQp + k * TILE, lane);
        if constexpr (STAGE_D) stage();
#pragma unroll
        for (int k = 0; k < NT; k++) Quu = Quu + Qp_[k];
        d4 Qr = Quu;
        Qr.x += 0.5 * lam2d[0]; Qr.y += 0.5 * lam2d[1]; Qr.z += 0.5 * lam2d[2]; Qr.w += 0.5 * lam2d[3];
        // X(w) = (Quu + lambda I)^-1 Quz(w).  Fast path as in riccati_mfma.hip: every wave keeps the running inverse
        // and refreshes it with Newton-Schulz steps on the matrix core (identical inputs in every wave, so the
        // decisions below are block-uniform); LDL' on the first step, on checked steps (PD verdict) and when the
        // residual is too large to converge fast.
        d4 X = zero;
        bool done = false;
        const bool tried = haveX && !check_pd;
        if (tried && kp_inverse_refresh<NCU>(Qr, Iu, Xinv, m)) {
            X = Pn(Xinv, Quzw, zero, NCU);
            done = true;
        }
        if (!done) {
            // The refresh gave up (residual beyond quadratic convergence in four steps): on data whose Quu jumps from step to step it
            // will again -- for the next KP_NS_HOLD factorised steps the fast path is neither seeded (one LDL' solve less per lane)
            // nor tried (round 5: ~750 of a step's 3 600 cycles in phase D on the independently drawn residual Jacobians; data on
            // which the refresh converges never gets here).  Block-uniform like every decision of this phase.
            if (tried) ns_hold = KP_NS_HOLD;
            lds_store(sQ, lane, Qr);                  // this wave's private image (same-wave LDS accesses are ordered)
            auto qel = [&](int i, int j) {
                if (PAD && (i >= m || j >= m)) return (i == j) ? 1.0 : 0.0;
                return sQ[(i >> 2) * 64 + j + 16 * (i & 3)];
            };
            double Lm[M][M], rd[M];
            const bool pos = kp_ldl_factor<M>([&](int i, int j) { return qel(i, j); }, Lm, rd);
            if (check_pd) {                       // CheckMatrixPD every pd_stride steps   :587-595
                if (!pos) { fail = t + 1; break; }
                pd_counter = 0;
            }
            double *winv = sRow + 256 + 256;
            if (!pos) {
                if (threadIdx.x == 0) {
                    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) sRow[i * 16 + j] = qel(i, j);
                    kp_slow_ldlt_inverse(m, sRow, 16, sRow + 256, winv, sRow + 768, (int *)(sRow + 784));
                }
                __syncthreads();
            }
            const double *zt = bufQuz + w * TILE;
            double x[M];
#pragma unroll
            for (int i = 0; i < M; i++) x[i] = zt[(i >> 2) * 64 + c + 16 * (i & 3)];
            if (pos) {
                kp_ldl_solve<M>(Lm, rd, x);
                if (ns_hold > 0) {
                    ns_hold--;
                    haveX = false;
                } else {
                double y[M];                          // seed the fast path: column c of the inverse in lane c (c < m)
#pragma unroll
                for (int i = 0; i < M; i++) y[i] = (i == c) ? 1.0 : 0.0;
                kp_ldl_solve<M>(Lm, rd, y);
                double yr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < M; i++)
                    if (q == (i & 3)) yr[i >> 2] = (c < m) ? y[i] : 0.0;
                Xinv.x = yr[0]; Xinv.y = yr[1]; Xinv.z = yr[2]; Xinv.w = yr[3];
                haveX = true;
                }
            } else {
                double y[M];
#pragma unroll
                for (int i = 0; i < M; i++) {
                    double sacc = 0.0;
#pragma unroll
                    for (int pp = 0; pp < M; pp++) sacc += (!PAD || (i < m && pp < m)) ? (-winv[i + pp * m]) * x[pp] : 0.0;
                    y[i] = -sacc;
                }
#pragma unroll
                for (int i = 0; i < M; i++) x[i] = y[i];
                haveX = false;
            }
            const int colx = 16 * w + c;
            double xr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < M; i++)
                if (q == (i & 3)) xr[i >> 2] = (colx <= n) ? x[i] : 0.0;
            X.x = xr[0]; X.y = xr[1]; X.z = xr[2]; X.w = xr[3];
        }
        d4 Gw;
        {
            const int col = 16 * w + c;
            lds_store(bufX + w * TILE, lane, X);
            __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void *)(Kout + ((size_t)b * T + t) * m * n), 0, m * n * 8, 0x00020000);
            __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(kout + ((size_t)b * T + t) * m), 0, m * 8, 0x00020000);
            const double xv[4] = {X.x, X.y, X.z, X.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) {
                const int row = 4 * r + q;
                const double kv = -xv[r];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2t, kv), rK, (row < m && col < n) ? 8 * (row + col * m) : OOBT, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2t, kv), rk, (row < m && col == n) ? 8 * row : OOBT, 0, 0);
            }
            if (col == n) {                           // delta_J -= lambda k'k: lane-local squares, reduced after the sweep
#pragma unroll
                for (int r = 0; r < NCU; r++) dJ -= lam * (xv[r] * xv[r]);
            }
            // G = (Quu + 2 lambda I) K' = -(Quz + lambda X) because (Quu + lambda I) X = Quz: no product needed
            Gw.x = -__builtin_fma(lam, X.x, Quzw.x); Gw.y = -__builtin_fma(lam, X.y, Quzw.y);
            Gw.z = -__builtin_fma(lam, X.z, Quzw.z); Gw.w = -__builtin_fma(lam, X.w, Quzw.w);
        }
        CYK(4)
        __syncthreads();
        CYK(5)
        // ---- EF: V'(i,w) = Qzz(i,w) + X_i' G_w for the owned tiles, out as tile (i,w) and transposed as tile (w,i) ----------------
#pragma unroll
        for (int d = 0; d < ND; d++) publish_V(d, Pn(lds_tile(bufX + ti[d] * TILE, lane), Gw, Qzz[d], NCU), true);
        CYK(8)
    }
#ifdef KP_CYC_COL
    if (b == 1 && lane == 0) printf("col wave %d: top %lld stage %lld | A %lld | wait1 %lld | BC %lld | wait2 %lld | D %lld | wait3 %lld | EF %lld  (cycles per step)\n", w,
                                    cyc[10] / T, cyc[9] / T, cyc[0] / T, cyc[1] / T, cyc[2] / T, cyc[3] / T, cyc[4] / T, cyc[5] / T, cyc[8] / T);
#endif
    dJ += __shfl_xor(dJ, 16);
    dJ += __shfl_xor(dJ, 32);
    if (w == tn && lane_nn) delta_J[b] = dJ;
    if (threadIdx.x == 0) status[b] = fail;
}

size_t backward_col_lds_bytes(int nt)
{
    const size_t tpad = (size_t)nt * nt * TPAD;     // bufT, which also hosts the slow-path work area (832 doubles)
    return sizeof(double) * ((size_t)(2 * nt * nt + 7 * nt) * TILE + (tpad > 832 ? tpad : 832));
}

size_t backward_tiled_lds_bytes(int nt) { return backward_col_lds_bytes(nt); }

static int tiled_nt(int n, int nt_min)
{
    int nt = (n + 1 + 15) / 16;
    if (nt < 2) nt = 2;
    if (nt_min > nt) nt = nt_min;                       // diagnostic (KPILQR_TILED_NT_MIN): more tiles than needed
    return nt;
}

int tiled_tiles(int n, int nt_min) { return tiled_nt(n, nt_min); }

bool backward_tiled_supported(int n, int m, int nt_min)
{
    const int nt = tiled_nt(n, nt_min);
    return nt >= 2 && nt <= 4 && m >= 1 && m <= 8 && backward_tiled_lds_bytes(nt) <= 160 * 1024;
}

// ---------------------------------------------------------------------------------------------
// The same sweep with one more wavefront per trajectory, the "u-wave" (materialised A, B and cost tiles; NT <= 3, so that
// the NT+1 waves still have a SIMD each).  In k_backward_tiled_col every wave repeats, behind the products, the chain
// Quu -> refresh of the running inverse (or LDL') -> X, which is a third of an n=20 step (2 260 of 7 830 cycles) and
// depends on nothing but V' Fu.  Here wave NT forms Tu = V'Fu (all row tiles), Quu = l_uu + Fu'Tu and the inverse WHILE the
// column waves form Tz, Quz, Qzz; they meet at the second barrier, behind which a column wave only multiplies
// X(w) = Xinv Quz(w).  The PD verdict (:587-595) and the LDL' fallbacks are the u-wave's alone and reach the others as a flag
// and as an inverse tile: on factorised steps X is Xinv Quz too (Xinv = the per-lane solves of the identity columns).
//   (F of the step above: V' complete)                                                                   | barrier
//   B  column wave w: Tz(:,w) = V'Fz(:,w) from registers, Fz(:,w), Fu(w) -> LDS  ||  u-wave: Tu = V'Fu    | barrier
//   C  column wave w: Quz(w), Qzz(:,w)               ||  u-wave: Quu, refresh / LDL' -> Xinv, flag       | barrier
//   D  X(w) = Xinv Quz(w), K / k stores, G(w)                                                            | barrier
//   EF as above: the owned tiles of V' and their transposes, wave-local (the u-wave only keeps the barrier count)
// The u-wave reads Fu from global memory itself (its slots of the per-step requests), so its chain -- the critical path of the
// step -- starts at the first barrier and not behind the column waves' staging.
// NCL = 4-row chunks of the last row tile that hold rows of z (compile-time here: a run-time count puts every chunk of the last
// tile behind its own branch, and the basic blocks that leaves keep the scheduler from interleaving independent MFMA chains).
template <int M, int NT, int NCL>
__global__ void __launch_bounds__(64 * (NT + 1))
k_backward_tiled_uw(RecLayout L, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                    int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                    double *__restrict__ delta_J, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double sh[];
    constexpr int NCU = (M + 3) / 4;
    constexpr int NZZ = NT * NT;
    constexpr bool PAD = (M == 8);
    const int n = L.n, m = PAD ? L.m : M;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // column tile of this wave; NT: the u-wave
    const bool uw = (w == NT);
    const int b = KP_TILED_TRAJ;
    const double lam = lambda[b];
    double *bufV = sh;
    double *bufF = bufV + NZZ * TILE;
    double *bufT = bufF + NZZ * TILE;
    double *bufFu = bufT + NZZ * TPAD;
    double *bufQuz = bufFu + NT * TILE;              // (unused here: Quz(w) stays in its wave)
    double *bufX = bufQuz + NT * TILE;
    double *bufG = bufX + NT * TILE;
    double *sQ = bufG + NT * TILE;                   // the u-wave's image of Quu + lambda I
    double *bufXi = sQ + TILE;                       // the inverse, u-wave -> column waves
    int *sFlag = (int *)(bufXi + TILE);              // PD verdict of the step (0 | t+1)
    double *sRow = bufT;                             // slow-path work area of the u-wave (phase BC: bufT is free between F and E)
    TileSrc S = {n, m, L.off_A, L.off_B, L.off_lxx, L.off_lx, L.off_luu, L.off_lu};
    auto nck = [](int kt) { return kt < NT - 1 ? 4 : NCL; };        // chunks of row tile kt
    auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };

    const double *R0 = rec + (size_t)b * T * L.stride;
    const int rec_bytes = L.rec * 8; (void)rec_bytes;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const int tn = n >> 4, cn = n & 15;
    const bool lane_nn = (c == cn) && (q == (cn & 3));
    const int reg_nn = cn >> 2;
    d4 nn_keep;
    nn_keep.x = (lane_nn && reg_nn == 0) ? 0.0 : 1.0; nn_keep.y = (lane_nn && reg_nn == 1) ? 0.0 : 1.0;
    nn_keep.z = (lane_nn && reg_nn == 2) ? 0.0 : 1.0; nn_keep.w = (lane_nn && reg_nn == 3) ? 0.0 : 1.0;
    const d4 nn_one = 1.0 - nn_keep;
    double lam2d[4];
#pragma unroll
    for (int r = 0; r < 4; r++) lam2d[r] = (4 * r + q == c && c < m) ? 2.0 * lam : 0.0;
    auto rsrc_of = [&](int t) { return __builtin_amdgcn_make_buffer_rsrc((void *)(R0 + (size_t)t * L.stride), 0, rec_bytes, 0x00020000); };
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, const int (&o)[4]) { d4 v = {tbld(rs, o[0]), tbld(rs, o[1]), tbld(rs, o[2]), tbld(rs, o[3])}; return v; };
    // V'(i,k)' Y(k) summed over the row tiles k, for every output tile i: NT independent accumulation chains issued interleaved.
    // (More, shallower chains -- one per (k,i), halves of Fu'Tu, split products in the refresh -- measured no gain: at one FP64 MFMA
    // per 64 cycles the u-wave's 18 products are issue-bound, not chain-bound; nor did a third-order refresh, with or without the
    // extrapolated first guess of kp_inverse_refresh_p: 7.2 / 8.4 against 6.8 ms on the pushing workload.)
    auto VtY = [&](const d4 (&Y)[NT], d4 (&Tq)[NT]) {
#pragma unroll
        for (int i = 0; i < NT; i++) Tq[i] = zero;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            d4 Vk[NT];
#pragma unroll
            for (int i = 0; i < NT; i++) Vk[i] = lds_tile(bufV + (k * NT + i) * TILE, lane);
#pragma unroll
            for (int r = 0; r < nck(k); r++)
#pragma unroll
                for (int i = 0; i < NT; i++) Tq[i] = MFMA(comp(Vk[i], r), comp(Y[k], r), Tq[i]);
        }
    };
    int pd_counter = 0, fail = 0, ns_hold = 0;
    double dJ = 0.0;
    (void)ns_hold;
#ifdef KP_CYC_UW
    int hist[6] = {0, 0, 0, 0, 0, 0};
    long long cy[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c0 = __builtin_readcyclecounter(), c1;
#define CYC(i) { c1 = __builtin_readcyclecounter(); cy[i] += c1 - c0; c0 = c1; }
#else
#define CYC(i)
#endif
    // The two roles run their own time loops (no joins of the two paths inside a step: a join behind a request copies the
    // requested registers and waits for every load on the spot) and meet at the five barriers of a step.
    if (uw) {
        // ================================================ u-wave ======================================================
        // sources, one step ahead: Fu (NT row tiles, consumed right behind the first barrier) and l_uu (consumed behind the second)
        int oFu[NT][4], oLuu[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int rowu = 4 * r + q;
#pragma unroll
            for (int k = 0; k < NT; k++) {
                const int row = 16 * k + 4 * r + q;
                oFu[k][r] = (row < n && c < m) ? 8 * (S.off_B + c * n + row) : OOBT;
            }
            oLuu[r] = (rowu < m && c < m) ? 8 * (S.off_luu + rowu * m + c) : OOBT;
        }
        d4 pFu[NT], pLuu;
        {
            __amdgpu_buffer_rsrc_t rs = rsrc_of(T - 1);
#pragma unroll
            for (int k = 0; k < NT; k++) pFu[k] = ld4(rs, oFu[k]);
            pLuu = ld4(rs, oLuu);
        }
        d4 Xinv = zero, Iu;                      // running inverse of Quu + lambda I, identity of the u-block
        bool haveX = false;
        Iu.x = (q == c && c < m) ? 1.0 : 0.0; Iu.y = (4 + q == c && c < m) ? 1.0 : 0.0;
        Iu.z = (8 + q == c && c < m) ? 1.0 : 0.0; Iu.w = (12 + q == c && c < m) ? 1.0 : 0.0;
        for (int t = T - 1; t >= 0; t--) {
            pd_counter++;
            const bool check_pd = pd_counter >= pd_stride;
            __amdgpu_buffer_rsrc_t rn = t > 0 ? rsrc_of(t - 1) : __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000);
            CYC(0)
            __syncthreads();                                       // V' of the step above is complete
            CYC(1)
            // Quu = l_uu + sum_i Fu(i)' (V'Fu)(i);  (V'Fu)(i) = sum_k V'(k,i)' Fu(k) by the symmetry of V'
            d4 Tu[NT], Qp[NT];
            VtY(pFu, Tu);
#pragma unroll
            for (int i = 0; i < NT; i++) Qp[i] = zero;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < NT; i++) {
                    if (r < nck(i)) Qp[i] = MFMA(comp(pFu[i], r), comp(Tu[i], r), Qp[i]);
                    if constexpr (KP_UW_SPREAD) {   // the next step's Fu, register by register behind its last use (see k_backward_tiled_col)
                        setc(pFu[i], r, tbld(rn, oFu[i][r]));
                        if (r < nck(i)) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!KP_UW_SPREAD) {
#pragma unroll
                for (int k = 0; k < NT; k++) pFu[k] = ld4(rn, oFu[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
            CYC(7)
            __syncthreads();                                       // (the column waves' Fz, Fu are in LDS: not used here)
            CYC(9)
            d4 Qr = pLuu;
#pragma unroll
            for (int i = 0; i < NT; i++) Qr = Qr + Qp[i];
            pLuu = ld4(rn, oLuu);
            Qr.x += 0.5 * lam2d[0]; Qr.y += 0.5 * lam2d[1]; Qr.z += 0.5 * lam2d[2]; Qr.w += 0.5 * lam2d[3];
            // running inverse refreshed by Newton-Schulz steps on the matrix core; LDL' on the first step, on checked
            // steps (PD verdict) and when the residual is too large to converge fast (as in k_backward_tiled_col)
#ifdef KP_CYC_UW
            {   // histogram of the refresh's starting residual (the bins of kp_inverse_refresh: 1, 2, 3, 4 steps, factorise)
                const d4 R0 = Iu - kp_P<NCU>(Qr, Xinv, zero);
                double rm = fabs(R0.x);
                if (NCU > 1) rm = fmax(rm, fabs(R0.y));
                const double e0 = (double)m * rm;
                const int bin = !haveX || check_pd ? 5 : (__builtin_amdgcn_ballot_w64(!(e0 < 0.11)) != 0) ? 4 : (__builtin_amdgcn_ballot_w64(e0 >= 1.3e-2) != 0) ? 3
                              : (__builtin_amdgcn_ballot_w64(e0 >= 1.7e-4) != 0) ? 2 : (__builtin_amdgcn_ballot_w64(e0 >= 3.0e-8) != 0) ? 1 : 0;
                hist[bin]++;
            }
#endif
            // (a refresh that gave up is not tried again for KP_NS_HOLD steps: see k_backward_tiled_col)
            const bool tried = haveX && !check_pd && ns_hold == 0;
            if (ns_hold > 0) ns_hold--;
            const bool refreshed = tried && kp_inverse_refresh<NCU>(Qr, Iu, Xinv, m);
            if (tried && !refreshed) ns_hold = KP_NS_HOLD;
            if (!refreshed) {
                lds_store(sQ, lane, Qr);
                auto qel = [&](int i, int j) {
                    if (PAD && (i >= m || j >= m)) return (i == j) ? 1.0 : 0.0;
                    return sQ[(i >> 2) * 64 + j + 16 * (i & 3)];
                };
                double Lm[M][M], rd[M];
                const bool pos = kp_ldl_factor<M>([&](int i, int j) { return qel(i, j); }, Lm, rd);
                if (check_pd && !pos) fail = t + 1;                // CheckMatrixPD every pd_stride steps   :587-595
                double yr[4] = {0.0, 0.0, 0.0, 0.0};
                if (pos) {
                    double y[M];                                   // column c of the inverse in lane c (c < m)
#pragma unroll
                    for (int i = 0; i < M; i++) y[i] = (i == c) ? 1.0 : 0.0;
                    kp_ldl_solve<M>(Lm, rd, y);
#pragma unroll
                    for (int i = 0; i < M; i++)
                        if (q == (i & 3)) yr[i >> 2] = (c < m) ? y[i] : 0.0;
                    haveX = true;
                } else if (!fail) {                                // indefinite between checks: the reference's LDLT route
                    double *winv = sRow + 256 + 256;
                    if (lane == 0) {
                        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) sRow[i * 16 + j] = qel(i, j);
                        kp_slow_ldlt_inverse(m, sRow, 16, sRow + 256, winv, sRow + 768, (int *)(sRow + 784));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int r = 0; r < NCU; r++) {
                        const int row = 4 * r + q;
                        yr[r] = (row < m && c < m) ? winv[row + c * m] : 0.0;
                    }
                    haveX = false;
                }
                Xinv.x = yr[0]; Xinv.y = yr[1]; Xinv.z = yr[2]; Xinv.w = yr[3];
            }
            CYC(8)
            lds_store(bufXi, lane, Xinv);
            if (lane == 0) sFlag[0] = fail;
            CYC(2)
            __syncthreads();                                       // the inverse and the verdict are out
            CYC(3)
            if (fail) break;
            if (check_pd) pd_counter = 0;
            __syncthreads();                                       // (the column waves' X tiles are out)
            CYC(4)
        }
    } else {
        // ============================================== column wave w ==================================================
        // sources, one step ahead: Fz(k,w), Fu(w) (consumed behind the first barrier) | Lzz(k,w), Luz(w) (joined at the end of C)
        // tiles this wave owns (see k_backward_tiled_col): ((w + d) mod NT, w), d < ND; the antipodal tile of a wave w >= NT/2
        // (even NT) is formed and dropped
        constexpr int ND = NT / 2 + 1;
        int ti[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) ti[d] = (w + d) % NT;
        const bool red_last = (NT % 2 == 0) && w >= NT / 2;
        double *scr = bufT + w * TPAD, *dummy = bufT + NT * TPAD;
        d4 pF[NT], pL[ND], pFu, pLuz;
        int oF[NT][4], oL[ND][4], oFu[4], oLuz[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int rowu = 4 * r + q, col = 16 * w + c, rowb = 16 * w + 4 * r + q;
#pragma unroll
            for (int k = 0; k < NT; k++) {
                const int row = 16 * k + 4 * r + q;
                oF[k][r] = (row < n && col < n) ? 8 * (S.off_A + col * n + row) : OOBT;
            }
#pragma unroll
            for (int d = 0; d < ND; d++) {
                const int row = 16 * ti[d] + 4 * r + q;
                oL[d][r] = (row < n && col < n) ? 8 * (S.off_lxx + row * n + col)
                         : (col == n && row < n) ? 8 * (S.off_lx + row) : (row == n && col < n) ? 8 * (S.off_lx + col) : OOBT;
            }
            oFu[r] = (rowb < n && c < m) ? 8 * (S.off_B + c * n + rowb) : OOBT;
            oLuz[r] = (rowu < m && col == n) ? 8 * (S.off_lu + rowu) : OOBT;
        }
        auto request = [&](__amdgpu_buffer_rsrc_t rs) {
#pragma unroll
            for (int k = 0; k < NT; k++) pF[k] = ld4(rs, oF[k]);
            pFu = ld4(rs, oFu);
        };
        auto request_cost = [&](__amdgpu_buffer_rsrc_t rs) {
#pragma unroll
            for (int d = 0; d < ND; d++) pL[d] = ld4(rs, oL[d]);
            pLuz = ld4(rs, oLuz);
        };
        {
            __amdgpu_buffer_rsrc_t rs = rsrc_of(T - 1);
            request(rs); request_cost(rs);
        }
        // an owned tile of V' goes out as tile (i,w) and, transposed through the private scratch tile (same-wave LDS accesses are
        // ordered), as tile (w,i); the diagonal tile is averaged with its own transpose (iLQR.cpp:610)
        auto publish_V = [&](int d, const d4 &acc, bool average) {
            double *pw = scr + q * 17 + c;
            pw[0] = acc.x; pw[4 * 17] = acc.y; pw[8 * 17] = acc.z; pw[12 * 17] = acc.w;
            const double *pt = scr + c * 17 + q;
            d4 at;
            at.x = pt[0]; at.y = pt[4]; at.z = pt[8]; at.w = pt[12];
            if (d == 0) {
                d4 na = average ? 0.5 * (acc + at) : acc;
                if (average && w == tn) na = na * nn_keep;
                lds_store(bufV + (w * NT + w) * TILE, lane, na);
            } else {
                const bool red = red_last && d == NT / 2;
                lds_store(red ? dummy : bufV + (ti[d] * NT + w) * TILE, lane, acc);
                lds_store(red ? dummy + TILE : bufV + (w * NT + ti[d]) * TILE, lane, at);
            }
        };
        // V' <- Lzz(T-1)   (iLQR.cpp:537-539)
#pragma unroll
        for (int d = 0; d < ND; d++) publish_V(d, pL[d], false);
        char *pK = (char *)(Kout + ((size_t)b * T + (T - 1)) * m * n), *pk = (char *)(kout + ((size_t)b * T + (T - 1)) * m);
        for (int t = T - 1; t >= 0; t--) {
            pd_counter++;
            const bool check_pd = pd_counter >= pd_stride;
            __amdgpu_buffer_rsrc_t rn = t > 0 ? rsrc_of(t - 1) : __builtin_amdgcn_make_buffer_rsrc((void *)R0, 0, 0, 0x00020000);
            CYC(0)
            __syncthreads();                                       // V' of the step above is complete
            CYC(1)
            // ---- B: Tz(:,w) = V' Fz(:,w) straight from the registers the tiles were loaded into; Fz(:,w), Fu(w) -> LDS ----
            d4 Tz[NT];
            {
                d4 Y[NT];
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    Y[k] = pF[k];
                    if (k == tn && w == tn) Y[k] = Y[k] + nn_one;          // Fz(n,n) = 1
                }
                if constexpr (KP_UW_SPREAD) {
                    // VtY with the next step's requests one by one under the products (see k_backward_tiled_col): Fz(k,w) register r
                    // behind the products that read it (from Y, its copy), Fu(w) under the first row tile
                    const d4 Fuw = pFu;
#pragma unroll
                    for (int i = 0; i < NT; i++) Tz[i] = zero;
#pragma unroll
                    for (int k = 0; k < NT; k++) {
                        d4 Vk[NT];
#pragma unroll
                        for (int i = 0; i < NT; i++) Vk[i] = lds_tile(bufV + (k * NT + i) * TILE, lane);
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if (r < nck(k)) {
#pragma unroll
                                for (int i = 0; i < NT; i++) Tz[i] = MFMA(comp(Vk[i], r), comp(Y[k], r), Tz[i]);
                                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                            }
                            setc(pF[k], r, tbld(rn, oF[k][r]));
                            if (k == 0) setc(pFu, r, tbld(rn, oFu[r]));
                            if (k == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                            else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NT; k++) lds_store(bufF + (k * NT + w) * TILE, lane, Y[k]);
                    lds_store(bufFu + w * TILE, lane, Fuw);
                } else {
                VtY(Y, Tz);
#pragma unroll
                for (int k = 0; k < NT; k++) lds_store(bufF + (k * NT + w) * TILE, lane, Y[k]);
                lds_store(bufFu + w * TILE, lane, pFu);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!KP_UW_SPREAD) request(rn);
            __builtin_amdgcn_sched_barrier(0);
            CYC(7)
            __syncthreads();                                       // every wave's Fz, Fu are in LDS
            CYC(9)
            // ---- C: Quz(w), Qzz(:,w): NT+1 chains interleaved; the cost tiles join at the end (their loads have had the step) ----
            d4 Quzw = zero, Qzz[ND];
#pragma unroll
            for (int d = 0; d < ND; d++) Qzz[d] = zero;
            if constexpr (KP_UW_SPREAD) {       // the cost tiles start the chains (requested a step ago), their registers are free for the next step's
                Quzw = pLuz;
#pragma unroll
                for (int d = 0; d < ND; d++) Qzz[d] = pL[d];
            }
#pragma unroll
            for (int k = 0; k < NT; k++) {
                d4 Fk[ND];
                const d4 Fuk = lds_tile(bufFu + k * TILE, lane);
#pragma unroll
                for (int d = 0; d < ND; d++) Fk[d] = lds_tile(bufF + (k * NT + ti[d]) * TILE, lane);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (r < nck(k)) {
                        Quzw = MFMA(comp(Fuk, r), comp(Tz[k], r), Quzw);
#pragma unroll
                        for (int d = 0; d < ND; d++) Qzz[d] = MFMA(comp(Fk[d], r), comp(Tz[k], r), Qzz[d]);
                        if constexpr (KP_UW_SPREAD) __builtin_amdgcn_sched_group_barrier(0x008, ND + 1, 0);
                    }
                    if constexpr (KP_UW_SPREAD) {               // request (k, r): register r of pL[k] (k < ND) and, at k = NT-1, of pLuz
                        if (k < ND) setc(pL[k], r, tbld(rn, oL[k][r]));
                        if (k == NT - 1) setc(pLuz, r, tbld(rn, oLuz[r]));
                        if (k < ND && k == NT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                        else if (k < ND || k == NT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
            if constexpr (!KP_UW_SPREAD) {
            Quzw = Quzw + pLuz;
#pragma unroll
            for (int d = 0; d < ND; d++) Qzz[d] = Qzz[d] + pL[d];
            }
            CYC(8)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!KP_UW_SPREAD) request_cost(rn);         // (issued while the u-wave still refreshes the inverse)
            __builtin_amdgcn_sched_barrier(0);
            CYC(2)
            __syncthreads();                                       // the inverse and the verdict are out
            CYC(3)
            // ---- D: X(w) = Xinv Quz(w), K / k, G(w) ----------------------------------------------------------------
            fail = sFlag[0];
            if (fail) break;
            if (check_pd) pd_counter = 0;
            const d4 X = Pn(lds_tile(bufXi, lane), Quzw, zero, NCU);
            const int col = 16 * w + c;
            lds_store(bufX + w * TILE, lane, X);
            __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void *)pK, 0, m * n * 8, 0x00020000);
            __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)pk, 0, m * 8, 0x00020000);
            pK -= (size_t)m * n * 8; pk -= (size_t)m * 8;
            const double xv[4] = {X.x, X.y, X.z, X.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) {
                const int row = 4 * r + q;
                const double kv = -xv[r];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2t, kv), rK, (row < m && col < n) ? 8 * (row + col * m) : OOBT, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2t, kv), rk, (row < m && col == n) ? 8 * row : OOBT, 0, 0);
            }
            if (col == n) {                           // delta_J -= lambda k'k: lane-local squares, reduced after the sweep
#pragma unroll
                for (int r = 0; r < NCU; r++) dJ -= lam * (xv[r] * xv[r]);
            }
            // G = (Quu + 2 lambda I) K' = -(Quz + lambda X) because (Quu + lambda I) X = Quz: no product needed
            d4 Gw;
            Gw.x = -__builtin_fma(lam, X.x, Quzw.x); Gw.y = -__builtin_fma(lam, X.y, Quzw.y);
            Gw.z = -__builtin_fma(lam, X.z, Quzw.z); Gw.w = -__builtin_fma(lam, X.w, Quzw.w);
            CYC(4)
            __syncthreads();
            // ---- EF: V'(i,w) = Qzz(i,w) + X_i' G_w for the owned tiles, out as tile (i,w) and transposed as tile (w,i) ------------
#pragma unroll
            for (int d = 0; d < ND; d++) publish_V(d, Pn(lds_tile(bufX + ti[d] * TILE, lane), Gw, Qzz[d], NCU), true);
            CYC(5)
        }
    }
#ifdef KP_CYC_UW
    if (b == 0 && lane == 0) printf("wave %d: A %lld | wait1 %lld | BC %lld | wait2 %lld | D %lld | E+wait3 %lld | wait4 %lld | B %lld wait1b %lld C %lld (cycles per step; F is in A)\n", w,
                                    cy[0] / T, cy[1] / T, cy[2] / T, cy[3] / T, cy[4] / T, cy[5] / T, cy[6] / T, cy[7] / T, cy[9] / T, cy[8] / T);
    if (b == 0 && lane == 0 && uw) printf("u-wave refresh, steps by Newton-Schulz count: 1: %d | 2: %d | 3: %d | 4: %d | residual too large, factorised: %d | first / checked steps: %d\n",
                                          hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
#endif
    dJ += __shfl_xor(dJ, 16);
    dJ += __shfl_xor(dJ, 32);
    if (w == tn && lane_nn) delta_J[b] = dJ;
    if (threadIdx.x == 0) status[b] = fail;
}

template <int M, int NT, int NCL>
static hipError_t launch_bt_uw2(Ctx *c, int pd_stride)
{
    const size_t ldc = backward_col_lds_bytes(NT);
    hipError_t e = hipFuncSetAttribute((const void *)k_backward_tiled_uw<M, NT, NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldc);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_backward_tiled_uw<M, NT, NCL>), dim3(c->d.batch), dim3(64 * (NT + 1)), ldc, c->stream, c->L, c->d.T, c->rec,
                       c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status);
    return hipGetLastError();
}
template <int M, int NT>
static hipError_t launch_bt_uw(Ctx *c, int pd_stride)
{
    if constexpr (NT <= 3) {
        const int rows = c->n + 1 - 16 * (NT - 1), ncl = rows >= 16 ? 4 : (rows + 3) / 4;
        switch (ncl > 1 ? ncl : 1) {                 // (a state that leaves the last tile empty -- KPILQR_TILED_NT_MIN -- still runs one chunk)
        case 1: return launch_bt_uw2<M, NT, 1>(c, pd_stride);
        case 2: return launch_bt_uw2<M, NT, 2>(c, pd_stride);
        case 3: return launch_bt_uw2<M, NT, 3>(c, pd_stride);
        case 4: return launch_bt_uw2<M, NT, 4>(c, pd_stride);
        }
    }
    return hipErrorInvalidValue;
}

template <int M, int NT, bool A6, int NCL>
static hipError_t launch_bt3(Ctx *c, int pd_stride)
{
    const size_t ldc = backward_col_lds_bytes(NT);
    hipError_t e = hipFuncSetAttribute((const void *)k_backward_tiled_col<M, NT, A6, NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldc);
    if (e != hipSuccess) return e;
    const CostSrc CS = {c->r, c->r_x, c->r_u, c->w_run, c->w_term, c->d.nr};
    hipLaunchKernelGGL((k_backward_tiled_col<M, NT, A6, NCL>), dim3(c->d.batch), dim3(64 * NT), ldc, c->stream, c->L, CS, c->d.T, c->rec,
                       c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status);
    return hipGetLastError();
}
template <int M, int NT, bool A6>
static hipError_t launch_bt2(Ctx *c, int pd_stride)
{
    // four tiles, records materialised: the interleaved products, instantiated per chunk count of the last row tile
    if constexpr (NT == 4) {
        const int rows = c->n + 1 - 16 * (NT - 1), ncl = rows >= 16 ? 4 : (rows + 3) / 4;
        if (c->tune.tiled_uw != 0) switch (ncl) {
        case 1: return launch_bt3<M, NT, A6, 1>(c, pd_stride);
        case 2: return launch_bt3<M, NT, A6, 2>(c, pd_stride);
        case 3: return launch_bt3<M, NT, A6, 3>(c, pd_stride);
        case 4: return launch_bt3<M, NT, A6, 4>(c, pd_stride);
        }
    }
    return launch_bt3<M, NT, A6, 0>(c, pd_stride);
}
template <int M, int NT>
static hipError_t launch_bt(Ctx *c, int pd_stride)
{
    if (NT <= 3 && !c->tiled_a6 && c->tune.tiled_uw != 0) return launch_bt_uw<M, NT>(c, pd_stride);
    return c->tiled_a6 ? launch_bt2<M, NT, true>(c, pd_stride) : launch_bt2<M, NT, false>(c, pd_stride);
}

hipError_t launch_backward_tiled(Ctx *c, int pd_stride)
{
    const int nt = tiled_nt(c->n, c->tune.tiled_nt_min), m = c->d.m;
    if (m == 7) { if (nt == 2) return launch_bt<7, 2>(c, pd_stride); if (nt == 3) return launch_bt<7, 3>(c, pd_stride); if (nt == 4) return launch_bt<7, 4>(c, pd_stride); }
    if (m == 1) { if (nt == 2) return launch_bt<1, 2>(c, pd_stride); if (nt == 3) return launch_bt<1, 3>(c, pd_stride); if (nt == 4) return launch_bt<1, 4>(c, pd_stride); }
    if (m >= 1 && m <= 8) { if (nt == 2) return launch_bt<8, 2>(c, pd_stride); if (nt == 3) return launch_bt<8, 3>(c, pd_stride); if (nt == 4) return launch_bt<8, 4>(c, pd_stride); }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Forward pass (a8).  Z = [dx; alpha; 1] (n+2 rows) x 16 alpha columns is NT row tiles.  One workgroup of
// NT wavefronts per trajectory: wave i owns row tile i of Z+ and of Lc Z (its column of Ya / Lc tiles goes
// global -> registers, prefetched one step ahead), every wave forms the (cheap) control law itself, and the
// Z tiles are exchanged through a double-buffered LDS image with ONE s_barrier per time-step.
// NCL > 0: chunk count of the last row tile at compile time, the two accumulation chains of a step (state cost rows Lc Z, next
// state Ya Z) issued interleaved; NCL = 0: run-time count, one chain after the other (see k_backward_tiled_uw).
template <int NT, bool A6, int NCL = 0>
__global__ void __launch_bounds__(64 * NT)
k_forward_tiled(RecLayout L, CostSrc CS, int T, int n_alpha, const double *__restrict__ rec, const double *__restrict__ Kin,
                const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha)
{
    extern __shared__ __attribute__((aligned(16))) double fsh[];
    double (*zbuf)[NT * TILE] = (double (*)[NT * TILE])fsh;                 // [2][NT * TILE]
    double *upart = fsh + 2 * NT * TILE;                                     // per-wave partials of K dx + alpha k
    double *red = upart + NT * TILE;                                         // [NT * 64]
    double *jpart = red + NT * 64;                                           // A6: per-wave partials of r_x dx  [NT * TILE]
    const int n = L.n, m = L.m, nz2 = n + 2;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wi = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // this wave's row tile (wave-uniform)
    const int b = KP_TILED_TRAJ;
    const int ncu = (m + 3) >> 2;
    auto nchunk = [&](int kt) { const int rows = nz2 - 16 * kt; return rows >= 16 ? 4 : (rows + 3) / 4; };
    const int ncl = nchunk(NT - 1);
    const int ncw = nchunk(wi);

    // per-lane source byte offsets (OOBT = structural zero); p = contraction (state) index, o = output index
    int oKw[4], okw[4], oA[NT][4], oLc[NT][4], oB[4], oLuu[4], olu[4], oub[4];
    double oneT[4], lo[4], hi[4];                    // identity entries of Ya: rows n, n+1 live in row tile tnz only
    const int tnz = n >> 4;
    const int o = 16 * wi + c;
#pragma unroll
    for (int k = 0; k < NT; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pp = 16 * k + 4 * r + q;
            oA[k][r] = (pp < n && o < n) ? 8 * L.a(o, pp) : OOBT;
            oLc[k][r] = (pp < n && o < n) ? 8 * (L.off_lxx + pp * n + o)
                      : (pp == n + 1 && o < n) ? 8 * (L.off_lx + o)
                      : (o == n + 1 && pp < n) ? 8 * (L.off_lx + pp) : OOBT;
        }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        const int pt = 16 * tnz + row;
        oneT[r] = ((pt == n && o == n) || (pt == n + 1 && o == n + 1)) ? 1.0 : 0.0;
        const int pw = 16 * wi + row;                                   // this wave's slice of the state index
        oKw[r] = (pw < n && c < m) ? 8 * (pw * m + c) : OOBT;
        okw[r] = (pw == n && c < m) ? 8 * c : OOBT;
        oB[r] = (row < m && o < n) ? 8 * L.b(o, row) : OOBT;
        oLuu[r] = (row < m && c < m) ? 8 * (L.off_luu + row * m + c) : OOBT;
        olu[r] = (row < m) ? 8 * (L.off_lu + row) : OOBT;
        oub[r] = (row < m) ? 8 * row : OOBT;
        lo[r] = (row < m) ? ctrl_lim[2 * row] : -1.0e300;
        hi[r] = (row < m) ? ctrl_lim[2 * row + 1] : 1.0e300;
    }
    // A6: the candidates are scored on the residuals, sum_k w_k [Jx_k (2 r_k + Jx_k) + Ju_k (2 r_k + Ju_k)] with
    // Jx = r_x dx (each wave forms the partial of its own slice of the state, summed through LDS) and Ju = r_u du,
    // instead of through l_xx, l_x, l_uu, l_u: NT*4 MFMAs per wave-step less, and l_xx is never read.
    const int nr = CS.nr;
    int oRxT[4], oRuT[4], oRr[4];
    double wrun[4], wterm[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        const int pw = 16 * wi + row;
        oRxT[r] = (A6 && pw < n && c < nr) ? 8 * (c * n + pw) : OOBT;          // RxT(p = pw, k = c) = r_x[k][p]
        oRuT[r] = (A6 && row < m && c < nr) ? 8 * (c * m + row) : OOBT;        // RuT(p = row, k = c) = r_u[k][p]
        oRr[r] = (A6 && row < nr) ? 8 * row : OOBT;
        wrun[r] = (A6 && row < nr) ? CS.w_run[row] : 0.0;
        wterm[r] = (A6 && row < nr) ? CS.w_term[row] : 0.0;
    }
    const double *rb = CS.r + (size_t)b * (T + 1) * nr, *rxb = CS.r_x + (size_t)b * (T + 1) * nr * n,
                 *rub = CS.r_u + (size_t)b * (T + 1) * nr * m;
    const double my_alpha = (c < n_alpha) ? alphas[c] : 0.0;
    d4 Zi;                                                               // this wave's tile of Z, kept in registers
    {
        double zr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wi + 4 * r + q;
            zr[r] = (row == n) ? my_alpha : (row == n + 1) ? 1.0 : 0.0;
        }
        Zi.x = zr[0]; Zi.y = zr[1]; Zi.z = zr[2]; Zi.w = zr[3];
        lds_store(zbuf[0] + wi * TILE, lane, Zi);
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double partial = 0.0;

    struct Tiles { d4 Ykw, Ya[NT], Lc[NT], Yb, Luu, lu, ub; double kv; };        // A6: Lc[0] = RxT (own slice), Luu = RuT, lu = r; kv: k' (FSPREAD)
    const int rec_bytes = L.rec * 8; (void)rec_bytes;
    auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, const int *off) -> d4 {
        d4 v; v.x = tbld(rs, off[0]); v.y = tbld(rs, off[1]); v.z = tbld(rs, off[2]); v.w = tbld(rs, off[3]);
        return v;
    };
    auto rs_of = [&](const double *base, size_t step_elems, int t, int bytes) {
        const bool ok = t < T;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + ((size_t)b * T + (ok ? t : 0)) * step_elems), 0, ok ? bytes : 0, 0x00020000);
    };
    auto rs_res = [&](const double *base_b, size_t step_elems, int t, int bytes) {        // residual arrays: base already at trajectory b
        const bool ok = t < T;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base_b + (size_t)(ok ? t : 0) * step_elems), 0, ok ? bytes : 0, 0x00020000);
    };
    // Single-buffered tiles: each group is re-requested for step t+1 right behind its last use in step t.
    // Every wave forms only ITS slice of the control law, P([K' ; k'] rows of tile wi, Z_wi) (4 MFMAs instead of
    // 4*NT); the slices are summed through LDS (one extra barrier), then every wave clamps the same U.
    // Register sets (round 4, the rule of fused_mfma.hip's forward sweep): the tiles of step t sit in set t mod NS and are
    // re-requested for step t + NS right behind their last use.  With ONE set (rounds 1-3) a tile was requested one step ahead:
    // a step of this kernel (~1.1 us at two tiles) is about one trip to HBM, and a wave's memory operations return in order,
    // so the step time WAS that trip.  NS = 2 where the tiles are materialised and the chunk counts compile-time (NCL > 0);
    // every step is instantiated with its set as a compile-time constant (no copies between sets).
    constexpr int NS = (NCL > 0 && !A6) ? KP_FT_SETS : 1;
    // FSPREAD (a6 inside, compile-time chunk counts; round 5, late): the requests for the next step's tiles go out one by one under
    // the products instead of in blocks behind them (the four waves of a trajectory share one address unit: a block of twenty
    // requests per wave stood ~1 600 cycles in front of it), registers that hold structural zeros only are not requested, k rides in
    // ONE request (the lanes of row n) and joins the gain operand at its use instead of behind the request
    constexpr bool FSPREAD = KP_FT_SPREAD && NCL > 0 && (A6 || NT > 2);       // (two tiles, materialised: 3.54 against 3.37 ms on pushing -- stays with the blocks)
    const bool k_here = tnz == wi;
    const int okn = (k_here && q == (n & 3) && c < m) ? 8 * c : OOBT;
    double kmask[4];
#pragma unroll
    for (int r = 0; r < 4; r++) kmask[r] = (r == ((n & 15) >> 2)) ? 1.0 : 0.0;
    (void)okn; (void)kmask;
    Tiles S[NS];
#pragma unroll
    for (int s0 = 0; s0 < NS; s0++) {
        Tiles &cur = S[s0];
        __amdgpu_buffer_rsrc_t rR = rs_of(rec, L.stride, s0, rec_bytes), rK = rs_of(Kin, (size_t)m * n, s0, m * n * 8);
        __amdgpu_buffer_rsrc_t rk = rs_of(kin, m, s0, m * 8), ru = rs_of(u_nom, m, s0, m * 8);
        cur.kv = 0.0;
        if constexpr (FSPREAD) { cur.Ykw = ld4(rK, oKw); cur.kv = tbld(rk, okn); }
        else cur.Ykw = ld4(rK, oKw) + ld4(rk, okw);
#pragma unroll
        for (int k = 0; k < NT; k++) { cur.Ya[k] = ld4(rR, oA[k]); if (!A6) cur.Lc[k] = ld4(rR, oLc[k]); }
        cur.Yb = ld4(rR, oB);
        cur.ub = ld4(ru, oub);
        if (A6) {
            cur.Lc[0] = ld4(rs_res(rxb, (size_t)nr * n, s0, nr * n * 8), oRxT);
            cur.Luu = ld4(rs_res(rub, (size_t)nr * m, s0, nr * m * 8), oRuT);
            cur.lu = ld4(rs_res(rb, nr, s0, nr * 8), oRr);
        } else { cur.Luu = ld4(rR, oLuu); cur.lu = ld4(rR, olu); }
    }
    __syncthreads();

#ifdef KP_CYC_FT
    long long cy[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c0 = __builtin_readcyclecounter(), c1;
#define CYF(i) { c1 = __builtin_readcyclecounter(); cy[i] += c1 - c0; c0 = c1; }
#else
#define CYF(i)
#endif
    auto step = [&](int t, auto set_tag) __attribute__((always_inline)) {
        Tiles &cur = S[decltype(set_tag)::value];
        const int tq = t + NS;                               // the step this set is re-requested for (beyond the horizon: empty descriptors)
        const __amdgpu_buffer_rsrc_t rR = rs_of(rec, L.stride, tq, rec_bytes), rK = rs_of(Kin, (size_t)m * n, tq, m * n * 8);
        const __amdgpu_buffer_rsrc_t rk = rs_of(kin, m, tq, m * 8), ru = rs_of(u_nom, m, tq, m * 8);
        const double *zc = zbuf[t & 1];
        double *zn = zbuf[(t + 1) & 1];
        // ---- this wave's slice of K dx + alpha k -------------------------------------------------------------
        if constexpr (FSPREAD) {
            auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
            const __amdgpu_buffer_rsrc_t rX = rs_res(rxb, (size_t)nr * n, tq, nr * n * 8);
            d4 Yk = cur.Ykw;                                  // K' rows of this wave's slice; + k' in row n
            const double kv = cur.kv;
            Yk.x = __builtin_fma(kmask[0], kv, Yk.x); Yk.y = __builtin_fma(kmask[1], kv, Yk.y);
            Yk.z = __builtin_fma(kmask[2], kv, Yk.z); Yk.w = __builtin_fma(kmask[3], kv, Yk.w);
            d4 Us = zero, Js = zero;
            // (every request unconditional: a request behind a wave-uniform branch meets the register's old value at the join, and the
            // copy that merges them waits for EVERY request in flight -- measured: the last wave's scoring phase 1 560 -> 4 480 cycles)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (r < ncw) {                                // (wave-uniform: only the last row tile is short)
                    Us = MFMA(comp(Yk, r), comp(Zi, r), Us);
                    if (A6) Js = MFMA(comp(cur.Lc[0], r), comp(Zi, r), Js);
                }
                setc(cur.Ykw, r, tbld(rK, oKw[r]));
                if (A6) setc(cur.Lc[0], r, tbld(rX, oRxT[r]));
            }
            cur.kv = tbld(rk, okn);
            lds_store(upart + wi * TILE, lane, Us);
            if (A6) lds_store(jpart + wi * TILE, lane, Js);
        } else {
        lds_store(upart + wi * TILE, lane, Pn(cur.Ykw, Zi, zero, ncw));
        if (A6) lds_store(jpart + wi * TILE, lane, Pn(cur.Lc[0], Zi, zero, ncw));        // this wave's slice of r_x dx
        __builtin_amdgcn_sched_barrier(0);
        cur.Ykw = ld4(rK, oKw) + ld4(rk, okw);
        if (A6) cur.Lc[0] = ld4(rs_res(rxb, (size_t)nr * n, tq, nr * n * 8), oRxT);
        __builtin_amdgcn_sched_barrier(0);
        }
        CYF(0)
        __syncthreads();
        CYF(1)
        // ---- control law + clamp (every wave; :876-890) --------------------------------------------------------
        const d4 ub = cur.ub;
        d4 U = ub;
#pragma unroll
        for (int k = 0; k < NT; k++) U = U + lds_tile(upart + k * TILE, lane);
        d4 Zk[NT];
#pragma unroll
        for (int k = 0; k < NT; k++) Zk[k] = lds_tile(zc + k * TILE, lane);
        __builtin_amdgcn_sched_barrier(0);
        cur.ub = ld4(ru, oub);
        __builtin_amdgcn_sched_barrier(0);
        d4 dU;
        {
            double u;
            u = U.x; if (u > hi[0]) u = hi[0]; if (u < lo[0]) u = lo[0]; U.x = u; dU.x = u - ub.x;
            u = U.y; if (u > hi[1]) u = hi[1]; if (u < lo[1]) u = lo[1]; U.y = u; dU.y = u - ub.y;
            u = U.z; if (u > hi[2]) u = hi[2]; if (u < lo[2]) u = lo[2]; U.z = u; dU.z = u - ub.z;
            u = U.w; if (u > hi[3]) u = hi[3]; if (u < lo[3]) u = lo[3]; U.w = u; dU.w = u - ub.w;
        }
        if (wi == NT - 1) {                        // control cost and U_alpha by the LAST wave (its row tile is the shortest)
            if (U_alpha && c < n_alpha) {
                double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
                const double uv[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
                for (int r = 0; r < 4; r++) { const int row = 4 * r + q; if (row < m) Ua[row] = uv[r]; }
            }
            if (A6) {
                d4 Jx = zero;
#pragma unroll
                for (int k = 0; k < NT; k++) Jx = Jx + lds_tile(jpart + k * TILE, lane);
                const d4 Ju = Pn(cur.Luu, dU, zero, ncu);
                const d4 r2 = cur.lu + cur.lu;
                const bool last = (t == T - 1);                        // terminal weights (Optimiser.cpp:209-211)
                const double w0 = last ? wterm[0] : wrun[0], w1 = last ? wterm[1] : wrun[1];
                const double w2 = last ? wterm[2] : wrun[2], w3 = last ? wterm[3] : wrun[3];
                partial += w0 * (Jx.x * (r2.x + Jx.x) + Ju.x * (r2.x + Ju.x)) + w1 * (Jx.y * (r2.y + Jx.y) + Ju.y * (r2.y + Ju.y))
                         + w2 * (Jx.z * (r2.z + Jx.z) + Ju.z * (r2.z + Ju.z)) + w3 * (Jx.w * (r2.w + Jx.w) + Ju.w * (r2.w + Ju.w));
                __builtin_amdgcn_sched_barrier(0);
                cur.Luu = ld4(rs_res(rub, (size_t)nr * m, tq, nr * m * 8), oRuT);
                cur.lu = ld4(rs_res(rb, nr, tq, nr * 8), oRr);
                __builtin_amdgcn_sched_barrier(0);
            } else {
            const d4 Wu = Pn(cur.Luu, dU, zero, ncu);
            partial += dU.x * (0.5 * Wu.x + cur.lu.x) + dU.y * (0.5 * Wu.y + cur.lu.y)
                     + dU.z * (0.5 * Wu.z + cur.lu.z) + dU.w * (0.5 * Wu.w + cur.lu.w);
            __builtin_amdgcn_sched_barrier(0);
            cur.Luu = ld4(rR, oLuu); cur.lu = ld4(rR, olu);
            __builtin_amdgcn_sched_barrier(0);
            }
        }
        CYF(2)
        // ---- state cost rows of this tile, then the linearised dynamics for this tile ------------------------------
        d4 Wz = zero, Zn = zero;
        if constexpr (FSPREAD) {
            auto nck = [](int kt) { return kt < NT - 1 ? 4 : NCL; };
            auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
#pragma unroll
            for (int k = 0; k < NT; k++) {
                d4 Ya = cur.Ya[k];
                if (k == tnz) { Ya.x += oneT[0]; Ya.y += oneT[1]; Ya.z += oneT[2]; Ya.w += oneT[3]; }
#pragma unroll
                for (int r = 0; r < nck(k); r++) {
                    Zn = MFMA(comp(Ya, r), comp(Zk[k], r), Zn);
                    if (!A6) Wz = MFMA(comp(cur.Lc[k], r), comp(Zk[k], r), Wz);
                    setc(cur.Ya[k], r, tbld(rR, oA[k][r]));
                    if (!A6) setc(cur.Lc[k], r, tbld(rR, oLc[k][r]));
                    __builtin_amdgcn_sched_group_barrier(0x008, A6 ? 1 : 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, A6 ? 1 : 2, 0);
                }
#pragma unroll
                for (int r = nck(k); r < 4; r++) {            // (chunks of the last row tile that hold no rows of z: structural zeros)
                    setc(cur.Ya[k], r, 0.0);
                    if (!A6) setc(cur.Lc[k], r, 0.0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
                if (r < ncu) Zn = MFMA(comp(cur.Yb, r), comp(dU, r), Zn);
                setc(cur.Yb, r, tbld(rR, oB[r]));
            }
            if (!A6) partial += 0.5 * (Zi.x * Wz.x + Zi.y * Wz.y + Zi.z * Wz.z + Zi.w * Wz.w);
        } else if constexpr (NCL > 0 && !A6) {
            auto nck = [](int kt) { return kt < NT - 1 ? 4 : NCL; };
            auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
#pragma unroll
            for (int k = 0; k < NT; k++) {
                d4 Ya = cur.Ya[k];
                if (k == tnz) { Ya.x += oneT[0]; Ya.y += oneT[1]; Ya.z += oneT[2]; Ya.w += oneT[3]; }
#pragma unroll
                for (int r = 0; r < nck(k); r++) {
                    Zn = MFMA(comp(Ya, r), comp(Zk[k], r), Zn);
                    Wz = MFMA(comp(cur.Lc[k], r), comp(Zk[k], r), Wz);
                }
            }
            Zn = Pn(cur.Yb, dU, Zn, ncu);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NT; k++) cur.Lc[k] = ld4(rR, oLc[k]);
            __builtin_amdgcn_sched_barrier(0);
            partial += 0.5 * (Zi.x * Wz.x + Zi.y * Wz.y + Zi.z * Wz.z + Zi.w * Wz.w);
        } else {
        if (!A6) {
#pragma unroll
            for (int k = 0; k < NT; k++) Wz = Pk<NT>(k, cur.Lc[k], Zk[k], Wz, ncl);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NT; k++) cur.Lc[k] = ld4(rR, oLc[k]);
            __builtin_amdgcn_sched_barrier(0);
            partial += 0.5 * (Zi.x * Wz.x + Zi.y * Wz.y + Zi.z * Wz.z + Zi.w * Wz.w);
        }
#pragma unroll
        for (int k = 0; k < NT; k++) {
            d4 Ya = cur.Ya[k];
            if (k == tnz) { Ya.x += oneT[0]; Ya.y += oneT[1]; Ya.z += oneT[2]; Ya.w += oneT[3]; }
            Zn = Pk<NT>(k, Ya, Zk[k], Zn, ncl);
        }
        Zn = Pn(cur.Yb, dU, Zn, ncu);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!FSPREAD) {
#pragma unroll
            for (int k = 0; k < NT; k++) cur.Ya[k] = ld4(rR, oA[k]);
            cur.Yb = ld4(rR, oB);
        }
        __builtin_amdgcn_sched_barrier(0);
        CYF(3)
        Zi = Zn;
        lds_store(zn + wi * TILE, lane, Zn);
        CYF(4)
        __syncthreads();
        CYF(5)
    };
    if constexpr (NS == 1) {
        for (int t = 0; t < T; t++) step(t, std::integral_constant<int, 0>{});
    } else {
        static_assert(NS == 2, "the time loop below is written for two sets");
        int t = 0;
        for (; t + 2 <= T; t += 2) { step(t, std::integral_constant<int, 0>{}); step(t + 1, std::integral_constant<int, 1>{}); }
        if (t < T) step(t, std::integral_constant<int, 0>{});
    }
#ifdef KP_CYC_FT
    if (b == 0 && lane == 0) printf("fwd wave %d: slice+requests %lld | wait1 %lld | law+clamp(+last wave's cost) %lld | products+requests %lld | Zn store %lld | wait2 %lld\n", wi,
                                    cy[0] / T, cy[1] / T, cy[2] / T, cy[3] / T, cy[4] / T, cy[5] / T);
#endif
    // column sums: over the q lane groups, then over the waves (fixed order: reproducible)
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

// ---------------------------------------------------------------------------------------------------------------
// Forward pass, STATE / COST wave groups (round 4; materialised tiles, two row tiles; while the 2 NT waves of a trajectory each
// find a SIMD: 2 NT x batch <= #SIMDs -- configs[2]: Panda pushing, 64 trajectories, 4 of a CU's SIMDs instead of 2).  Only the
// state recursion is serial in time (as in fused_mfma.hip's k_forward_fused_sc): state wave i keeps row tile i of Z and runs
//     slice of K dx + alpha k | barrier | sum of the slices, clamp | Z' = A dx + B du | barrier          (4 + 4 NT + 2 MFMAs)
// and publishes Z_t (a three-slot LDS ring) and dU_t; cost wave i, ONE STEP BEHIND, scores the candidates' row tile i --
// Wz = Lc Z (4 NT MFMAs), 0.5 Z'Wz, and on the last cost wave the control cost dU'(0.5 l_uu dU + l_u) -- from the published
// state: the 4 NT cost products, their 4 NT tile loads and the accumulation leave the recursion's wave (the products of one wave
// do not overlap with its own VALU work; a second SIMD does them for free).  All waves pass the same two s_barrier per tick; a
// cost wave passes the first one AT ONCE (with its scoring in front of it the recursion waited for the scoring: 3.64 against
// 3.34 ms) and scores under the state waves' long phase.  Role-specialised time loops (a join of the roles behind a request
// costs register copies and a wait for every load in flight: DESIGN.md, u-wave).
// The state role is instantiated per wave (WI) and per chunk count of the last row tile (NCL) and of the controls (NCU): a
// register that holds structural zeros only -- rows beyond n + 2 in the last tile, rows beyond num_ctrl of a control operand -- is
// not loaded, not stored, not clamped (an out-of-range buffer load moves no data but still costs its ~30 cycles of issue on a lone
// wave; 24 -> 15 loads per state wave and step at n = 20, m = 7).
template <int NC> __device__ __forceinline__ d4 Pc(const d4 &Y, const d4 &X, d4 acc)
{
    acc = MFMA(Y.x, X.x, acc);
    if constexpr (NC > 1) acc = MFMA(Y.y, X.y, acc);
    if constexpr (NC > 2) acc = MFMA(Y.z, X.z, acc);
    if constexpr (NC > 3) acc = MFMA(Y.w, X.w, acc);
    return acc;
}
template <int NC> __device__ __forceinline__ d4 lds_tile_n(const double *t, int lane)
{
    d4 v = {0.0, 0.0, 0.0, 0.0};
    v.x = t[lane];
    if constexpr (NC > 1) v.y = t[64 + lane];
    if constexpr (NC > 2) v.z = t[128 + lane];
    if constexpr (NC > 3) v.w = t[192 + lane];
    return v;
}
template <int NC> __device__ __forceinline__ void lds_store_n(double *t, int lane, const d4 &v)
{
    t[lane] = v.x;
    if constexpr (NC > 1) t[64 + lane] = v.y;
    if constexpr (NC > 2) t[128 + lane] = v.z;
    if constexpr (NC > 3) t[192 + lane] = v.w;
}
template <int NC> __device__ __forceinline__ d4 ld4n(__amdgpu_buffer_rsrc_t rs, const int *off)
{
    d4 v = {0.0, 0.0, 0.0, 0.0};
    v.x = tbld(rs, off[0]);
    if constexpr (NC > 1) v.y = tbld(rs, off[1]);
    if constexpr (NC > 2) v.z = tbld(rs, off[2]);
    if constexpr (NC > 3) v.w = tbld(rs, off[3]);
    return v;
}
// the same with a wave-uniform byte offset in the loads' scalar operand: ONE descriptor per array and trajectory, the step selected
// by an SGPR, instead of a descriptor rebuilt (64-bit address arithmetic) per array and step
__device__ __forceinline__ double tblds(__amdgpu_buffer_rsrc_t r, int byte_off, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, soff, 0));
}
template <int NC> __device__ __forceinline__ d4 ld4ns(__amdgpu_buffer_rsrc_t rs, const int *off, int soff)
{
    d4 v = {0.0, 0.0, 0.0, 0.0};
    v.x = tblds(rs, off[0], soff);
    if constexpr (NC > 1) v.y = tblds(rs, off[1], soff);
    if constexpr (NC > 2) v.z = tblds(rs, off[2], soff);
    if constexpr (NC > 3) v.w = tblds(rs, off[3], soff);
    return v;
}

template <int NT, int WI, int NCL, int NCU>
__device__ __forceinline__ void ft_state_role(double *zring, double *upart, double *dubuf, RecLayout L, int T, int n_alpha, int b,
                                              const double *__restrict__ rec, const double *__restrict__ Kin, const double *__restrict__ kin,
                                              const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                                              const double *__restrict__ alphas, double *__restrict__ U_alpha)
{
    constexpr int wi = WI;
    constexpr int NCW = WI < NT - 1 ? 4 : NCL;                              // chunks of this wave's own row tile
    const int n = L.n, m = L.m;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int tnz = n >> 4;
    const int o = 16 * wi + c;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const int rec_bytes = L.rec * 8; (void)rec_bytes;
    auto rs_of = [&](const double *base, size_t step_elems, int t, int bytes) {
        const bool ok = t < T;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + ((size_t)b * T + (ok ? t : 0)) * step_elems), 0, ok ? bytes : 0, 0x00020000);
    };
    int oKw[4], oA[NT][4], oB[4], oub[4];
    double oneT[4], lo[NCU], hi[NCU];
#pragma unroll
    for (int k = 0; k < NT; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pp = 16 * k + 4 * r + q;
            oA[k][r] = (pp < n && o < n) ? 8 * L.a(o, pp) : OOBT;
        }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        const int pt = 16 * tnz + row;
        oneT[r] = ((pt == n && o == n) || (pt == n + 1 && o == n + 1)) ? 1.0 : 0.0;
        const int pw = 16 * wi + row;
        oKw[r] = (pw < n && c < m) ? 8 * (pw * m + c) : OOBT;
        oB[r] = (row < m && o < n) ? 8 * L.b(o, row) : OOBT;
        oub[r] = (row < m) ? 8 * row : OOBT;
        if (r < NCU) {
            lo[r] = (row < m) ? ctrl_lim[2 * row] : -1.0e300;
            hi[r] = (row < m) ? ctrl_lim[2 * row + 1] : 1.0e300;
        }
    }
    // k rides in row n of the gain operand: ONE load (the lanes of that row), added into the register that holds the row
    const bool k_here = tnz == wi;
    const int rk = (n & 15) >> 2;
    const int okn = (k_here && q == (n & 3) && c < m) ? 8 * c : OOBT;
    double kmask[4];                                                     // 1 in the register that holds row n (no branch in the loop)
#pragma unroll
    for (int r = 0; r < 4; r++) kmask[r] = (r == rk) ? 1.0 : 0.0;
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
        lds_store(zring + wi * TILE, lane, Zi);
    }
    struct STiles { d4 Ykw, Ya[NT], Yb, ub; double kk; } cur;
    // one descriptor per array for the whole trajectory; the step goes into the loads' scalar offset (behind the last step: step
    // T-1 again, never used)
    (void)rs_of;
    const __amdgpu_buffer_rsrc_t rRec = __builtin_amdgcn_make_buffer_rsrc((void *)(rec + (size_t)b * T * L.stride), 0, (int)((size_t)T * L.stride * 8 < 0x7fffff00u ? (size_t)T * L.stride * 8 : 0x7fffff00u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rKt = __builtin_amdgcn_make_buffer_rsrc((void *)(Kin + (size_t)b * T * m * n), 0, T * m * n * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rkt = __builtin_amdgcn_make_buffer_rsrc((void *)(kin + (size_t)b * T * m), 0, T * m * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rut = __builtin_amdgcn_make_buffer_rsrc((void *)(u_nom + (size_t)b * T * m), 0, T * m * 8, 0x00020000);
    const int recB = L.stride * 8;
    auto request_A = [&](int t) {
        const int so = (t < T ? t : T - 1) * recB;
#pragma unroll
        for (int k = 0; k < NT - 1; k++) cur.Ya[k] = ld4ns<4>(rRec, oA[k], so);
        cur.Ya[NT - 1] = ld4ns<NCL>(rRec, oA[NT - 1], so);
    };
    auto request_B = [&](int t) {
        const int so = (t < T ? t : T - 1) * recB;
        cur.Yb = ld4ns<NCU>(rRec, oB, so);
    };
    {
        cur.Ykw = ld4ns<NCW>(rKt, oKw, 0); cur.kk = tblds(rkt, okn, 0);
        request_A(0); request_B(0);
        cur.ub = ld4ns<NCU>(rut, oub, 0);
    }
    __syncthreads();                                                     // Z_0 is published
    int slot = 0;                                                        // t mod 3
    for (int t = 0; t < T; t++) {
        const int tn1 = t + 1 < T ? t + 1 : T - 1;
        const int sK = tn1 * m * n * 8, sk = tn1 * m * 8;
        const double *zc = zring + slot * NT * TILE;
        const int nslot = slot == 2 ? 0 : slot + 1;
        double *zn = zring + nslot * NT * TILE;
        d4 Yk = cur.Ykw;
        {
            const double kk = cur.kk;
            Yk.x = __builtin_fma(kmask[0], kk, Yk.x);
            if constexpr (NCW > 1) Yk.y = __builtin_fma(kmask[1], kk, Yk.y);
            if constexpr (NCW > 2) Yk.z = __builtin_fma(kmask[2], kk, Yk.z);
            if constexpr (NCW > 3) Yk.w = __builtin_fma(kmask[3], kk, Yk.w);
        }
        // Z_t has been in its ring slot since the barrier that ended step t-1: everything that needs only Z_t -- this wave's
        // slice of the control law AND its row tile of A dx -- is formed in front of the mid-step barrier (two independent MFMA
        // chains back to back, the slice's LDS store under the second one); behind the barrier the chain is clamp -> B du only.
        // (every tile through LDS, the wave's own one too: taking that one from the registers it sits in measured SLOWER, 2.99 against
        // 2.92 ms on pushing, 2.78 against 2.67 on walker -- instruction placement, as so often in these sweeps)
        d4 Zk[NT];
#pragma unroll
        for (int k = 0; k < NT - 1; k++) Zk[k] = lds_tile(zc + k * TILE, lane);
        Zk[NT - 1] = lds_tile_n<NCL>(zc + (NT - 1) * TILE, lane);
        d4 Zn = zero;
        if constexpr (KP_SC_SPREAD) {
            // the next step's gain rows and A tiles, register by register behind the product that read the register (round 5, late:
            // the waves of a trajectory share one address unit, and a block of thirteen requests stood in front of the barrier)
            auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
            const int soA = tn1 * recB;
            d4 Us = zero;
#pragma unroll
            for (int r = 0; r < NCW; r++) {
                Us = MFMA(comp(Yk, r), comp(Zi, r), Us);
                setc(cur.Ykw, r, tblds(rKt, oKw[r], sK));
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            cur.kk = tblds(rkt, okn, sk);
            lds_store_n<NCU>(upart + wi * TILE, lane, Us);
#pragma unroll
            for (int k = 0; k < NT; k++) {
                d4 Ya = cur.Ya[k];
                if (k == tnz) { Ya.x += oneT[0]; Ya.y += oneT[1]; Ya.z += oneT[2]; Ya.w += oneT[3]; }
#pragma unroll
                for (int r = 0; r < (k < NT - 1 ? 4 : NCL); r++) {
                    Zn = MFMA(comp(Ya, r), comp(Zk[k], r), Zn);
                    setc(cur.Ya[k], r, tblds(rRec, oA[k][r], soA));
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        } else {
        lds_store_n<NCU>(upart + wi * TILE, lane, Pc<NCW>(Yk, Zi, zero));      // this wave's slice of K dx + alpha k (rows < num_ctrl)
#pragma unroll
        for (int k = 0; k < NT; k++) {
            d4 Ya = cur.Ya[k];
            if (k == tnz) { Ya.x += oneT[0]; Ya.y += oneT[1]; Ya.z += oneT[2]; Ya.w += oneT[3]; }
            if (k < NT - 1) Zn = Pc<4>(Ya, Zk[k], Zn); else Zn = Pc<NCL>(Ya, Zk[k], Zn);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur.Ykw = ld4ns<NCW>(rKt, oKw, sK); cur.kk = tblds(rkt, okn, sk);
        request_A(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---- control law + clamp (every state wave; :876-890) ----------------------------------------------------------
        const d4 ub = cur.ub;
        d4 U = ub;
#pragma unroll
        for (int k = 0; k < NT; k++) U = U + lds_tile_n<NCU>(upart + k * TILE, lane);
        __builtin_amdgcn_sched_barrier(0);
        cur.ub = ld4ns<NCU>(rut, oub, sk);
        __builtin_amdgcn_sched_barrier(0);
        d4 dU = zero;
        {
            double u;
            u = U.x; if (u > hi[0]) u = hi[0]; if (u < lo[0]) u = lo[0]; U.x = u; dU.x = u - ub.x;
            if constexpr (NCU > 1) { u = U.y; if (u > hi[1]) u = hi[1]; if (u < lo[1]) u = lo[1]; U.y = u; dU.y = u - ub.y; }
            if constexpr (NCU > 2) { u = U.z; if (u > hi[2]) u = hi[2]; if (u < lo[2]) u = lo[2]; U.z = u; dU.z = u - ub.z; }
            if constexpr (NCU > 3) { u = U.w; if (u > hi[3]) u = hi[3]; if (u < lo[3]) u = lo[3]; U.w = u; dU.w = u - ub.w; }
        }
        Zn = Pc<NCU>(cur.Yb, dU, Zn);
        __builtin_amdgcn_sched_barrier(0);
        request_B(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WI == 0) lds_store_n<NCU>(dubuf + (t & 1) * TILE, lane, dU);        // for the control cost, one tick later
        if constexpr (WI == NT - 1) {
            if (U_alpha && c < n_alpha) {
                double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
                const double uv[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
                for (int r = 0; r < NCU; r++) { const int row = 4 * r + q; if (row < m) Ua[row] = uv[r]; }
            }
        }
        Zi = Zn;
        lds_store_n<NCW>(zn + wi * TILE, lane, Zn);
        slot = nslot;
        __syncthreads();
    }
    __syncthreads(); __syncthreads();                                    // the tick in which the cost waves score step T-1
}

template <int NT, int NCL, int NCU>
__global__ void __launch_bounds__(128 * NT)
k_forward_tiled_sc(RecLayout L, int T, int n_alpha, const double *__restrict__ rec, const double *__restrict__ Kin,
                   const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                   const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha)
{
    static_assert(NT == 2, "the role dispatch below is written for two row tiles");
    extern __shared__ __attribute__((aligned(16))) double fsh[];
    double *zring = fsh;                                                     // [3][NT * TILE]: Z_t in slot t mod 3
    double *upart = zring + 3 * NT * TILE;                                   // per-wave partials of K dx + alpha k
    double *dubuf = upart + NT * TILE;                                       // [2][TILE]: dU_t in slot t & 1
    double *red = dubuf + 2 * TILE;                                          // [NT * 64]
    const int n = L.n, m = L.m;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_state = wv < NT;
    const int wi = is_state ? wv : wv - NT;                                  // this wave's row tile (wave-uniform)
    const int b = KP_TILED_TRAJ;
    // a tile whose every register is dead (m > 4 NCU cannot happen: the launcher picks NCU from m) keeps the LDS images clean:
    // the slots read with fewer registers than a tile has are the ones written with as few
    if (is_state) {
        if (wi == 0) ft_state_role<NT, 0, NCL, NCU>(zring, upart, dubuf, L, T, n_alpha, b, rec, Kin, kin, u_nom, ctrl_lim, alphas, U_alpha);
        else ft_state_role<NT, 1, NCL, NCU>(zring, upart, dubuf, L, T, n_alpha, b, rec, Kin, kin, u_nom, ctrl_lim, alphas, U_alpha);
    } else {
        // ---- scoring, one tick behind ------------------------------------------------------------------------------------------
        const int o = 16 * wi + c;
        const d4 zero = {0.0, 0.0, 0.0, 0.0};
        const int rec_bytes = L.rec * 8;
        auto ld4 = [&](__amdgpu_buffer_rsrc_t rs, const int *off) -> d4 {
            d4 v; v.x = tbld(rs, off[0]); v.y = tbld(rs, off[1]); v.z = tbld(rs, off[2]); v.w = tbld(rs, off[3]);
            return v;
        };
        auto rs_of = [&](const double *base, size_t step_elems, int t, int bytes) {
            const bool ok = t < T;
            return __builtin_amdgcn_make_buffer_rsrc((void *)(base + ((size_t)b * T + (ok ? t : 0)) * step_elems), 0, ok ? bytes : 0, 0x00020000);
        };
        int oLc[NT][4], oLuu[4], olu[4];
#pragma unroll
        for (int k = 0; k < NT; k++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int pp = 16 * k + 4 * r + q;
                oLc[k][r] = (pp < n && o < n) ? 8 * (L.off_lxx + pp * n + o)
                          : (pp == n + 1 && o < n) ? 8 * (L.off_lx + o)
                          : (o == n + 1 && pp < n) ? 8 * (L.off_lx + pp) : OOBT;
            }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + q;
            oLuu[r] = (wi == NT - 1 && row < m && c < m) ? 8 * (L.off_luu + row * m + c) : OOBT;
            olu[r] = (wi == NT - 1 && row < m) ? 8 * (L.off_lu + row) : OOBT;
        }
        struct CTiles { d4 Lc[NT], Luu, lu; } cur;
        const __amdgpu_buffer_rsrc_t rRec = __builtin_amdgcn_make_buffer_rsrc((void *)(rec + (size_t)b * T * L.stride), 0, (int)((size_t)T * L.stride * 8 < 0x7fffff00u ? (size_t)T * L.stride * 8 : 0x7fffff00u), 0x00020000);
        const int recB = L.stride * 8;
        {
            __amdgpu_buffer_rsrc_t rR = rs_of(rec, L.stride, 0, rec_bytes);
#pragma unroll
            for (int k = 0; k < NT - 1; k++) cur.Lc[k] = ld4(rR, oLc[k]);
            cur.Lc[NT - 1] = ld4n<NCL>(rR, oLc[NT - 1]);
            cur.Luu = ld4n<NCU>(rR, oLuu); cur.lu = ld4n<NCU>(rR, olu);
        }
        double partial = 0.0;
        __syncthreads();
        __syncthreads(); __syncthreads();                                    // tick 0: the state waves run step 0
        int slot = 0;
        for (int t = 0; t < T; t++) {
            // the tick's FIRST barrier at once: it is the one the state waves reach behind their slice product (4 MFMAs), and
            // nothing of this wave's work must stand in front of it; the scoring runs under the state waves' long phase
            __syncthreads();
            const int so = (t + 1 < T ? t + 1 : T - 1) * recB;
            const double *zc = zring + slot * NT * TILE;
            slot = slot == 2 ? 0 : slot + 1;
            d4 Zk[NT];
#pragma unroll
            for (int k = 0; k < NT - 1; k++) Zk[k] = lds_tile(zc + k * TILE, lane);
            Zk[NT - 1] = lds_tile_n<NCL>(zc + (NT - 1) * TILE, lane);
            const d4 dU = lds_tile_n<NCU>(dubuf + (t & 1) * TILE, lane);
            d4 Wz = zero;
            d4 Wu = zero;
            const d4 lu = cur.lu;
            if constexpr (KP_SC_SPREAD) {
                auto comp = [](const d4 &v, int r) { return r == 0 ? v.x : r == 1 ? v.y : r == 2 ? v.z : v.w; };
#pragma unroll
                for (int k = 0; k < NT; k++)
#pragma unroll
                    for (int r = 0; r < (k < NT - 1 ? 4 : NCL); r++) {
                        Wz = MFMA(comp(cur.Lc[k], r), comp(Zk[k], r), Wz);
                        setc(cur.Lc[k], r, tblds(rRec, oLc[k][r], so));
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
#pragma unroll
                for (int r = 0; r < NCU; r++) {
                    Wu = MFMA(comp(cur.Luu, r), comp(dU, r), Wu);
                    setc(cur.Luu, r, tblds(rRec, oLuu[r], so));
                    setc(cur.lu, r, tblds(rRec, olu[r], so));
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
            } else {
#pragma unroll
            for (int k = 0; k < NT - 1; k++) Wz = Pc<4>(cur.Lc[k], Zk[k], Wz);
            Wz = Pc<NCL>(cur.Lc[NT - 1], Zk[NT - 1], Wz);
            Wu = Pc<NCU>(cur.Luu, dU, zero);                                 // (l_uu, l_u: the last cost wave's; zeros elsewhere)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NT - 1; k++) cur.Lc[k] = ld4ns<4>(rRec, oLc[k], so);
            cur.Lc[NT - 1] = ld4ns<NCL>(rRec, oLc[NT - 1], so);
            cur.Luu = ld4ns<NCU>(rRec, oLuu, so); cur.lu = ld4ns<NCU>(rRec, olu, so);
            __builtin_amdgcn_sched_barrier(0);
            }
            d4 Zi = Zk[0];                                                   // this wave's own row tile (wi is wave-uniform)
#pragma unroll
            for (int k = 1; k < NT; k++) if (wi == k) Zi = Zk[k];
            partial += 0.5 * (Zi.x * Wz.x + Zi.y * Wz.y + Zi.z * Wz.z + Zi.w * Wz.w);
            if (wi == NT - 1)
                partial += dU.x * (0.5 * Wu.x + lu.x) + dU.y * (0.5 * Wu.y + lu.y) + dU.z * (0.5 * Wu.z + lu.z) + dU.w * (0.5 * Wu.w + lu.w);
            __syncthreads();
        }
        // column sums: over the q lane groups, then over the cost waves (fixed order: reproducible)
        partial += __shfl_xor(partial, 16);
        partial += __shfl_xor(partial, 32);
        red[wi * 64 + lane] = partial;
    }
    __syncthreads();
    if (!is_state && wi == 0 && q == 0 && c < n_alpha) {
        double sum = 0.0;
        for (int w = 0; w < NT; w++) sum += red[w * 64 + lane];
        cost_pred[(size_t)b * n_alpha + c] = sum;
    }
}

bool forward_tiled_supported(int n, int m, int n_alpha, int nt_min)
{
    const int nt = tiled_nt(n, nt_min);
    return nt >= 2 && nt <= 4 && m <= 16 && n_alpha <= 16;
}

template <int NT, bool A6, int NCL>
static hipError_t launch_ft3(Ctx *c, double *U_alpha_dev)
{
    const CostSrc CS = {c->r, c->r_x, c->r_u, c->w_run, c->w_term, c->d.nr};
    const size_t lds = sizeof(double) * ((size_t)(3 + (A6 ? 1 : 0)) * NT * TILE + NT * 64);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_forward_tiled<NT, A6, NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_forward_tiled<NT, A6, NCL>), dim3(c->d.batch), dim3(64 * NT), lds, c->stream, c->L, CS, c->d.T,
                       c->d.n_alpha, c->rec, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev);
    return hipGetLastError();
}
template <int NT, bool A6>
static hipError_t launch_ft2(Ctx *c, double *U_alpha_dev)
{
    // compile-time chunk count of the last row tile: two row tiles materialised -- the interleaved chains with two register sets (pushing
    // 3.52 -> 3.41 ms, walker 3.40 -> 3.19; with three tiles that code was SLOWER, 4.98 -> 5.56 ms on light clutter n=38); everything
    // else -- the requests one by one under the products (round 5: light clutter 5.04 -> 4.61, configs[4] 12.5 -> 9.15 ms)
    if constexpr (KP_FT_SPREAD || (!A6 && NT == 2)) {
        const int rows = c->n + 2 - 16 * (NT - 1), ncl = rows >= 16 ? 4 : (rows + 3) / 4;
        if (c->tune.tiled_uw != 0) switch (ncl > 1 ? ncl : 1) {
        case 1: return launch_ft3<NT, A6, 1>(c, U_alpha_dev);
        case 2: return launch_ft3<NT, A6, 2>(c, U_alpha_dev);
        case 3: return launch_ft3<NT, A6, 3>(c, U_alpha_dev);
        case 4: return launch_ft3<NT, A6, 4>(c, U_alpha_dev);
        }
    }
    return launch_ft3<NT, A6, 0>(c, U_alpha_dev);
}
template <int NT, int NCL, int NCU>
static hipError_t launch_ft_sc2(Ctx *c, double *U_alpha_dev)
{
    const size_t lds = sizeof(double) * ((size_t)(4 * NT + 2) * TILE + NT * 64);
    // the LDS images are read with the register counts they are written with; what lies beyond is never read -- but start clean
    hipLaunchKernelGGL((k_forward_tiled_sc<NT, NCL, NCU>), dim3(c->d.batch), dim3(128 * NT), lds, c->stream, c->L, c->d.T, c->d.n_alpha, c->rec, c->K, c->k,
                       c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev);
    return hipGetLastError();
}
template <int NT>
static hipError_t launch_ft_sc(Ctx *c, double *U_alpha_dev)
{
    // (a state smaller than one tile run on two: the second tile is all structural zeros -- one chunk of them)
    const int rows = c->n + 2 - 16 * (NT - 1), ncl = rows >= 16 ? 4 : rows < 1 ? 1 : (rows + 3) / 4, ncu = (c->d.m + 3) / 4;
#define KP_SC(NCL_, NCU_) if (ncl == NCL_ && ncu == NCU_) return launch_ft_sc2<NT, NCL_, NCU_>(c, U_alpha_dev);
    KP_SC(1, 1) KP_SC(2, 1) KP_SC(3, 1) KP_SC(4, 1) KP_SC(1, 2) KP_SC(2, 2) KP_SC(3, 2) KP_SC(4, 2)
#undef KP_SC
    return hipErrorInvalidValue;
}

// The state / cost wave groups (k_forward_tiled_sc): materialised tiles, two row tiles, while every one of the 2 NT waves of a
// trajectory finds a SIMD of its own (configs[2]: 64 trajectories x 4 waves on 1024 SIMDs).  KPILQR_TILED_FSC = 0 | 1 overrides.
bool forward_tiled_sc_selected(const Ctx *c)
{
    const int nt = tiled_nt(c->n, c->tune.tiled_nt_min);
    if (nt != 2 || c->tiled_a6 || c->d.m > 8) return false;
    if (c->tune.tiled_fsc >= 0) return c->tune.tiled_fsc != 0;
    return 2 * nt * c->d.batch <= c->n_simd;
}

template <int NT>
static hipError_t launch_ft(Ctx *c, double *U_alpha_dev)
{
    if constexpr (NT == 2) { if (forward_tiled_sc_selected(c)) return launch_ft_sc<NT>(c, U_alpha_dev); }
    return c->tiled_a6 ? launch_ft2<NT, true>(c, U_alpha_dev) : launch_ft2<NT, false>(c, U_alpha_dev);
}

hipError_t launch_forward_tiled(Ctx *c, double *U_alpha_dev)
{
    const int nt = tiled_nt(c->n, c->tune.tiled_nt_min);
    if (nt == 2) return launch_ft<2>(c, U_alpha_dev);
    if (nt == 3) return launch_ft<3>(c, U_alpha_dev);
    if (nt == 4) return launch_ft<4>(c, U_alpha_dev);
    return hipErrorInvalidValue;
}

}  // namespace kpilqr
