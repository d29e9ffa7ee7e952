// riccati_mfma.hip -- backward pass (a7) for n+1 <= 16: ONE wavefront per trajectory, the value
// function resident in registers, every per-step product on the FP64 matrix core
// (v_mfma_f64_16x16x4_f64), the small Q_uu solve on the VALU with its operands broadcast through LDS.
//
// Reference: iLQR::BackwardsPassQuuRegularisation + CheckMatrixPD, src/Optimiser/iLQR.cpp:535-670.
//
// Formulation.  With z = [dx; 1] the first-order terms ride in row/column n of
// 16x16 tiles:
//     V' = [V_xx V_x; V_x' 0]   Fz = [A 0; 0 1]   Fu = [B; 0]   Lzz = [l_xx l_x; l_x' 0]   Luz = [0 l_u]
//     Tz = V' Fz,  Tu = V' Fu
//     Qzz = Lzz + Fz' Tz        (= [Q_xx Q_x; Q_x' *])
//     Quz = Luz + Fu' Tz        (= [Q_ux Q_u])
//     Quu = l_uu + Fu' Tu
//     K'  = -(Quu + lambda I)^-1 Quz          (= [K k])
//     V'  = Qzz - K''(Quu + 2 lambda I) K'     (= Qzz + K''Quu K' + K''Quz + Quz'K' of iLQR.cpp:606-607, both
//                                               lines at once, after substituting Quz = -(Quu + lambda I) K'),
//                                               then (V'+V'')/2
//     delta_J -= lambda k'k                    (= k'Q_uu k + k'Q_u of iLQR.cpp:612-613, same substitution)
// Tiles live in the MFMA accumulator ("D") layout: lane (c = lane&15, q = lane>>4), register r holds
// element (row 4r+q, col c).  In that layout a tile is directly the B operand of the next MFMA (k-chunk
// r = register r) and, used as the A operand, it is its own transpose.  So the single primitive is
//     P(Y, X) = Y' X      (sum over the shared ROW index, 4 rows per MFMA)
// and the whole recursion chains without any cross-lane movement except one LDS transpose per step
// for the symmetrisation.  V' and Quu are symmetric up to rounding, which is what lets V' Fz be
// computed as P(V', Fz).
//
// The association order differs from the reference's ((A'V)A vs A'(VA)) and the 7x7 system is solved
// by an unpivoted LDL' per lane instead of Eigen's pivoted LDLT + explicit inverse; both change K by
// O(1e-14) relative for the PD, lambda-regularised systems the algorithm accepts.  When a pivot is
// not positive (Q_uu + lambda I indefinite between two PD checks) the kernel falls back to a
// line-for-line port of Eigen's pivoted LDLT so that even that case follows the reference.
#include "mfma_common.h"

namespace kpilqr {

#define MFMA KP_MFMA

// acc + Y' X over NC row-chunks (rows 0 .. 4*NC-1 of Y and X).
template <int NC>
__device__ __forceinline__ d4 P(const d4 &Y, const d4 &X, d4 acc)
{
    acc = MFMA(Y.x, X.x, acc);
    if (NC > 1) acc = MFMA(Y.y, X.y, acc);
    if (NC > 2) acc = MFMA(Y.z, X.z, acc);
    if (NC > 3) acc = MFMA(Y.w, X.w, acc);
    return acc;
}

// Step-record loads go through a buffer descriptor that covers exactly ONE record: lanes whose tile
// element is a structural zero carry an out-of-range offset and the hardware returns 0 for them
// without touching memory.  No select on the loaded value means nothing consumes it until the MFMA
// that needs it, so the loads of step t-1 stay in flight behind the whole of step t.
#define OOB 0x7ffffff0

__device__ __forceinline__ double bld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}

// Pull one whole step record into this XCD's L2 a few steps before its tile loads are issued: one
// LDS-DMA dword per 64-byte sector (the data lands in a scratch LDS row nobody reads).  It costs no
// VGPR, and it takes the page-table walk and the HBM fetch of that record off the critical path --
// with records freshly written by the elementwise stages the translations are cold and a walk per
// step was worth ~3 ms per launch (B=1024, T=3000).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
// Issued through inline asm on purpose: hipcc treats a builtin LDS-DMA as a pending LDS write and
// drains vmcnt(0) before the step's next LDS access, which would expose the latency of the tile loads
// just issued.  The scratch row is never read, so no wait is needed; loads hidden from the compiler
// only make its counted vmcnt waits more conservative (the counter is in order).
__device__ __forceinline__ void glds_dword(const void *gaddr, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gaddr), "s"(lds_byte_addr));
}
__device__ __forceinline__ void l2_prefetch_record(const double *R, int rec_bytes, int lane, double *lds_scratch)
{
    const unsigned lds_addr = (unsigned)(size_t)(lds_void_t *)lds_scratch;
    const int o0 = lane * 64;
    const int o1 = (64 + lane) * 64;
    glds_dword((const char *)R + (o0 < rec_bytes ? o0 : rec_bytes - 8), lds_addr);
    if (o1 < rec_bytes) glds_dword((const char *)R + o1, lds_addr);
}
#define PF_DIST 0
#ifndef KP_RIC_SETS
#define KP_RIC_SETS 2              // tile sets of the backward sweep (requests run this many steps ahead)
#endif

struct TileOffs { int fz[4], fu[4], lzz[4], luz[4], luu[4]; };

struct StepTiles { d4 Fz, Fu, Lzz, Luz, Luu; };

template <int NCU>
__device__ __forceinline__ void load_step(const double *R, int rec_bytes, const TileOffs &o, StepTiles &s)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)R, 0, rec_bytes, 0x00020000);
    s.Fz.x = bld(r, o.fz[0]); s.Fz.y = bld(r, o.fz[1]); s.Fz.z = bld(r, o.fz[2]); s.Fz.w = bld(r, o.fz[3]);
    s.Fu.x = bld(r, o.fu[0]); s.Fu.y = bld(r, o.fu[1]); s.Fu.z = bld(r, o.fu[2]); s.Fu.w = bld(r, o.fu[3]);
    s.Lzz.x = bld(r, o.lzz[0]); s.Lzz.y = bld(r, o.lzz[1]); s.Lzz.z = bld(r, o.lzz[2]); s.Lzz.w = bld(r, o.lzz[3]);
    s.Luz.x = bld(r, o.luz[0]); s.Luu.x = bld(r, o.luu[0]);
    s.Luz.y = NCU > 1 ? bld(r, o.luz[1]) : 0.0; s.Luu.y = NCU > 1 ? bld(r, o.luu[1]) : 0.0;
    s.Luz.z = NCU > 2 ? bld(r, o.luz[2]) : 0.0; s.Luu.z = NCU > 2 ? bld(r, o.luu[2]) : 0.0;
    s.Luz.w = NCU > 3 ? bld(r, o.luz[3]) : 0.0; s.Luu.w = NCU > 3 ? bld(r, o.luu[3]) : 0.0;
}

// LDS map (doubles).  MS: row stride of the Quu image; MZ: column stride of the Quz image (odd ->
// conflict-free column reads); VS: row stride of the transpose image.
#define MS 16
#define MZ 17
#define VS 17
#define LDS_Q 0
#define LDS_Z (LDS_Q + 16 * MS)
#define LDS_V (LDS_Z + 16 * MZ)
#define LDS_SLOW (LDS_V + 16 * VS)
#define LDS_PF (LDS_SLOW + 2 * 256 + 16 + 16)
#define LDS_TOTAL (LDS_PF + 32)

// ABL: ablation switches for tools/ablate_backward.cpp only (0 in the product): 1 = always load the
// same record (no HBM streaming), 4 = skip the K/k stores, 8 = skip the symmetrisation transpose.
template <int R> __device__ __forceinline__ void set_reg(d4 &v, double x) { if (R == 0) v.x = x; else if (R == 1) v.y = x; else if (R == 2) v.z = x; else v.w = x; }

template <int N, int M, int ABL>
__device__ __forceinline__ void backward_body(RecLayout L, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                double *__restrict__ delta_J, int *__restrict__ status)
{
    constexpr int NCZ = (N + 1 + 3) / 4;      // row chunks covering z = [dx; 1]
    constexpr int NCU = (M + 3) / 4;          // row chunks covering u
    constexpr int n = N, m = M;
    __shared__ __attribute__((aligned(16))) double sh[LDS_TOTAL];
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    const double lam = lambda[b];

    TileOffs o;
    double one[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        o.fz[r] = (row < n && c < n) ? 8 * L.a(row, c) : OOB;
        one[r] = (row == n && c == n) ? 1.0 : 0.0;
        o.fu[r] = (row < n && c < m) ? 8 * L.b(row, c) : OOB;
        o.lzz[r] = (row < n && c < n) ? 8 * (L.off_lxx + row * n + c)
                 : (c == n && row < n) ? 8 * (L.off_lx + row)
                 : (row == n && c < n) ? 8 * (L.off_lx + c) : OOB;
        o.luz[r] = (row < m && c == n) ? 8 * (L.off_lu + row) : OOB;
        o.luu[r] = (row < m && c < m) ? 8 * (L.off_luu + row * m + c) : OOB;
    }
    const int rec_bytes = L.rec * 8;
    int oKst[4], okst[4];      // byte offsets of this lane's K / k elements, OOB where it owns none
    double lam2d[4];           // 2*lambda on the diagonal of the u-block
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        oKst[r] = (row < m && c < n) ? 8 * (row + c * m) : OOB;
        okst[r] = (row < m && c == n) ? 8 * row : OOB;
        lam2d[r] = (row == c && row < m) ? 2.0 * lam : 0.0;
    }
    // element (n,n) of a tile lives in lane (c = n, q = n & 3), register n >> 2
    const bool lane_nn = (c == n) && (q == (n & 3));
    constexpr int REG_NN = n >> 2;

    const double *R0 = rec + (size_t)b * T * L.stride;
    // Two tile sets (step t works on set (T-1-t) & 1): each is re-requested for step t-2 right behind its use in step t, so a
    // request has two steps to land (round 3; one set, copied at the top of the step, waited a trip to HBM per step AND sat out
    // the acknowledgement of the gains stored at the end of the step before -- a wave's loads and stores return in order, fused_mfma.hip).
    // The time loop is a body of two steps without conditions around memory operations.
    constexpr int NS = KP_RIC_SETS;
    StepTiles S[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) load_step<NCU>(R0 + (size_t)(T - 1 - k > 0 ? T - 1 - k : 0) * L.stride, rec_bytes, o, S[k]);
    d4 V = S[0].Lzz;                        // V_x = l_x[T-1]; V_xx = l_xx[T-1]   (iLQR.cpp:537-539)

    int pd_counter = 0;
    double dJ = 0.0;
    int fail = 0;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 Xinv = zero, Xprev = zero, Iu;            // running inverse of Quu + lambda I, the one before it, the identity of the u-block
    bool haveX = false;
    Iu.x = (q == c && c < m) ? 1.0 : 0.0; Iu.y = (4 + q == c && c < m) ? 1.0 : 0.0;
    Iu.z = (8 + q == c && c < m) ? 1.0 : 0.0; Iu.w = (12 + q == c && c < m) ? 1.0 : 0.0;

    // the gains of a step leave one step LATE, in front of the next requests (Kst / tst: the pending step)
    d4 Kst = zero;
    int tst = -1;
    auto store_gains = [&](int t_, const d4 &Kp) {
        const int t = __builtin_amdgcn_readfirstlane(t_);
        if (!(ABL & 4)) {
            // K (m x n column-major) and k out through bounds-checked buffer stores (lanes that own no
            // element carry an out-of-range offset and are dropped): straight-line code.
            __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void *)(Kout + ((size_t)b * T + t) * m * n), 0, m * n * 8, 0x00020000);
            __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)(kout + ((size_t)b * T + t) * m), 0, m * 8, 0x00020000);
            const double kv[4] = {Kp.x, Kp.y, Kp.z, Kp.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kv[r]), rK, oKst[r], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, kv[r]), rk, okst[r], 0, 0);
            }
        }
    };
    auto step = [&](int t, StepTiles &cur) {
        pd_counter++;
        const bool check_pd = pd_counter >= pd_stride;
        d4 Fz = cur.Fz;
        Fz.x += one[0]; Fz.y += one[1]; Fz.z += one[2]; Fz.w += one[3];

        // ---- Tu = V' Fu ; Quu = l_uu + Fu' Tu --------------------------------------- :577
        d4 Tu = P<NCZ>(V, cur.Fu, zero);
        d4 Quu = P<NCZ>(cur.Fu, Tu, cur.Luu);
        // ---- Tz, Quz, Qzz --------------------------------------------------------------- :570-579
        d4 Tz = P<NCZ>(V, Fz, zero);
        d4 Quz = P<NCZ>(cur.Fu, Tz, cur.Luz);
        d4 Qzz = P<NCZ>(Fz, Tz, cur.Lzz);
        // the tiles are used up: the gains of the step above go out, then this set is requested for step t-2
        // (behind step 1 the request repeats step 0's record: never used, but no condition around the loads)
        __builtin_amdgcn_sched_barrier(0);
        if (tst >= 0) store_gains(tst, Kst);
        // (readfirstlane: the step index is wave-uniform, but the compiler lost track of that in the first step of the pair and
        // wrapped each load in a readfirstlane loop)
        const int tn = __builtin_amdgcn_readfirstlane((ABL & 1) ? T - 1 : (t >= NS ? t - NS : 0));
        load_step<NCU>(R0 + (size_t)tn * L.stride, rec_bytes, o, cur);
        __builtin_amdgcn_sched_barrier(0);
        d4 Qr = Quu;                                  // Quu + lambda I
        Qr.x += 0.5 * lam2d[0]; Qr.y += 0.5 * lam2d[1]; Qr.z += 0.5 * lam2d[2]; Qr.w += 0.5 * lam2d[3];

        // ---- X = (Quu + lambda I)^-1 Quz.  Fast path: the inverse changes little from one step to the next, so
        //      it is refreshed by Newton-Schulz steps Xinv <- Xinv + Xinv (I - Q Xinv) on the matrix core (4 MFMAs
        //      each, quadratic convergence; the count is chosen from the measured residual so that it ends below
        //      1e-15) and X = Xinv Quz is one more product -- the reference, too, forms the explicit inverse and
        //      multiplies (iLQR.cpp:597-604).  The LDL' path below runs on the first step, on every checked step
        //      (it gives the PD verdict of :587-595), and whenever the residual is too large to converge fast.
        d4 Xp = zero;
        bool done = false;
        if (haveX && !check_pd && kp_inverse_refresh_p<NCU>(Qr, Iu, Xinv, Xprev, m)) {
            Xp = P<NCU>(Xinv, Quz, zero);                                  // Xinv' Quz = (Quu + lambda I)^-1 Quz
            done = true;
        }
        if (!done) {
            // Quu + lambda I -> LDS image (row-major, stride MS) and Quz (column-major, stride MZ) for the per-lane solve
            sh[LDS_Q + q * MS + c] = Qr.x;
            if (NCU > 1) sh[LDS_Q + (4 + q) * MS + c] = Qr.y;
            if (NCU > 2) sh[LDS_Q + (8 + q) * MS + c] = Qr.z;
            if (NCU > 3) sh[LDS_Q + (12 + q) * MS + c] = Qr.w;
            sh[LDS_Z + c * MZ + q] = Quz.x;
            if (NCU > 1) sh[LDS_Z + c * MZ + 4 + q] = Quz.y;
            if (NCU > 2) sh[LDS_Z + c * MZ + 8 + q] = Quz.z;
            if (NCU > 3) sh[LDS_Z + c * MZ + 12 + q] = Quz.w;
            __syncthreads();
            // ---- unpivoted LDL' of Quu + lambda I, redundantly in every lane (lower triangle) ----
            double Lm[M][M], rd[M];
            // (every lane factorises the same image; the ballot tells the compiler that the verdict -- and with it `fail` and the
            // exit of the time loop -- is wave-uniform: with a per-lane verdict the buffer descriptors of the step became
            // 'divergent' and every load was wrapped in a readfirstlane loop)
            const bool pos = __builtin_amdgcn_ballot_w64(!kp_ldl_factor<M>([&](int i, int j) { return sh[LDS_Q + i * MS + j]; }, Lm, rd)) == 0;
            if (check_pd) {                       // CheckMatrixPD every pd_stride steps   :587-595
                if (!pos) { if (!fail) fail = t + 1; return; }
                pd_counter = 0;
            }
            // ---- every lane solves (Quu + lambda I) x = Quz[:, c] for its own column c --------------
            double x[M];
            if (pos) {
#pragma unroll
                for (int i = 0; i < M; i++) x[i] = sh[LDS_Z + c * MZ + i];
                kp_ldl_solve<M>(Lm, rd, x);
                // seed the fast path: column c of the inverse in lane c (c < m), as a tile
                double y[M];
#pragma unroll
                for (int i = 0; i < M; i++) y[i] = (i == c) ? 1.0 : 0.0;
                kp_ldl_solve<M>(Lm, rd, y);
                double yr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < M; i++)
                    if (q == (i & 3)) yr[i >> 2] = (c < m) ? y[i] : 0.0;
                Xinv.x = yr[0]; Xinv.y = yr[1]; Xinv.z = yr[2]; Xinv.w = yr[3];
                Xprev = Xinv;
                haveX = true;
            } else {
                // Q_uu + lambda I is not PD and this is not a checked step: follow Eigen's pivoted
                // LDLT + explicit inverse exactly (iLQR.cpp:597-604).
                double *wa = sh + LDS_SLOW, *wx = wa + 256, *wt = wx + 256;
                int *tr = (int *)(wt + 16);
                if (lane == 0) kp_slow_ldlt_inverse(L.m, sh + LDS_Q, MS, wa, wx, wt, tr);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < M; i++) {
                    double sacc = 0.0;
#pragma unroll
                    for (int p = 0; p < M; p++) sacc += (-wx[i + p * m]) * sh[LDS_Z + c * MZ + p];
                    x[i] = -sacc;
                }
                __syncthreads();
                haveX = false;
            }
            // X as a tile in D layout: register r of lane (c,q) = X[4r+q][c]
            double xr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < M; i++)
                if (q == (i & 3)) xr[i >> 2] = x[i];
            Xp.x = xr[0]; Xp.y = xr[1]; Xp.z = xr[2]; Xp.w = xr[3];
        }
        Kst = -Xp; tst = t;
        // ---- delta_J += k'Q_u + k'Q_uu k  (:612-613).  With (Q_uu + lambda I) k = -Q_u this is
        //      k'(Q_uu k + Q_u) = -lambda k'k, evaluated in that cancellation-free form: the lanes of column n keep
        //      the squares of their rows, the four row groups are added once after the sweep.
        if (c == n) {
            dJ -= lam * (Xp.x * Xp.x);
            if (NCU > 1) dJ -= lam * (Xp.y * Xp.y);
            if (NCU > 2) dJ -= lam * (Xp.z * Xp.z);
            if (NCU > 3) dJ -= lam * (Xp.w * Xp.w);
        }
        // ---- V' = Qzz + K''Quu K' + K''Quz + Quz'K'   (:606-607).  Substituting Quz = -(Quu + lambda I) K'
        //      gives V' = Qzz - K''(Quu + 2 lambda I) K' = Qzz + X'[(Quu + 2 lambda I) K']: two products.
        //      and since (Quu + lambda I) X = Quz that bracket is -(Quz + lambda X): ONE product.
        d4 G;
        G.x = -__builtin_fma(lam, Xp.x, Quz.x); G.y = -__builtin_fma(lam, Xp.y, Quz.y);
        G.z = -__builtin_fma(lam, Xp.z, Quz.z); G.w = -__builtin_fma(lam, Xp.w, Quz.w);
        d4 acc = P<NCU>(Xp, G, Qzz);

        // ---- V' = (V' + V'')/2 through an LDS transpose   (:610) -----------------------------------
        if (ABL & 8) { V = acc; return; }
        sh[LDS_V + (q) * VS + c] = acc.x;
        sh[LDS_V + (4 + q) * VS + c] = acc.y;
        sh[LDS_V + (8 + q) * VS + c] = acc.z;
        sh[LDS_V + (12 + q) * VS + c] = acc.w;
        __syncthreads();
        V.x = 0.5 * (acc.x + sh[LDS_V + c * VS + q]);
        V.y = 0.5 * (acc.y + sh[LDS_V + c * VS + 4 + q]);
        V.z = 0.5 * (acc.z + sh[LDS_V + c * VS + 8 + q]);
        V.w = 0.5 * (acc.w + sh[LDS_V + c * VS + 12 + q]);
        if (lane_nn) set_reg<REG_NN>(V, 0.0);      // element (n,n) carries nothing: keep it at zero
        __syncthreads();
    };
    // (one exit per pair, at its end: an exit between the two steps makes the compiler carry the tile sets through copies at
    // the back edge, and a copy of a requested register waits for the request.  A step that follows a failed one works on the
    // stale V' -- valid numbers, results nobody reads: the status says so -- and leaves `fail` alone.)
    int t = T - 1;
    for (; t >= NS - 1; t -= NS) {
#pragma unroll
        for (int k = 0; k < NS; k++) step(t - k, S[k]);
        if (fail) break;
    }
#pragma unroll
    for (int k = 0; k < NS - 1; k++)
        if (t - k >= 0 && !fail) step(t - k, S[k]);
    if (tst >= 0) store_gains(tst, Kst);           // the last completed step
    // delta_J: the four row groups of column n; status is uniform
    dJ += __shfl_xor(dJ, 16);
    dJ += __shfl_xor(dJ, 32);
    if (lane_nn) delta_J[b] = dJ;
    if (lane == 0) status[b] = fail;
}

// Two entry points over the same body.  One wavefront owns a SIMD's FP64 pipe for the whole horizon,
// so when the batch fits the chip (<= 1 wave per SIMD) the "exclusive" form declares at most one wave
// per SIMD: the hardware dispatcher can then never stack two trajectories on one SIMD while others
// sit idle (it did, after kernels of a different shape: 8.5 ms instead of 6.2 ms at B=1024, T=3000).
// Larger batches use the shared form (two co-resident waves per SIMD hide each other's latencies).
template <int N, int M, int ABL = 0>
__global__ void __launch_bounds__(64)
k_backward_mfma(RecLayout L, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                double *__restrict__ delta_J, int *__restrict__ status)
{
    backward_body<N, M, ABL>(L, T, rec, lambda, pd_stride, Kout, kout, delta_J, status);
}

template <int N, int M, int ABL = 0>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_backward_mfma_excl(RecLayout L, int T, const double *__restrict__ rec, const double *__restrict__ lambda,
                     int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                     double *__restrict__ delta_J, int *__restrict__ status)
{
    backward_body<N, M, ABL>(L, T, rec, lambda, pd_stride, Kout, kout, delta_J, status);
}

bool backward_mfma_supported(int n, int m)
{
    return kp_t1_shape(n, m);
}

hipError_t launch_backward_mfma(Ctx *c, int pd_stride)
{
    const int n = c->n, m = c->d.m;
    dim3 grid(c->d.batch), block(64);
    const bool excl = c->d.batch <= c->n_simd;
#define LAUNCH(NN, MM)                                                                                    \
    do {                                                                                                  \
        if (excl)                                                                                         \
            hipLaunchKernelGGL((k_backward_mfma_excl<NN, MM>), grid, block, 0, c->stream, c->L, c->d.T,   \
                               c->rec, c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status);          \
        else                                                                                              \
            hipLaunchKernelGGL((k_backward_mfma<NN, MM>), grid, block, 0, c->stream, c->L, c->d.T,        \
                               c->rec, c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status);          \
    } while (0)
#define KP_X(NN, MM) if (n == NN && m == MM) { LAUNCH(NN, MM); return hipGetLastError(); }
    KP_T1_SHAPES(KP_X)
#undef KP_X
#undef LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace kpilqr
