// fused_mfma.hip -- the "never materialise A, B, l_*" form of the backward (a7) and forward (a8) passes
// for n+2 <= 16: the interpolation of the dynamics Jacobians (a4) and the
// Gauss-Newton cost derivatives (a6) are evaluated INSIDE the two sweeps, from the differenced key-point columns in the
// key-point column store kpc [entry][3][n] (common.h: 3n doubles per (trajectory, DoF, key-point), no step records) and
// from the residuals / residual Jacobians.  The one-wave backward sweep also has a RAW form that differences the
// key-point ordered FD payload (x+ / x- in entry order, kpilqr_upload_fd_kp) itself at the segment crossings -- the
// arithmetic of Differentiator.cpp:166-222,441-457 -- and leaves kpc behind for the forward sweep: no differencing kernel,
// no intermediate pass over the payload.
//
//   * a4 (KeypointGenerator::InterpolateDerivatives, src/KeyPointGenerator/KeyPointGenerator.cpp:840-954):
//     in the MFMA "D" layout lane (c,q) holds rows 4r+q of COLUMN c of A (and of B), and the reference
//     interpolates column-wise with one key-point list per DoF -- so each lane walks the key-point list of
//     its own DoF and keeps (value at segment start, slope) in registers:
//         A_t[:,c] = A_s[:,c] + (t-s) * ((A_e[:,c] - A_s[:,c]) / (e-s))          (:898-905, :933-948)
//     the slope is the correctly rounded quotient k_interpolate forms; the value is ONE fused multiply-add of it
//     (KP_LERP_FMA, round 4: a single rounding where k_interpolate has two -- the sweeps are held to the oracle at 1e-9, the
//     materialised A, B of kpilqr_interpolate keep the reference's bits).  Key-point columns of the next segment are
//     fetched one segment ahead (time indices two ahead).
//   * a6 (ModelTranslator::CostDerivativesFromResiduals, src/ModelTranslator/ModelTranslator.cpp:552-583):
//     with Rz = [r_x | r] (nr x (n+1)) and W = diag(2w):  Lzz = [l_xx l_x; l_x' *] = Rz' W Rz is ONE
//     P(Rz, W Rz); l_uu and l_u come out of P(Ru, W [Ru | r]) (columns < m and column n).  The reference
//     has no l_ux term, so the cross product r_u' W r_x is never formed.  Terminal weights at t = T-1
//     (Optimiser::ComputeCostDerivatives, src/Optimiser/Optimiser.cpp:202-211).
//     The forward pass scores a candidate with the same model written on the residuals:
//         l_x'dx + dx'l_xx dx/2 = sum_k w_k Jx_k (2 r_k + Jx_k),   Jx = r_x dx      (likewise for du).
//
// Wave organisations (launch_backward_fused / launch_forward_fused pick by batch size):
//   backward: one wave per trajectory (batch > #SIMDs/2) | producer / consumer pair | consumer / side / producer triple
//             (batch <= #CUs) | control / state split (kept as a tested alternative);
//   forward:  one wave -- in a form for uniform key-point sets (no LDS transpose) and a general one, chosen on the device --
//             | state / cost pair | state / cost / staging triple.
// Everything else (homogeneous form, LDL' solve, slow path, stores) is riccati_mfma.hip / forward_mfma.hip.
// Per launch the kernels read Kp*(n^2+nm) + T*nr*(1+n+m) doubles instead of T*(2n^2+nm+n+m^2+m).
#include <cstdlib>
#include <type_traits>
#include "mfma_common.h"

namespace kpilqr {

typedef unsigned int u32x2f __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define OOBF 0x7ffffff0
#ifndef KP_NS
#define KP_NS 1
#endif
#define BIGT 0x3fffffff
// An offset no trajectory's slice of kpc reaches (fused_supported keeps a slice below 1 GB): a lane / register / entry with
// nothing to load adds it instead of selecting -- BIGOFF and BIGOFF + BIGOFF are both out of range of the descriptor and do
// not wrap, so the load returns 0 (the store is dropped) without a compare-and-select per access.
#define BIGOFF 0x40000000

// Latency probe (-DKP_PROBE_SAMEB, tools/build_variant.sh): every workgroup of the one-wave sweeps works on one of eight
// trajectories, so all of its loads hit in cache -- the sweep time that is left is what the memory system does NOT account for
// (results are only meaningful for a batch that repeats 8 problems, synth.tile_problem).
#ifndef KP_FWD_SETS
#define KP_FWD_SETS 4              // register sets of the one-wave forward sweep (requests run this many steps ahead)
#endif
#ifndef KP_FWD_SETS_GEN
#define KP_FWD_SETS_GEN 6          // ... of its general (per-DoF list) form (late round 4, same box: 3 / 4 / 5 / 6 / 8 sets = 2.37 / 2.32 / 2.30 / 2.29 / 2.46 ms
                                   // on velocity_change lists, 2.73 / 2.70 / 2.65 / 2.64 / 2.81 on adaptive_jerk lists)
#endif
#ifndef KP_BWD_LATE_STORE
#define KP_BWD_LATE_STORE 1
#endif
#ifndef KP_FSC_SETS
#define KP_FSC_SETS 4              // tile sets of the forward state / cost wave groups
#endif
#ifndef KP_PROBE_BWD
#define KP_PROBE_BWD 0
#endif
// The launchers (and with them the kernel instantiations) of this file compile as THREE translation units (Makefile: -DKP_FUSED_PART=1|2|3;
// unset or 0: everything in one): 1 one wave per trajectory backward + the form choices, 2 the backward wave pairs / triple,
// 3 the forward sweeps -- the file takes two minutes as one unit.
#ifndef KP_FUSED_PART
#define KP_FUSED_PART 0
#endif
#define KP_PART(n) (KP_FUSED_PART == 0 || KP_FUSED_PART == (n))
#ifndef KP_KINK4
#define KP_KINK4 1                  // 0: the step below a key-point refreshes the running inverse like every other step (round 3; A/B builds)
#endif
#ifndef KP_FWD_TRIM
#define KP_FWD_TRIM 1               // 0: the one-wave forward sweep starts its control-law product from u_nom and clamps with compare-and-select (A/B builds)
#endif
#ifndef KP_FWD_SQW
#define KP_FWD_SQW 1                // 0: the headline's forward sweep scores on the unscaled r_x dx (round 4; A/B builds)
#endif
#ifndef KP_BWD_RV2
#define KP_BWD_RV2 1                // 0: the headline backward sweep fetches r_t with four 8-byte requests: A/B builds
#endif
#ifndef KP_FWD_RV2
#define KP_FWD_RV2 1                // 0: the headline forward sweep fetches r_t with four 8-byte requests (rows in their natural order): A/B builds
#endif
#ifndef KP_RXC_CXX
#define KP_RXC_CXX 1                // 0: the RXC sweeps form Lzz = Rz' W Rz with four products at every step (round 4; A/B builds)
#endif
#ifndef KP_SLOPES
#define KP_SLOPES 1                 // 0: the general forms divide at their crossings (rounds 1-3) instead of reading the slope store (A/B builds)
#endif
#ifdef KP_PROBE_SAMEB
#define KP_BLOCK_TRAJ ((int)(blockIdx.x & 7))
#else
#define KP_BLOCK_TRAJ ((int)blockIdx.x)
#endif
template <int NC>
__device__ __forceinline__ d4 PS(const d4 &Y, const d4 &X, d4 acc)
{
    acc = MFMA(Y.x, X.x, acc);
    if (NC > 1) acc = MFMA(Y.y, X.y, acc);
    if (NC > 2) acc = MFMA(Y.z, X.z, acc);
    if (NC > 3) acc = MFMA(Y.w, X.w, acc);
    return acc;
}
__device__ __forceinline__ d4 PR(const d4 &Y, const d4 &X, d4 acc, int nc)      // nc wave-uniform
{
    acc = MFMA(Y.x, X.x, acc);
    if (nc > 1) acc = MFMA(Y.y, X.y, acc);
    if (nc > 2) acc = MFMA(Y.z, X.z, acc);
    if (nc > 3) acc = MFMA(Y.w, X.w, acc);
    return acc;
}
__device__ __forceinline__ double fbld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
// the same with a wave-uniform byte offset in the instruction's scalar operand: ONE descriptor per array and trajectory,
// the step selected by an SGPR, instead of a descriptor rebuilt (64-bit address arithmetic, ~8 SALU) per array and step
__device__ __forceinline__ double fblds(__amdgpu_buffer_rsrc_t r, int byte_off, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, soff, 0));
}
// two consecutive doubles with one request (16-byte aligned)
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fbld2s(__amdgpu_buffer_rsrc_t r, int byte_off, int soff, double &a, double &b)
{
    const u32x4f v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, soff, 0);
    const u32x2f lo = {v.x, v.y}, hi = {v.z, v.w};
    a = __builtin_bit_cast(double, lo); b = __builtin_bit_cast(double, hi);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frsrc(const void *p, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, bytes, 0x00020000);
}
// correctly rounded a/den for a normal-range quotient: one residual correction on top of a ~1 ulp reciprocal
// (the hardware division expansion without its scaling / fix-up tail; den is a small positive integer)
__device__ __forceinline__ double fdiv(double a, double den, double rinv)
{
    const double q0 = a * rinv;
    const double rr = __builtin_fma(-den, q0, a);
    return __builtin_fma(rr, rinv, q0);
}
// v_min_f64 / v_max_f64 as they are: the builtins come with a canonicalising v_max x, x, x of every operand under the IEEE mode
// of compute kernels -- per use, not hoisted -- which is what the clamp of the control law (iLQR.cpp:883-889) does NOT need: its
// limits are finite constants (a NaN candidate stays visible in the state and the predicted cost either way)
__device__ __forceinline__ double vmin64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double bits_or(double a, double b)
{
    return __builtin_bit_cast(double, __builtin_bit_cast(u64, a) | __builtin_bit_cast(u64, b));
}
__device__ __forceinline__ double bits_and(double a, u64 mask)
{
    return __builtin_bit_cast(double, __builtin_bit_cast(u64, a) & mask);
}
// start + dt*slope.  KP_LERP_FMA = 0: two instructions, never contracted -- k_interpolate's operation order, the bits the
// materialising pipeline writes (KeyPointGenerator.cpp:933-948).  1: ONE fused multiply-add (a single rounding: at least as close
// to the exact interpolant) -- 16 VALU instructions less per forward step and 8 per backward step on a wave that pays ~9 cycles
// for each; the sweeps' results move by ~1e-16 relative (they are held to the oracle at 1e-9, not bit for bit: the MFMA
// accumulation order differs from the reference's loops anyway); kpilqr_interpolate / get_AB still give the reference's bits.
#ifndef KP_LERP_FMA
#define KP_LERP_FMA 1
#endif
__device__ __forceinline__ double lerp_nc(double sv, double dt, double av)
{
#if KP_LERP_FMA
    return __builtin_fma(dt, av, sv);
#else
#pragma clang fp contract(off)
    const double p = dt * av;
    return sv + p;
#endif
}

struct FusedArgs {
    const int *kp_offsets, *kp_times;              // per (b, dof) CSR of key-point time indices
    const double *r, *r_x, *r_u, *w_run, *w_term;  // [b][T+1][nr], [..][nr][n], [..][nr][m], [nr], [nr]
    int dof, nr;
    double *kpc;                                   // key-point column store [entry][3][n] (read; written by the raw backward sweep)
    const char *fdk;                               // key-point ordered FD payload (raw backward sweep only): one record per entry,
                                                   // [(x+, x-) pairs of the 3n elements | int32 mode: bit k = kind k is one-sided | pad] = (6n + 2) * 8 bytes
    double eps2, rinv_2eps;                        // 2 eps and the host's correctly rounded 1 / (2 eps): with it fdiv IS the IEEE
                                                   // quotient (eps and 1/eps are exact halves / doubles of them)
    const double *rx_const;                        // RXC: r_x [nr][n], the same for every trajectory and step (kpilqr_upload_residual_jacobians_const)
    const double *kps;                             // slope store [entry][3][n][2] beside kpc: every key-point column with its slope to the
                                                   // next key-point of its DoF list (k_kp_slopes); read by the general (per-DoF list) forms
};

// ---- column tracker: lane (c,q) interpolates rows 4r+q of column c of A and of B ------------------------
// col[0..3] = A(4r+q, c), col[4..7] = B(4r+q, c) (c < m).  The whole trajectory's key-point columns -- entries
// [E0, E0 + NE) of kpc, 3n doubles each: position column | velocity column | control column of the entry's DoF -- sit
// behind one buffer descriptor; tk is the entry RELATIVE to E0 (anything outside [0, NE) reads zeros), and a lane with
// nothing to load carries an out-of-range offset and reads 0.
struct ColOffs { int a[4], b[4]; };

// offsets of lane (c, q) inside an entry: rows 4r+q of A column c (the position column of DoF c, or the velocity column of
// DoF c - dof) and of B column c
__device__ __forceinline__ void col_offsets(ColOffs &co, int n, int m, int dof, int c, int q)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        co.a[r] = (row < n && c < n) ? 8 * ((c < dof ? 0 : n) + row) : BIGOFF;
        co.b[r] = (row < n && c < m) ? 8 * (2 * n + row) : BIGOFF;
    }
}

__device__ __forceinline__ void load_col(__amdgpu_buffer_rsrc_t rT, const ColOffs &o, int tk, int T, int strideB, double *col)
{
    const int base = ((unsigned)tk < (unsigned)T) ? tk * strideB : BIGOFF;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        col[r] = fbld(rT, base + o.a[r]);
        col[4 + r] = fbld(rT, base + o.b[r]);
    }
}

// (value, slope) pairs of the slope store kps [entry][3][n][2] (k_kp_slopes): ONE 16-byte load per element gives a segment's start
// value and its slope to the next key-point.  Offsets are those of kpc doubled.
typedef unsigned int u32x4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fbld2(__amdgpu_buffer_rsrc_t r, int byte_off, double &v, double &a)
{
    const u32x4f x = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    v = __builtin_bit_cast(double, u32x2f{x.x, x.y});
    a = __builtin_bit_cast(double, u32x2f{x.z, x.w});
}
__device__ __forceinline__ int dbl_off(int o) { return o == BIGOFF ? BIGOFF : 2 * o; }
__device__ __forceinline__ void load_col2(__amdgpu_buffer_rsrc_t rS, const ColOffs &o, int tk, int T, int strideB2, double *col, double *slp)
{
    const int base = ((unsigned)tk < (unsigned)T) ? tk * strideB2 : BIGOFF;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        fbld2(rS, base + dbl_off(o.a[r]), col[r], slp[r]);
        fbld2(rS, base + dbl_off(o.b[r]), col[4 + r], slp[4 + r]);
    }
}

// The same for a lane whose four registers are rows q, 4+q, 8+q, 12+q of ONE column (the column layout of the backward
// sweep): one offset per column and lane, the rows in the loads' immediate field.  Registers 4r+3 < N are valid in every
// lane that holds the column at all; the one partial register (N % 4 rows) has a per-lane offset of its own; registers
// beyond the column are zero without a load.
struct ColOffsN { int a, al, b, bl; };
template <int N>
__device__ __forceinline__ void col_offsets_n(ColOffsN &co, int m, int dof, int c, int q)
{
    constexpr int NF = N / 4;
    co.a = (c < N) ? 8 * ((c < dof ? 0 : N) + q) : BIGOFF;
    co.b = (c < m) ? 8 * (2 * N + q) : BIGOFF;
    co.al = (c < N && 4 * NF + q < N) ? 8 * ((c < dof ? 0 : N) + 4 * NF + q) : BIGOFF;
    co.bl = (c < m && 4 * NF + q < N) ? 8 * (2 * N + 4 * NF + q) : BIGOFF;
}
template <int N>
__device__ __forceinline__ void load_col_n(__amdgpu_buffer_rsrc_t rT, const ColOffsN &o, int base, double *col)
{
    constexpr int NF = N / 4;
    const int va = base + o.a, vb = base + o.b;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        col[r] = r < NF ? fbld(rT, va + 32 * r) : (r == NF && (N & 3)) ? fbld(rT, base + o.al) : 0.0;
        col[4 + r] = r < NF ? fbld(rT, vb + 32 * r) : (r == NF && (N & 3)) ? fbld(rT, base + o.bl) : 0.0;
    }
}
// (value, slope) pairs from the slope store, column layout (o2: col_offsets_n's offsets doubled)
template <int N>
__device__ __forceinline__ void load_col2_n(__amdgpu_buffer_rsrc_t rS, const ColOffsN &o2, int base2, double *col, double *slp)
{
    constexpr int NF = N / 4;
    const int va = base2 + o2.a, vb = base2 + o2.b;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r < NF) { fbld2(rS, va + 64 * r, col[r], slp[r]); fbld2(rS, vb + 64 * r, col[4 + r], slp[4 + r]); }
        else if (r == NF && (N & 3)) { fbld2(rS, base2 + o2.al, col[r], slp[r]); fbld2(rS, base2 + o2.bl, col[4 + r], slp[4 + r]); }
        else { col[r] = 0.0; slp[r] = 0.0; col[4 + r] = 0.0; slp[4 + r] = 0.0; }
    }
}
template <int N>
__device__ __forceinline__ void store_col_n(__amdgpu_buffer_rsrc_t rT, const ColOffsN &o, int base, const double *col)
{
    constexpr int NF = N / 4;
    const int va = base + o.a, vb = base + o.b;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (r < NF) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, col[r]), rT, va + 32 * r, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, col[4 + r]), rT, vb + 32 * r, 0, 0);
        } else if (r == NF && (N & 3)) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, col[r]), rT, base + o.al, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, col[4 + r]), rT, base + o.bl, 0, 0);
        }
    }
}

// Every kernel of this file takes (RecLayout L, FusedArgs F, ...) first, so F sits at a fixed place of the kernel-argument
// segment.  Scalars that only a segment CROSSING needs (descriptors of the key-point stores, the key-point times, eps) are
// re-read from there inside the crossing -- scalar loads, every few steps -- instead of staying live across the hot loop,
// where the register allocator would park them in VGPR lanes and pay a v_readlane / v_writelane (a VALU slot each) per use.
static_assert(sizeof(RecLayout) == 40 && alignof(FusedArgs) == 8, "kernel-argument layout assumed by kernarg_fused()");
typedef const __attribute__((address_space(4))) FusedArgs *KArgF;
__device__ __forceinline__ KArgF kernarg_fused()
{
    const __attribute__((address_space(4))) char *ka = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));                 // opaque: the loads below cannot be hoisted out of the loop that holds the call
    return (KArgF)(ka + 40);
}

// LDS map of the backward kernel (doubles) -- as riccati_mfma.hip
#define FMS 16
#define FMZ 17
#define FVS 17
#define FLDS_Q 0
#define FLDS_Z (FLDS_Q + 16 * FMS)
#define FLDS_V (FLDS_Z + 16 * FMZ)
#define FLDS_SLOW (FLDS_V + 16 * FVS)
#define FLDS_TOTAL (FLDS_SLOW + 2 * 256 + 16 + 16)

template <int R> __device__ __forceinline__ void fset_reg(d4 &v, double x) { if (R == 0) v.x = x; else if (R == 1) v.y = x; else if (R == 2) v.z = x; else v.w = x; }

// PC = false: one wavefront per trajectory does everything.  PC = true: this is the CONSUMER wave of a two-wave block; the
// step's tiles Fz, Fu, Lzz, [l_uu | l_u] come from the LDS ring filled by the producer wave (fusedpc_producer below), LDS
// hand-offs inside the step are wave-local, and the only block barrier is the one that ends a step.
#define FPC_TILES 4
#define FPC_BUF (FPC_TILES * 256)
// behind the two ring slots (offsets from the ring base): the side wave's Quz and Qzz of the wave triple
#define FPC_SIDE_QUZ (2 * FPC_BUF)
#define FPC_SIDE_QZZ (FPC_SIDE_QUZ + 256)
__device__ __forceinline__ d4 lds_tile4(const double *t, int lane);
__device__ __forceinline__ void lds_store4(double *t, int lane, const d4 &v);
// RU0: the context's r_u buffer was never written (a task without control residuals: r_u = 0, e.g. reaching,
// src/ModelTranslator/Reaching.cpp:43-54), so [l_uu | l_u] = Ru' W [Ru | r] is exactly zero: the product and the r_u loads
// are left out (4 of the step's 40 MFMAs).
// RAW (one wave per trajectory only): the sweep reads the key-point ordered FD payload (F.xp, F.xm, F.mode) instead of kpc,
// differences every column when it becomes a segment start -- (x+ - x-) / (2 eps), or / eps for a one-sided job, with
// the host's correctly rounded reciprocals, i.e. the bytes k_fd_kp_difference would write -- and stores it to kpc, which
// the forward sweep then reads.
// UNI (one wave per trajectory only): every DoF of the trajectory has the SAME key-point list (set_interval -- the reference's
// default -- and whatever else comes out uniform; the device flag of k_kp_uniform says so).  Segment start, end and position
// are then wave-uniform scalars, the sweep is a loop over SEGMENTS with the crossing in straight-line code at its top and the
// first step of the segment peeled behind it, and the loads of a crossing have a whole Riccati chain to land before anything
// waits for them.  (In the general form the crossing sits in a per-lane branch, behind which the compiler has to wait for
// every memory operation in flight at the next load-dependent instruction: ~1 us of HBM latency per crossing, 7.5 against
// 4.5 ms per sweep with a key-point at every step.)
// STATS (diagnostic instantiation, kpilqr_backward_stats): counts per trajectory what every step did to obtain
// (Quu + lambda I)^-1 -- hist[0] third-order refresh only, [1..3] plus that many second-order steps, [4] LDL' factorisation
// (first step, checked steps, re-seeds), [5] the pivoted slow path -- into hist [batch][6].
// RXC (one wave per trajectory, with RU0): the residual Jacobian r_x is the same [nr][n] matrix at every step of every
// trajectory (reaching: selector rows, src/ModelTranslator/Reaching.cpp:43-54; any task whose residuals are affine in the state):
// the Rx tile is loaded ONCE and stays in registers, the sweep issues no r_x loads (T nr n doubles per trajectory: 5.0 of the
// 8.1 MB a trajectory's backward sweep reads at the headline shape).  Same products in the same order: bit-identical gains.
template <int N, int M, bool PC, bool RU0 = false, bool SIDE = false, bool RAW = false, bool UNI = false, bool STATS = false, bool RXC = false>
__device__ __forceinline__ void backward_fused_body(double *sh, const double *pcbuf, int *sflag, RecLayout L, FusedArgs F, int T,
                const double *__restrict__ lambda, int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                double *__restrict__ delta_J, int *__restrict__ status, int *__restrict__ hist = nullptr)
{
    int hcnt[6] = {0, 0, 0, 0, 0, 0};
    (void)hcnt;
    constexpr int NCZ = (N + 1 + 3) / 4;
    constexpr int NCU = (M + 3) / 4;
    constexpr int n = N, m = M;
    // one wavefront per workgroup (or the consumer wave of a group): its LDS operations execute in order, so a write ->
    // transposed read pair needs the compiler's ordering only (4.94 -> 4.87 ms against __syncthreads() in the one-wave forms).
    // (Tried: the transposed half requested at the end of a step and averaged in front of the next step's first Riccati
    // product, so that the LDS round trip runs under that step's a4 / a6 -- 7.73 ms: eight more live registers across the
    // top of the step.)
    auto wsync = [&]() { __builtin_amdgcn_wave_barrier(); };
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int b = KP_BLOCK_TRAJ;
    const double lam = lambda[b];
    // (probe builds, -DKP_PROBE_BWD=bits: 1 residual loads, 2 key-point stores, 4 gain stores go to one of eight trajectories)
    const int bR = (KP_PROBE_BWD & 1) ? (int)(blockIdx.x & 7) : b, bP = (KP_PROBE_BWD & 2) ? (int)(blockIdx.x & 7) : b, bS = (KP_PROBE_BWD & 4) ? (int)(blockIdx.x & 7) : b;
    constexpr bool RV2B = KP_BWD_RV2 && KP_RXC_CXX && RXC && RU0 && !PC;     // (see below)
    // 4-row chunks of the residual index that hold residuals (RV2B: registers 0, 1 hold residuals 0 .. 7, registers 2, 3 residuals 8 .. 15)
    const int nr = F.nr, ncr = RV2B ? (nr > 9 ? 4 : nr > 8 ? 3 : nr > 1 ? 2 : 1) : (nr + 3) >> 2;
    constexpr int strideB = 3 * N * 8;                            // bytes of one key-point entry: three columns

    ColOffsN co;
    col_offsets_n<N>(co, m, F.dof, c, q);
    double w2run[4], w2term[4], lam2d[4];
    int oRx[4], oR1[4], oRu[4], oKst[4], okst[4];
    // RV2B (the headline's one-wave sweep: constant r_x, no control residuals): the residuals sit in the rows of the Rz tiles in the order
    //     row 4r + q  <->  residual 8 (r >> 1) + 2 q + (r & 1)
    // -- registers (0, 1) and (2, 3) of a lane are consecutive residuals, so r_t arrives with two 16-byte requests instead of four
    // 8-byte ones (see RV2 in forward_fused_body).  The rows are the contraction index of every product they enter (Cxx, r_x' W r, the
    // terminal step's Rz' W Rz): a relabelling, with the weights relabelled alike.
    const int oR12[2] = {(2 * q < nr) ? 16 * q : OOBF, (8 + 2 * q < nr) ? 64 + 16 * q : OOBF};
    (void)oR12;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int kr = RV2B ? 8 * (r >> 1) + 2 * q + (r & 1) : 4 * r + q;     // the residual in row 4r + q
        oRx[r] = (kr < nr && c < n) ? 8 * (kr * n + c) : OOBF;        // Rz(k, c) = r_x[k][c]
        w2run[r] = (kr < nr) ? 2.0 * F.w_run[kr] : 0.0;
        w2term[r] = (kr < nr) ? 2.0 * F.w_term[kr] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        // (RXC one-wave sweeps: r[k] in EVERY column -- the lanes of column c form their part of (r_x' W r)(c) from it, see CXX below)
        oR1[r] = (row < nr && (c == n || (KP_RXC_CXX && RXC && !PC))) ? 8 * row : OOBF;               //            ... | r[k] in column n
        oRu[r] = (row < nr && c < m) ? 8 * (row * m + c) : OOBF;      // Ru(k=row, c) = r_u[k][c]
        oKst[r] = (row < m && c < n) ? 8 * (row + c * m) : OOBF;
        okst[r] = (row < m && c == n) ? 8 * row : OOBF;
        lam2d[r] = (row == c && row < m) ? 2.0 * lam : 0.0;
    }
    const d4 LamI = {0.5 * lam2d[0], 0.5 * lam2d[1], 0.5 * lam2d[2], 0.5 * lam2d[3]};     // lambda on the diagonal of the u-block
    const u64 mask_n = (c == n) ? ~0ull : 0ull, mask_u = (c < m) ? ~0ull : 0ull;
    const bool lane_nn = (c == n) && (q == (n & 3));
    constexpr int REG_NN = n >> 2;

    // this trajectory's key-point entries [E0, E0 + NE): one descriptor over its slice of kpc (and of the raw payload)
    const int E0 = F.kp_offsets[(size_t)bP * F.dof], NE = F.kp_offsets[(size_t)(bP + 1) * F.dof] - E0;
    __amdgpu_buffer_rsrc_t rT = frsrc(F.kpc + (size_t)E0 * 3 * n, NE * strideB);
    const double *rb = F.r + (size_t)bR * (T + 1) * nr;
    const double *rxb = F.r_x + (size_t)bR * (T + 1) * nr * n;
    const double *rub = F.r_u + (size_t)bR * (T + 1) * nr * m;

    struct ResTiles { d4 Rx, R1, Ru; };
    // The sweep asks for the steps in order (T-1, T-2, ...), so the three residual arrays and the two gain arrays are walked
    // with running pointers: a subtraction per array and step instead of a 64-bit (trajectory, step) product (-24 scalar
    // instructions per step, 4.84 -> 4.77 ms).  `t` only says whether there IS a step t (the general form asks once more at the bottom).
    const double *pRx = rxb + (size_t)(T - 1) * nr * n, *pR = rb + (size_t)(T - 1) * nr, *pRu = rub + (size_t)(T - 1) * nr * m;
    auto load_res = [&](int t, ResTiles &s) {
        __amdgpu_buffer_rsrc_t rR = frsrc(pR, nr * 8);
        __amdgpu_buffer_rsrc_t rR2 = frsrc(pR, nr * 8 + 8);
        (void)rR2;
        __amdgpu_buffer_rsrc_t rRu = frsrc(pRu, nr * m * 8);
        if constexpr (!RXC) {
            __amdgpu_buffer_rsrc_t rRx = frsrc(pRx, nr * n * 8);
            s.Rx.x = fbld(rRx, oRx[0]); s.Rx.y = fbld(rRx, oRx[1]); s.Rx.z = fbld(rRx, oRx[2]); s.Rx.w = fbld(rRx, oRx[3]);
            if (t > 0) pRx -= nr * n;
        }
        if (t > 0) { pR -= nr; pRu -= nr * m; }       // (behind step 0: stay on it)
        if constexpr (RV2B) {
            // (descriptor one element longer: the last pair of an odd residual count reaches into the next row -- the array has T + 1)
            double r0, r1, r2, r3;
            fbld2s(rR2, oR12[0], 0, r0, r1); fbld2s(rR2, oR12[1], 0, r2, r3);
            s.R1.x = r0; s.R1.y = r1; s.R1.z = r2; s.R1.w = r3;
        } else {
        s.R1.x = fbld(rR, oR1[0]); s.R1.y = fbld(rR, oR1[1]); s.R1.z = fbld(rR, oR1[2]); s.R1.w = fbld(rR, oR1[3]);
        }
        if constexpr (!RU0) { s.Ru.x = fbld(rRu, oRu[0]); s.Ru.y = fbld(rRu, oRu[1]); s.Ru.z = fbld(rRu, oRu[2]); s.Ru.w = fbld(rRu, oRu[3]); }
    };

    // ---- column tracker, walking DOWN in time ----------------------------------------------------------
    const int kd = (c < F.dof) ? c : c - F.dof;
    const bool has = !PC && c < n;
    const int lo = has ? F.kp_offsets[(size_t)bP * F.dof + kd] : 0;
    const int hi = has ? F.kp_offsets[(size_t)bP * F.dof + kd + 1] : 0;
    int idx = hi - 1;
    int s = has ? F.kp_times[idx] : -1;                       // == T-1 for canonical key-points
    int nb = (has && idx - 1 >= lo) ? F.kp_times[idx - 1] : -1;      // the time of the next segment start, one crossing ahead
    double sv[8], av[8], pv[8];
    // SLP (general form on a differenced column store): the SLOPE of the next segment is prefetched with its start column (pm),
    // from the slope store k_kp_slopes wrote -- a crossing is then two register moves per value, no division on the wave's
    // serial chain (8 correctly rounded divisions + a reciprocal per lane before: the whole wave paid them on every step on which
    // ANY lane crossed, which with per-DoF lists is most steps)
    constexpr bool SLP = KP_SLOPES && !PC && !UNI && !RAW;
    // RAW: the prefetched column of the next segment start waits as x+ (in pv) and x- (pm) until the crossing differences it
    double pm[8];
    int pmode = 0;
    // the raw payload: ONE descriptor over the trajectory's records; x+ and x- of an element sit side by side (one 16-byte load
    // gives both: 9 loads per crossing instead of 17) and the mode word behind them
    constexpr int strideR = (6 * N + 2) * 8, offMode = 6 * N * 8, offM = 3 * N * 8;
    (void)offM;
    __amdgpu_buffer_rsrc_t rP = rT;                               // SLP: the trajectory's slice of the slope store: (value, slope) pairs
    ColOffsN co2 = co;
    co2.a = dbl_off(co.a); co2.al = dbl_off(co.al); co2.b = dbl_off(co.b); co2.bl = dbl_off(co.bl);
    auto ebase2 = [&](int e_rel) { return ((unsigned)e_rel < (unsigned)NE) ? e_rel * (2 * strideB) : BIGOFF; };
    (void)co2;
    if constexpr (SLP) rP = frsrc(F.kps + (size_t)E0 * 6 * n, NE * 2 * strideB);
    const int bitA = (c < F.dof) ? 1 : 2;                         // mode bit of this lane's A column (position / velocity job)
    auto ebase = [&](int e_rel) { return ((unsigned)e_rel < (unsigned)NE) ? e_rel * strideB : BIGOFF; };
    auto load_raw = [&](int e_rel, double *xp_, double *xm_, int &mo) {
        const int base = ((unsigned)e_rel < (unsigned)NE) ? e_rel * strideR : BIGOFF;
#if KP_RAW_PAIRS
        load_col2_n<N>(rP, co2, base, xp_, xm_);
#else
        load_col_n<N>(rP, co, base, xp_);
        load_col_n<N>(rP, co, base + offM, xm_);
#endif
        mo = __builtin_amdgcn_raw_buffer_load_b32(rP, base + ((c < n) ? offMode : BIGOFF), 0, 0);
    };
    double eps2 = F.eps2, rinv2 = F.rinv_2eps;
    auto difference_arith = [&](double *xp_, const double *xm_, int mo) {           // xp_ <- the differenced column
        // central differences are the rule (a control at its limit is the exception, Differentiator.cpp:94-143): one
        // wave-uniform test keeps the per-lane denominator selects off the usual path
        if (__builtin_amdgcn_ballot_w64(mo != 0) == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) xp_[i] = fdiv(xp_[i] - xm_[i], eps2, rinv2);
        } else {
            const bool oa = (mo & bitA) != 0, ob = (mo & 4) != 0;
            const double dA = oa ? 0.5 * eps2 : eps2, rA = oa ? 2.0 * rinv2 : rinv2;     // exact halves / doubles
            const double dB = ob ? 0.5 * eps2 : eps2, rB = ob ? 2.0 * rinv2 : rinv2;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xp_[i] = fdiv(xp_[i] - xm_[i], dA, rA);
                xp_[4 + i] = fdiv(xp_[4 + i] - xm_[4 + i], dB, rB);
            }
        }
    };
    auto difference = [&](double *xp_, const double *xm_, int mo, int e_rel) {      // ... and out to kpc
        difference_arith(xp_, xm_, mo);
        store_col_n<N>(rT, co, ebase(e_rel), xp_);
    };
    (void)pm; (void)pmode; (void)bitA;
    ResTiles cur;
    // CXX (RXC, one wave per trajectory; round 5): with ONE r_x for every step, l_xx = r_x' W r_x IS THE SAME MATRIX at every step
    // below the terminal one (running weights) -- only l_x = r_x' W r moves.  The four dependent products of Lzz = Rz' W Rz become
    //     Lzz~ = Cxx + 2 e_n v',      v(c) = sum_k r_x[k][c] 2 w_k r[k]
    // ONE product: lane (c, q) adds up its four rows k = 4r + q on the VALU (p, four FMAs on the resident r_x W tile and a broadcast
    // load of r), and the sum over q IS the contraction of a 16x16x4 product whose first operand is 2 e_n in every k --
    // D(i, j) = sum_q 2 e_n(i) p(j, q) -- accumulated onto the resident tile Cxx.  Row n then carries 2 l_x and column n nothing:
    // Lzz~ only ever enters V' = Qzz + K'G, which the step symmetrises ((V' + V'')/2, iLQR.cpp:610), and (Lzz~ + Lzz~')/2 = Lzz
    // (element (n, n), the constant of the value function, is dropped by the sweep anyway).  The terminal step (V = Lzz under the
    // terminal weights, :537-539) keeps the full product.  31.4 matrix instructions per step instead of 34.4, and a chain of four at
    // the top of the step becomes one.  The accumulation order differs from the per-step form's: results agree to ~1e-15, not bit for bit.
    constexpr bool CXX = KP_RXC_CXX && RXC && !PC;
    d4 Cxx = {0.0, 0.0, 0.0, 0.0}, RxW = Cxx;
    const double en2 = (c == n) ? 2.0 : 0.0;
    (void)Cxx; (void)RxW; (void)en2;
    if constexpr (RXC) {                           // the one r_x of the task: in registers for the whole sweep
        __amdgpu_buffer_rsrc_t rRx = frsrc(F.rx_const, nr * n * 8);
        cur.Rx.x = fbld(rRx, oRx[0]); cur.Rx.y = fbld(rRx, oRx[1]); cur.Rx.z = fbld(rRx, oRx[2]); cur.Rx.w = fbld(rRx, oRx[3]);
        if constexpr (CXX) {
            RxW.x = cur.Rx.x * w2run[0]; RxW.y = cur.Rx.y * w2run[1]; RxW.z = cur.Rx.z * w2run[2]; RxW.w = cur.Rx.w * w2run[3];
            Cxx = PR(cur.Rx, RxW, Cxx, ncr);
        }
    }
    // UNI: lane offsets of the lane's own DoF list (kd * KpU entries into the slice), the position in the soffset operand
    const int KpU = F.kp_offsets[(size_t)bP * F.dof + 1] - E0;
    ColOffsN cu = co, cr = co;                                 // kpc / raw payload
    auto shift = [&](int v, int by) { return v == BIGOFF ? BIGOFF : v + by; };
    if constexpr (UNI) {
        cu.a = shift(co.a, kd * KpU * strideB); cu.al = shift(co.al, kd * KpU * strideB);
        cu.b = shift(co.b, kd * KpU * strideB); cu.bl = shift(co.bl, kd * KpU * strideB);
#if KP_RAW_PAIRS
        cr.a = shift(co2.a, kd * KpU * strideR); cr.al = shift(co2.al, kd * KpU * strideR);      // (the payload's elements are 16 bytes)
        cr.b = shift(co2.b, kd * KpU * strideR); cr.bl = shift(co2.bl, kd * KpU * strideR);
#else
        cr.a = shift(co.a, kd * KpU * strideR); cr.al = shift(co.al, kd * KpU * strideR);
        cr.b = shift(co.b, kd * KpU * strideR); cr.bl = shift(co.bl, kd * KpU * strideR);
#endif
    }
    int up = KpU - 1;                                          // UNI: position (in every list) of the current segment's start
    int us = T, unb = T - 1, unb_v = T - 1;                    // UNI: its time, and the time of the next start (uniform; unb_v: as loaded)
    (void)up; (void)us; (void)unb; (void)unb_v; (void)cu; (void)cr;
    if constexpr (!PC && UNI) {
        // both columns of the first crossing are the last key-point's (slope 0 over the virtual segment [T-1, T])
        if constexpr (RAW) {
            rP = frsrc(F.fdk + (size_t)E0 * strideR, NE * strideR);
#if KP_RAW_PAIRS
            load_col2_n<N>(rP, cr, up * strideR, pv, pm);
#else
            load_col_n<N>(rP, cr, up * strideR, pv);
            load_col_n<N>(rP, cr, up * strideR + offM, pm);
#endif
            pmode = __builtin_amdgcn_raw_buffer_load_b32(rP, up * strideR + kd * KpU * strideR + ((c < n) ? offMode : BIGOFF), 0, 0);
#pragma unroll
            for (int i = 0; i < 8; i++) sv[i] = 0.0;
        } else {
            load_col_n<N>(rT, cu, up * strideB, pv);
#pragma unroll
            for (int i = 0; i < 8; i++) sv[i] = pv[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) av[i] = 0.0;
        load_res(T - 1, cur);
    } else if constexpr (!PC) {
        const int e_s = has ? idx - E0 : -1, e_nb = (has && idx - 1 >= lo) ? idx - 1 - E0 : -1;
        if constexpr (RAW) {
            rP = frsrc(F.fdk + (size_t)E0 * strideR, NE * strideR);
            int mo0;
            load_raw(e_s, sv, pm, mo0);
            difference(sv, pm, mo0, e_s);
            load_raw(e_nb, pv, pm, pmode);
        } else {
            load_col_n<N>(rT, co, ebase(e_s), sv);
            if constexpr (SLP) load_col2_n<N>(rP, co2, ebase2(e_nb), pv, pm);
            else load_col_n<N>(rT, co, ebase(e_nb), pv);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) av[i] = 0.0;
        // Fz(n,n) = 1: lane c == n walks no list (never crosses), so its constant start value carries the 1
#pragma unroll
        for (int r = 0; r < 4; r++) if (c == n && 4 * r + q == n) sv[r] = 1.0;
        // single-buffered: the residual tiles are consumed by the first MFMAs of a step and re-requested for the
        // next step right behind them (a whole step of latency cover, no register copies)
        load_res(T - 1, cur);
    } else {
        __syncthreads();                               // the producer has published step T-1
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    d4 V = zero, accp = zero;                                  // accp (SIDE): V' of the step before, not yet symmetrised
    (void)accp;
    d4 W2 = {w2term[0], w2term[1], w2term[2], w2term[3]};     // terminal weights at t = T-1, running after
    int pd_counter = 0, fail = 0;
    double dJ = 0.0;
    d4 Xinv = zero, Xprev = zero, Iu;            // running inverse of Quu + lambda I (KP_NS), the one before it, the identity of the u-block
    bool haveX = false;
    Iu.x = (q == c && c < m) ? 1.0 : 0.0; Iu.y = (4 + q == c && c < m) ? 1.0 : 0.0;
    Iu.z = (8 + q == c && c < m) ? 1.0 : 0.0; Iu.w = (12 + q == c && c < m) ? 1.0 : 0.0;
    (void)haveX; (void)Iu;

    // The gains of a step leave the wave one step late (one-wave forms), in front of the next residual requests: a wait
    // for those requests also waits for every store issued behind them (the counter is shared and in order), and stores
    // issued at the end of a step are acknowledged by a loaded memory system long after the requests have come back --
    // the wait at the top of every step sat out that acknowledgement.  In front of the requests the stores are a step old
    // when anything waits for them.
    d4 Kst = zero;
    int tst = -1, kst_pos = -1;
    (void)Kst; (void)tst; (void)kst_pos;
    double *pK = Kout + ((size_t)bS * T + T - 1) * m * n, *pk = kout + ((size_t)bS * T + T - 1) * m;
    auto store_gains = [&](int t, const d4 &Kp) {
        (void)t;                                       // the steps are stored in order, T-1 first
        __amdgpu_buffer_rsrc_t rK = frsrc(pK, m * n * 8);
        __amdgpu_buffer_rsrc_t rk = frsrc(pk, m * 8);
        pK -= m * n; pk -= m;
        const double kv[4] = {Kp.x, Kp.y, Kp.z, Kp.w};
#pragma unroll
        for (int r = 0; r < NCU; r++) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, kv[r]), rK, oKst[r], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, kv[r]), rk, okst[r], 0, 0);
        }
    };
    // one step of the sweep; false: the PD check of this step failed (the sweep ends)
#ifdef KP_CYC
    unsigned long long cyc_cross = 0, cyc_peel = 0, cyc_inner = 0, cyc_a = 0, cyc_b = 0, cyc_c = 0, cyc_d = 0, cyc_e = 0;
    const unsigned long long cyc_all0 = __builtin_readcyclecounter();
#endif
    // may_be_first: whether this call site can see the terminal step t = T-1 (only the first step of a sweep can: the call
    // sites inside the loops say no, and the selects of the terminal value function and weights leave the hot path)
    auto step = [&](int t, auto may_be_first) __attribute__((always_inline)) -> bool {
#ifdef KP_CYC
        const unsigned long long cyc_s0 = __builtin_readcyclecounter();
#endif
        d4 Fz, Fu, Lzz, LU;
        const bool term = decltype(may_be_first)::value && (t == T - 1);
        bool cross = false;                            // general form: this lane has crossed the start of its segment
        (void)cross;
        if constexpr (PC) {
            const double *tb = pcbuf + (t & 1) * FPC_BUF;
            Fu = lds_tile4(tb + 256, lane);
            if constexpr (RU0) LU = zero; else LU = lds_tile4(tb + 768, lane);
            if constexpr (SIDE) {                     // Fz, Lzz are the side wave's operands; only V = Lzz of the last step is ours
                Fz = zero; Lzz = zero;
                if (term) Lzz = lds_tile4(tb + 512, lane);
                else {
                    // (V' + V'')/2 of the step before: its transposed half is read here, behind the barrier that ended that
                    // step, together with this step's tiles -- no write / wait / read round trip at the end of a step
                    V.x = 0.5 * (accp.x + sh[FLDS_V + c * FVS + q]);
                    V.y = 0.5 * (accp.y + sh[FLDS_V + c * FVS + 4 + q]);
                    V.z = 0.5 * (accp.z + sh[FLDS_V + c * FVS + 8 + q]);
                    V.w = 0.5 * (accp.w + sh[FLDS_V + c * FVS + 12 + q]);
                    if (lane_nn) fset_reg<REG_NN>(V, 0.0);
                }
            } else {
                Fz = lds_tile4(tb, lane); Lzz = lds_tile4(tb + 512, lane);
            }
        } else {
        // ---- a4: this step's A and B columns --------------------------------------------------------------
        // General form (per-lane lists): a lane that has crossed the start of its segment takes the column prefetched a
        // crossing ago as the new start and forms the slope -- arithmetic only up here.  Everything of the crossing that is
        // a memory operation (the differenced column out to kpc, the next column and its time in) waits for `late_cross`
        // below, BEHIND the wait for this step's residual tiles: in front of it the wait, which the compiler has to place
        // conservatively behind a divergent branch, drained the requests the branch had just issued -- a trip to HBM on
        // every step on which any lane crossed (ragged key-point sets: most steps; 8.89 -> 8.32 and 9.61 -> 8.72 ms on two such sets).
        if constexpr (!UNI) {
            cross = t < s;
            if (cross) {
                if constexpr (SLP) {
                    // the new segment [nb, s): its start column and its slope were requested a crossing ago
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        asm volatile("v_mov_b64 %0, %1" : "+v"(av[i]) : "v"(pm[i]));
                        asm volatile("v_mov_b64 %0, %1" : "+v"(sv[i]) : "v"(pv[i]));
                    }
                } else {
                if constexpr (RAW) { const KArgF Fk = kernarg_fused(); eps2 = Fk->eps2; rinv2 = Fk->rinv_2eps; }
                const double den = (double)(s - nb);
                const double rinv = kp_rcp(den);
                if constexpr (RAW) difference_arith(pv, pm, pmode);      // the prefetched x+ / x- become the column
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    av[i] = fdiv(sv[i] - pv[i], den, rinv);
                    // a real move, in sv's own register: left to the compiler, sv is renamed onto pv's register, the loads below
                    // land somewhere else, and the two arrays are copied back and forth -- behind a wait for every memory
                    // operation in flight -- on EVERY step of the sweep instead of at the crossings
                    asm volatile("v_mov_b64 %0, %1" : "+v"(sv[i]) : "v"(pv[i]));
                }
                }
                idx--;
                s = nb;
            }
        }
        const double dt = UNI ? (double)(t - us) : (double)(t - s);
        Fz.x = lerp_nc(sv[0], dt, av[0]); Fz.y = lerp_nc(sv[1], dt, av[1]);
        Fz.z = lerp_nc(sv[2], dt, av[2]); Fz.w = lerp_nc(sv[3], dt, av[3]);
        Fu.x = lerp_nc(sv[4], dt, av[4]); Fu.y = lerp_nc(sv[5], dt, av[5]);
        Fu.z = lerp_nc(sv[6], dt, av[6]); Fu.w = lerp_nc(sv[7], dt, av[7]);
        // ---- a6: Lzz, l_uu, l_u from the residuals --------------------------------------------------------
        if constexpr (CXX) {
            if (term) {                                 // (compile-time false away from a sweep's first step)
                d4 Rz;
                Rz.x = bits_or(cur.Rx.x, bits_and(cur.R1.x, mask_n)); Rz.y = bits_or(cur.Rx.y, bits_and(cur.R1.y, mask_n));
                Rz.z = bits_or(cur.Rx.z, bits_and(cur.R1.z, mask_n)); Rz.w = bits_or(cur.Rx.w, bits_and(cur.R1.w, mask_n));
                Lzz = PR(Rz, Rz * W2, zero, ncr);
            } else {
                double p = RxW.x * cur.R1.x;
                p = __builtin_fma(RxW.y, cur.R1.y, p); p = __builtin_fma(RxW.z, cur.R1.z, p); p = __builtin_fma(RxW.w, cur.R1.w, p);
                Lzz = MFMA(en2, p, Cxx);
            }
        } else {
        d4 Rz;
        Rz.x = bits_or(cur.Rx.x, cur.R1.x); Rz.y = bits_or(cur.Rx.y, cur.R1.y);
        Rz.z = bits_or(cur.Rx.z, cur.R1.z); Rz.w = bits_or(cur.Rx.w, cur.R1.w);
        Lzz = PR(Rz, Rz * W2, zero, ncr);
        }
        if constexpr (RU0) {
            LU = zero;
        } else {
            d4 Rur;
            Rur.x = bits_or(cur.Ru.x, cur.R1.x); Rur.y = bits_or(cur.Ru.y, cur.R1.y);
            Rur.z = bits_or(cur.Ru.z, cur.R1.z); Rur.w = bits_or(cur.Ru.w, cur.R1.w);
            LU = PR(cur.Ru, Rur * W2, zero, ncr);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the step above, IN FRONT of the requests.  (The test is false only at the first step; compiling it out of the loops'
        // call sites was measured and is 0.04 ms SLOWER at B=1024 -- the scheduler's placement of the stores changes.)
        if constexpr (KP_BWD_LATE_STORE) { if (tst >= 0) store_gains(tst, Kst); }
        if constexpr (!UNI) {
            if (cross) {                               // late_cross: the memory operations of the crossing at the top of this step
                const KArgF Fk = kernarg_fused();      // crossing-only scalars, from the kernel-argument segment
                rT = frsrc(Fk->kpc + (size_t)E0 * 3 * n, NE * strideB);
                if constexpr (RAW) {
                    rP = frsrc(Fk->fdk + (size_t)E0 * strideR, NE * strideR);
                    store_col_n<N>(rT, co, ebase(idx - E0), sv);          // the column that has just become the start value
                }
                nb = (idx - 1 >= lo) ? Fk->kp_times[idx - 1] : -1;
                const int e_nb = (idx - 1 >= lo) ? idx - 1 - E0 : -1;
                if constexpr (RAW) load_raw(e_nb, pv, pm, pmode);
                else if constexpr (SLP) {
                    rP = frsrc(Fk->kps + (size_t)E0 * 6 * n, NE * 2 * strideB);
                    load_col2_n<N>(rP, co2, ebase2(e_nb), pv, pm);             // start column AND slope of the next segment: one load each
                } else load_col_n<N>(rT, co, ebase(e_nb), pv);
            }
        }
        if constexpr (RAW && UNI) {
            // the differenced column of the crossing above (it is the segment's start value now): like the gains it leaves
            // behind the wait for this step's tiles -- in front of it, the wait sat out the stores' acknowledgement
            if (kst_pos >= 0) { store_col_n<N>(rT, cu, kst_pos * strideB, sv); kst_pos = -1; }
        }
        // (unconditional: a request behind `if (t > 0)` costs a scalar branch per step and makes the compiler's later waits
        // conservative -- in the general form its wait for the crossing's columns drained the tiles; 4.76 -> 4.70 ms in the
        // uniform form.  Behind step 0 the request repeats step 0's tiles, never used.)
        load_res(t > 0 ? t - 1 : 0, cur);
        __builtin_amdgcn_sched_barrier(0);
#ifdef KP_CYC
        cyc_a += __builtin_readcyclecounter() - cyc_s0;
#endif
        }
        d4 Luu, Luz;
        Luu.x = bits_and(LU.x, mask_u); Luu.y = bits_and(LU.y, mask_u); Luu.z = bits_and(LU.z, mask_u); Luu.w = bits_and(LU.w, mask_u);
        Luz.x = bits_and(LU.x, mask_n); Luz.y = bits_and(LU.y, mask_n); Luz.z = bits_and(LU.z, mask_n); Luz.w = bits_and(LU.w, mask_n);
        if (term) {                                     // V_x = l_x[T-1]; V_xx = l_xx[T-1]   (iLQR.cpp:537-539)
            V = Lzz;
            W2.x = w2run[0]; W2.y = w2run[1]; W2.z = w2run[2]; W2.w = w2run[3];
        }
        pd_counter++;
        const bool check_pd = pd_counter >= pd_stride;

        // ---- Tu = V' Fu ; Quu = l_uu + Fu' Tu --------------------------------------- :577
        // (one wave per trajectory: the regularisation rides in the accumulator's initial value, Qr = (l_uu + lambda I) + Fu'Tu -- nothing
        // downstream wants the un-regularised Quu: V' and delta_J are written on Qr, G and lambda below -- so the adds leave the chain
        // behind the product: 4.15 -> 4.10 ms at B=1024.  The consumer wave of the pair measured 1 % SLOWER with it and keeps the adds.)
        d4 Tu = PS<NCZ>(V, Fu, zero);
        d4 Qr = PS<NCZ>(Fu, Tu, PC ? Luu : Luu + LamI);          // Quu (+ lambda I)
        if constexpr (PC) { Qr.x += LamI.x; Qr.y += LamI.y; Qr.z += LamI.z; Qr.w += LamI.w; }
        // ---- Tz, Quz, Qzz --------------------------------------------------------------- :570-579
        d4 Quz, Qzz;
        if constexpr (SIDE) {
            Quz = zero; Qzz = zero;                   // the side wave forms them meanwhile (fusedpc_side): read behind the refresh
        } else {
            d4 Tz = PS<NCZ>(V, Fz, zero);
            Quz = PS<NCZ>(Fu, Tz, Luz);
            Qzz = PS<NCZ>(Fz, Tz, Lzz);
        }
#ifdef KP_CYC
        asm volatile("" :: "v"(Qr.x), "v"(Quz.x), "v"(Qzz.x));
        const unsigned long long cyc_s1 = __builtin_readcyclecounter();
        cyc_b += cyc_s1 - cyc_s0;
#endif

        // ---- X = (Quu + lambda I)^-1 Quz.  Fast path (KP_NS): the inverse changes little from one step to the next,
        //      so it is refreshed by Newton-Schulz steps Xinv <- Xinv + Xinv (I - Q Xinv) on the matrix core (4 MFMAs
        //      each, quadratic convergence; the count is chosen from the measured residual so that it ends below
        //      1e-15) and X = Xinv Quz is one more product -- the reference, too, forms the explicit inverse and
        //      multiplies (iLQR.cpp:597-604).  The LDL' path below runs on the first step, on every checked step
        //      (it gives the PD verdict of :587-595), and whenever the residual is too large to converge fast.
        bool done = false;
#if KP_NS
        int ns_steps = 0;
        // (segment-loop forms: the peeled step is the one right below a key-point -- known at compile time)
        constexpr bool KINK = KP_KINK4 && UNI && !PC && decltype(may_be_first)::value;
        const bool refreshed = haveX && !check_pd && kp_inverse_refresh_n<NCU, KINK, PC>(Qr, Iu, Xinv, Xprev, m, STATS ? &ns_steps : nullptr);    // Xinv, Xprev: NEGATED inverses
        if constexpr (STATS) { if (refreshed) hcnt[ns_steps < 0 ? 0 : ns_steps > 3 ? 3 : ns_steps]++; }
#else
        const bool refreshed = false;
#endif
        if constexpr (SIDE) {                         // mid-step barrier of the triple: Quz, Qzz are published
            __syncthreads();
            Quz = lds_tile4(pcbuf + FPC_SIDE_QUZ, lane); Qzz = lds_tile4(pcbuf + FPC_SIDE_QZZ, lane);
        }
#ifdef KP_CYC
        asm volatile("" :: "v"(Xinv.x));
        const unsigned long long cyc_s2 = __builtin_readcyclecounter();
        cyc_c += cyc_s2 - cyc_s1;
#endif
        d4 Kp = zero;                                 // the gains -X
        if (refreshed) {
            Kp = PS<NCU>(Xinv, Quz, zero);               // -(Quu + lambda I)^-1 Quz: the gains, with their sign
            done = true;
        }
        if (!done) {
            sh[FLDS_Q + q * FMS + c] = Qr.x;
            if (NCU > 1) sh[FLDS_Q + (4 + q) * FMS + c] = Qr.y;
            if (NCU > 2) sh[FLDS_Q + (8 + q) * FMS + c] = Qr.z;
            if (NCU > 3) sh[FLDS_Q + (12 + q) * FMS + c] = Qr.w;
            sh[FLDS_Z + c * FMZ + q] = Quz.x;
            if (NCU > 1) sh[FLDS_Z + c * FMZ + 4 + q] = Quz.y;
            if (NCU > 2) sh[FLDS_Z + c * FMZ + 8 + q] = Quz.z;
            if (NCU > 3) sh[FLDS_Z + c * FMZ + 12 + q] = Quz.w;
            wsync();
            // ---- unpivoted LDL' of Quu + lambda I, redundantly in every lane ----
            double Lm[M][M], rd[M];
            const bool pos = kp_ldl_factor<M>([&](int i, int j) { return sh[FLDS_Q + i * FMS + j]; }, Lm, rd);
            if (check_pd) {                       // CheckMatrixPD every pd_stride steps   :587-595
                if (!pos) {
                    fail = t + 1;
                    if (PC) { if (lane == 0) sflag[0] = fail; __syncthreads(); }      // the producer leaves with us
                    return false;
                }
                pd_counter = 0;
            }
            double x[M];
            if (pos) {
#pragma unroll
                for (int i = 0; i < M; i++) x[i] = sh[FLDS_Z + c * FMZ + i];
                kp_ldl_solve<M>(Lm, rd, x);
#if KP_NS
                // seed the fast path: column c of the inverse in lane c (c < m), as a tile
                double y[M];
#pragma unroll
                for (int i = 0; i < M; i++) y[i] = (i == c) ? 1.0 : 0.0;
                kp_ldl_solve<M>(Lm, rd, y);
                double yr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < M; i++)
                    if (q == (i & 3)) yr[i >> 2] = (c < m) ? y[i] : 0.0;
                Xinv.x = -yr[0]; Xinv.y = -yr[1]; Xinv.z = -yr[2]; Xinv.w = -yr[3];       // the running inverse is kept negated
                Xprev = Xinv;
                haveX = true;
#endif
                if constexpr (STATS) hcnt[4]++;
            } else {
                if constexpr (STATS) hcnt[5]++;
                double *wa = sh + FLDS_SLOW, *wx = wa + 256, *wt = wx + 256;
                int *tr = (int *)(wt + 16);
                if (lane == 0) kp_slow_ldlt_inverse(L.m, sh + FLDS_Q, FMS, wa, wx, wt, tr);
                wsync();
#pragma unroll
                for (int i = 0; i < M; i++) {
                    double sacc = 0.0;
#pragma unroll
                    for (int p = 0; p < M; p++) sacc += (-wx[i + p * m]) * sh[FLDS_Z + c * FMZ + p];
                    x[i] = -sacc;
                }
                wsync();
                haveX = false;
            }
            double xr[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < M; i++)
                if (q == (i & 3)) xr[i >> 2] = x[i];
            Kp.x = -xr[0]; Kp.y = -xr[1]; Kp.z = -xr[2]; Kp.w = -xr[3];
        }
        if constexpr (KP_BWD_LATE_STORE && !PC) { Kst = Kp; tst = t; }
        else store_gains(t, Kp);
        // delta_J += k'Q_u + k'Q_uu k = -lambda k'k (:612-613): lanes of column n keep the squares of their rows,
        // the four row groups are added once after the sweep
        // (every lane accumulates; only the lanes of column n are read behind the sweep)
        // (the squares are summed, -lambda joins behind the sweep: one FMA per register and step)
        dJ = __builtin_fma(Kp.x, Kp.x, dJ);
        if (NCU > 1) dJ = __builtin_fma(Kp.y, Kp.y, dJ);
        if (NCU > 2) dJ = __builtin_fma(Kp.z, Kp.z, dJ);
        if (NCU > 3) dJ = __builtin_fma(Kp.w, Kp.w, dJ);
        // V' = Qzz + K'Quu K + K'Quz + Quz'K (:606-607) with K = -X, (Quu + lambda I) X = Quz:
        //    = Qzz - X'(Quz + lambda X) = Qzz + K'(Quz - lambda K)   -- one product; G = (Quu + 2 lambda I)K' is never formed
        d4 G;
        G.x = __builtin_fma(-lam, Kp.x, Quz.x); G.y = __builtin_fma(-lam, Kp.y, Quz.y);
        G.z = __builtin_fma(-lam, Kp.z, Quz.z); G.w = __builtin_fma(-lam, Kp.w, Quz.w);
        d4 acc = PS<NCU>(Kp, G, Qzz);
#ifdef KP_CYC
        asm volatile("" :: "v"(acc.x));
        const unsigned long long cyc_s3 = __builtin_readcyclecounter();
        cyc_d += cyc_s3 - cyc_s2;
#endif
        sh[FLDS_V + (q) * FVS + c] = acc.x;
        sh[FLDS_V + (4 + q) * FVS + c] = acc.y;
        sh[FLDS_V + (8 + q) * FVS + c] = acc.z;
        sh[FLDS_V + (12 + q) * FVS + c] = acc.w;
        if constexpr (SIDE) {
            accp = acc;                                // symmetrised at the top of the next step, by this wave and by the side wave
        } else {
            wsync();
            V.x = 0.5 * (acc.x + sh[FLDS_V + c * FVS + q]);
            V.y = 0.5 * (acc.y + sh[FLDS_V + c * FVS + 4 + q]);
            V.z = 0.5 * (acc.z + sh[FLDS_V + c * FVS + 8 + q]);
            V.w = 0.5 * (acc.w + sh[FLDS_V + c * FVS + 12 + q]);
            if (lane_nn) fset_reg<REG_NN>(V, 0.0);
        }
        if (PC) __syncthreads();                       // end of step: the ring slot is free, the next one is full
        else wsync();
#ifdef KP_CYC
        asm volatile("" :: "v"(V.x));
        cyc_e += __builtin_readcyclecounter() - cyc_s3;
#endif
        return true;
    };
    if constexpr (UNI && !PC) {
        // (Tried: ONE loop over the steps with the crossing's arithmetic behind a uniform branch at the top of the step and its
        // memory operations behind the tile wait, as in the forward sweep and in the general form -- 5.64 against 4.81 ms, also
        // with a key-point every 20 steps (5.20 against 4.41): the loop as a whole came out slower, not the crossing.)
        // segments from the top: [k_p, k_p+1) with k_KpU := T.  The crossing -- slope of the new segment from the column
        // prefetched a segment ago, the next column requested (raw: x+ / x- differenced and kept for the forward sweep) --
        // is straight-line code; the segment's first step is peeled behind it so that the wait for its residual tiles
        // (requested BEFORE the crossing's loads) can leave those in flight.
        bool ok = true;
        for (; up >= 0 && ok; up--) {
#ifdef KP_CYC
            const unsigned long long cyc0 = __builtin_readcyclecounter();
#endif
            const KArgF Fk = kernarg_fused();                      // crossing-only scalars, from the kernel-argument segment
            rT = frsrc(Fk->kpc + (size_t)E0 * 3 * n, NE * strideB);
            if constexpr (RAW) { rP = frsrc(Fk->fdk + (size_t)E0 * strideR, NE * strideR); eps2 = Fk->eps2; rinv2 = Fk->rinv_2eps; }
            // the time of the new segment's start was requested a crossing ago, BEHIND that crossing's column loads: memory
            // operations return in order, and the wait for the residual tiles of the peeled step ("all but the youngest
            // loads") would otherwise have to sit out this load's latency too (measured: +840 cycles on every peeled step)
            unb = __builtin_amdgcn_readfirstlane(unb_v);
            const int gap = us - unb;                              // steps of the new segment (k_p+1 - k_p; 1 for the first)
            const double den = (double)gap, rinv = kp_rcp(den);
            if constexpr (RAW) {
                if (__builtin_amdgcn_ballot_w64(pmode != 0) == 0) {
#pragma unroll
                    for (int i = 0; i < 8; i++) pv[i] = fdiv(pv[i] - pm[i], eps2, rinv2);
                } else {
                    const bool oa = (pmode & bitA) != 0, ob = (pmode & 4) != 0;
                    const double dA = oa ? 0.5 * eps2 : eps2, rA = oa ? 2.0 * rinv2 : rinv2;     // exact halves / doubles
                    const double dB = ob ? 0.5 * eps2 : eps2, rB = ob ? 2.0 * rinv2 : rinv2;
#pragma unroll
                    for (int i = 0; i < 4; i++) { pv[i] = fdiv(pv[i] - pm[i], dA, rA); pv[4 + i] = fdiv(pv[4 + i] - pm[4 + i], dB, rB); }
                }
                kst_pos = up;                                      // stored from sv behind the tile wait of the step below
                if (up == KpU - 1) {
#pragma unroll
                    for (int i = 0; i < 8; i++) sv[i] = pv[i];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) { av[i] = fdiv(sv[i] - pv[i], den, rinv); sv[i] = pv[i]; }
            if (c == n) {                                          // Fz(n,n) = 1 rides in the idle lane's constant start value
#pragma unroll
                for (int r = 0; r < 4; r++) if (4 * r + q == n) { sv[r] = 1.0; av[r] = 0.0; }
            }
            const int t_hi = us - 1;
            us = unb;
            const int pn = up > 0 ? up - 1 : 0;                    // the next start (position 0 again at the bottom: never used)
            if constexpr (RAW) {
#if KP_RAW_PAIRS
                load_col2_n<N>(rP, cr, pn * strideR, pv, pm);
#else
                load_col_n<N>(rP, cr, pn * strideR, pv);
                load_col_n<N>(rP, cr, pn * strideR + offM, pm);
#endif
                pmode = __builtin_amdgcn_raw_buffer_load_b32(rP, pn * strideR + kd * KpU * strideR + ((c < n) ? offMode : BIGOFF), 0, 0);
            } else {
                load_col_n<N>(rT, cu, pn * strideB, pv);
            }
            __builtin_amdgcn_sched_barrier(0);
            unb_v = Fk->kp_times[E0 + pn];                         // behind the column loads (see above)
            __builtin_amdgcn_sched_barrier(0);
#ifdef KP_CYC
            const unsigned long long cyc1 = __builtin_readcyclecounter();
            cyc_cross += cyc1 - cyc0;
#endif
            ok = step(t_hi, std::true_type{});                     // peeled: straight-line behind the crossing
#ifdef KP_CYC
            const unsigned long long cyc2 = __builtin_readcyclecounter();
            cyc_peel += cyc2 - cyc1;
#endif
            for (int t = t_hi - 1; t >= us && ok; t--) ok = step(t, std::false_type{});
#ifdef KP_CYC
            cyc_inner += __builtin_readcyclecounter() - cyc2;
#endif
        }
    } else {
        if (step(T - 1, std::true_type{}))
            for (int t = T - 2; t >= 0; t--) if (!step(t, std::false_type{})) break;
    }
    if constexpr (KP_BWD_LATE_STORE && !PC) { if (tst >= 0) store_gains(tst, Kst); }      // the last completed step
    dJ *= -lam;
    dJ += __shfl_xor(dJ, 16);
    dJ += __shfl_xor(dJ, 32);
    if (lane_nn) delta_J[b] = dJ;
#ifdef KP_CYC
    if (lane_nn) { delta_J[b] = (double)cyc_cross; Kout[(size_t)b * T * m * n] = (double)cyc_peel; Kout[(size_t)b * T * m * n + 1] = (double)cyc_inner;
                   Kout[(size_t)b * T * m * n + 2] = (double)(__builtin_readcyclecounter() - cyc_all0); Kout[(size_t)b * T * m * n + 3] = (double)cyc_a;
                   Kout[(size_t)b * T * m * n + 4] = (double)cyc_b; Kout[(size_t)b * T * m * n + 5] = (double)cyc_c;
                   Kout[(size_t)b * T * m * n + 6] = (double)cyc_d; Kout[(size_t)b * T * m * n + 7] = (double)cyc_e; }
#endif
    if (lane == 0) status[b] = fail;
    if constexpr (STATS) {
        if (lane == 0) for (int i = 0; i < 6; i++) hist[(size_t)b * 6 + i] = hcnt[i];
    }
}

template <int N, int M>
__global__ void __launch_bounds__(64)
k_backward_fused_stats(RecLayout L, FusedArgs F, int T, const double *__restrict__ lambda,
                       int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                       double *__restrict__ delta_J, int *__restrict__ status, int *__restrict__ hist)
{
    __shared__ __attribute__((aligned(16))) double sh[FLDS_TOTAL];
    backward_fused_body<N, M, false, false, false, false, false, true>(sh, nullptr, nullptr, L, F, T, lambda, pd_stride, Kout, kout, delta_J, status, hist);
}

// The one-wave backward sweep comes in two forms, launched back to back like the forward sweep's: UNI for key-point sets in
// which every DoF of a trajectory has the same list, the general form otherwise; each looks at the device flag first and
// leaves if the set is not its kind.
template <int N, int M, bool RU0, bool RAW, bool UNI, bool RXC = false>
__global__ void __launch_bounds__(64)
k_backward_fused(RecLayout L, FusedArgs F, int T, const double *__restrict__ lambda,
                 int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                 double *__restrict__ delta_J, int *__restrict__ status, const int *__restrict__ kp_uniform)
{
    __shared__ __attribute__((aligned(16))) double sh[FLDS_TOTAL];
    if ((*kp_uniform != 0) != UNI) return;
    backward_fused_body<N, M, false, RU0, false, RAW, UNI, false, RXC>(sh, nullptr, nullptr, L, F, T, lambda, pd_stride, Kout, kout, delta_J, status);
}
template <int N, int M, bool RU0, bool RAW, bool UNI, bool RXC = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_backward_fused_excl(RecLayout L, FusedArgs F, int T, const double *__restrict__ lambda,
                      int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                      double *__restrict__ delta_J, int *__restrict__ status, const int *__restrict__ kp_uniform)
{
    __shared__ __attribute__((aligned(16))) double sh[FLDS_TOTAL];
    if ((*kp_uniform != 0) != UNI) return;
    backward_fused_body<N, M, false, RU0, false, RAW, UNI, false, RXC>(sh, nullptr, nullptr, L, F, T, lambda, pd_stride, Kout, kout, delta_J, status);
}


// ---------------------------------------------------------------------------------------------------------
// Column trackers of the helper wave (consumer / helper pair below): a4 walking DOWN in time, NV values per lane.
template <int NV>
__device__ __forceinline__ void load_vals(__amdgpu_buffer_rsrc_t rT, const int *offs, int tk, int T, int strideB, double *out)
{
    const int base = ((unsigned)tk < (unsigned)T) ? tk * strideB : BIGOFF;
#pragma unroll
    for (int r = 0; r < NV; r++) out[r] = fbld(rT, base + offs[r]);
}


__device__ __forceinline__ d4 lds_tile4(const double *t, int lane)
{
    d4 v; v.x = t[lane]; v.y = t[64 + lane]; v.z = t[128 + lane]; v.w = t[192 + lane];
    return v;
}
__device__ __forceinline__ void lds_store4(double *t, int lane, const d4 &v)
{
    t[lane] = v.x; t[64 + lane] = v.y; t[128 + lane] = v.z; t[192 + lane] = v.w;
}

// Column tracker walking down for NV values per lane (a4), shared by the two roles.
template <int NV>
struct DownTracker {
    int offs[NV];
    int lo, idx, s, nb, nb2;
    double sv[NV], av[NV], pv[NV];
    // rT: the trajectory's slice of kpc, entries [E0, E0 + NE) of strideB bytes each (see load_col)
    int E0, NE;
    __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t rT, const int *kp_offsets, const int *kp_times, bool has, size_t list, int E0_, int NE_, int strideB)
    {
        E0 = E0_; NE = NE_;
        lo = has ? kp_offsets[list] : 0;
        const int hi = has ? kp_offsets[list + 1] : 0;
        idx = hi - 1;
        s = has ? kp_times[idx] : -1;
        nb = (has && idx - 1 >= lo) ? kp_times[idx - 1] : -1;
        nb2 = (has && idx - 2 >= lo) ? kp_times[idx - 2] : -1;
        load_vals<NV>(rT, offs, has ? idx - E0 : -1, NE, strideB, sv);
        load_vals<NV>(rT, offs, (has && idx - 1 >= lo) ? idx - 1 - E0 : -1, NE, strideB, pv);
#pragma unroll
        for (int i = 0; i < NV; i++) av[i] = 0.0;
    }
    // a crossing in two halves: its arithmetic, and -- behind the wave's wait for
    // its residual tiles -- the request of the next column (a wait placed behind the divergent branch is conservative and would
    // sit out requests issued in front of it)
    bool need = false;
    __device__ __forceinline__ void cross(int t)
    {
        need = t < s;
        if (need) {
            const double den = (double)(s - nb);
            const double rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const double ev = sv[i];
                sv[i] = pv[i];
                av[i] = fdiv(ev - sv[i], den, rinv);
            }
            s = nb; idx--;
            nb = nb2;
        }
    }
    __device__ __forceinline__ void request(__amdgpu_buffer_rsrc_t rT, const int *kp_times, int strideB)
    {
        if (need) {
            nb2 = (idx - 2 >= lo) ? kp_times[idx - 2] : -1;
            load_vals<NV>(rT, offs, (idx - 1 >= lo) ? idx - 1 - E0 : -1, NE, strideB, pv);
        }
    }
    __device__ __forceinline__ double value(int i, double dt) const { return lerp_nc(sv[i], dt, av[i]); }
};

// The same on the SLOPE STORE (kps [entry][3][n][2], (value, slope) pairs: k_kp_slopes / k_fd_kp_difference<SLOPES>) -- per-DoF lists
// in the helper wave of the consumer / helper pair: a crossing is register moves and 16-byte loads, no division (with the lists of
// reaching.yaml's velocity_change some lane crosses on ~40 % of the steps).  The requests are issued by ALL lanes behind a
// wave-uniform branch (a lane that did not cross asks again for the entry it holds): loads inside a per-lane branch meet the
// old values at its join, and the wait the compiler places there drains every request in flight.
struct DownTrackerSlp {
    static constexpr int NV = 8;
    int offs[NV];                                // byte offsets inside a kpc entry (doubled for the slope store)
    int lo, idx, s, nb, nb2;
    double sv[NV], av[NV], pv[NV], pa[NV];
    int E0, NE;
    bool need = false;
    __device__ __forceinline__ void load_pairs(__amdgpu_buffer_rsrc_t rS, int e_rel, int stride2, double *v, double *a) const
    {
        const int base = ((unsigned)e_rel < (unsigned)NE) ? e_rel * stride2 : BIGOFF;
#pragma unroll
        for (int r = 0; r < NV; r++) fbld2(rS, base + dbl_off(offs[r]), v[r], a[r]);
    }
    __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t rS, const int *kp_offsets, const int *kp_times, bool has, size_t list, int E0_, int NE_, int stride2)
    {
        E0 = E0_; NE = NE_;
        lo = has ? kp_offsets[list] : 0;
        const int hi = has ? kp_offsets[list + 1] : 0;
        idx = hi - 1;
        s = has ? kp_times[idx] : -1;
        nb = (has && idx - 1 >= lo) ? kp_times[idx - 1] : -1;
        nb2 = (has && idx - 2 >= lo) ? kp_times[idx - 2] : -1;
        load_pairs(rS, has ? idx - E0 : -1, stride2, sv, av);                     // (the last key-point's slope is 0)
        load_pairs(rS, (has && idx - 1 >= lo) ? idx - 1 - E0 : -1, stride2, pv, pa);
    }
    __device__ __forceinline__ void cross(int t)
    {
        need = t < s;                            // per lane: crossed the start of the current segment
        if (need) {
#pragma unroll
            for (int i = 0; i < NV; i++) { sv[i] = pv[i]; av[i] = pa[i]; }
            s = nb; idx--;
            nb = nb2;
        }
    }
    __device__ __forceinline__ void request(__amdgpu_buffer_rsrc_t rS, const int *kp_times, int stride2)
    {
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            nb2 = (idx - 2 >= lo) ? kp_times[idx - 2] : -1;
            load_pairs(rS, (idx - 1 >= lo) ? idx - 1 - E0 : -1, stride2, pv, pa);
        }
    }
    __device__ __forceinline__ double value(int i, double dt) const { return lerp_nc(sv[i], dt, av[i]); }
};

// The same walking the KEY-POINT ORDERED FD PAYLOAD (producer wave of the pair / triple, RAWP): the prefetched x+ / x- of the
// next segment start are differenced when the lane reaches it -- (x+ - x-) / (2 eps), / eps for a one-sided job: the arithmetic
// and the bytes of k_fd_kp_difference -- and the column goes out to kpc for the forward sweep.  The producer is a step ahead of the
// consumer and not the longer wave: the differencing costs the sweep nothing (a streaming kernel in front of it: 0.09 ... 0.41 ms
// at 128 ... 512 trajectories).  Values 0..3 of a lane belong to its A column (kind 0 / 1), 4..7 to its B column (kind 2).
struct DownTrackerRaw {
    static constexpr int NV = 8;
    int offs[NV];
    int lo, idx, s, nb, nb2, pmode, bitA;
    double sv[NV], av[NV], pv[NV], pm[NV];
    int E0, NE, strideR, offM, offMode, strideB;
    bool has, fresh;                             // fresh: pv / pm still hold x+ / x- (not differenced yet)
    __device__ __forceinline__ void load_raw(__amdgpu_buffer_rsrc_t rP, int e_rel, double *xp, double *xm, int &mo) const
    {
        const int base = ((unsigned)e_rel < (unsigned)NE) ? e_rel * strideR : BIGOFF;
#pragma unroll
#if KP_RAW_PAIRS
        for (int r = 0; r < NV; r++) fbld2(rP, base + dbl_off(offs[r]), xp[r], xm[r]);      // (x+, x-) of an element: one 16-byte load
#else
        for (int r = 0; r < NV; r++) { xp[r] = fbld(rP, base + offs[r]); xm[r] = fbld(rP, base + offM + offs[r]); }
#endif
        mo = __builtin_amdgcn_raw_buffer_load_b32(rP, base + (has ? offMode : BIGOFF), 0, 0);
    }
    __device__ __forceinline__ void diff(double *xp, const double *xm, int mo, double eps2, double rinv2) const
    {
        const bool oa = (mo & bitA) != 0, ob = (mo & 4) != 0;
        const double dA = oa ? 0.5 * eps2 : eps2, rA = oa ? 2.0 * rinv2 : rinv2;     // exact halves / doubles
        const double dB = ob ? 0.5 * eps2 : eps2, rB = ob ? 2.0 * rinv2 : rinv2;
#pragma unroll
        for (int i = 0; i < 4; i++) { xp[i] = fdiv(xp[i] - xm[i], dA, rA); xp[4 + i] = fdiv(xp[4 + i] - xm[4 + i], dB, rB); }
    }
    __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rT, int e_rel, const double *x) const
    {
        const int base = ((unsigned)e_rel < (unsigned)NE) ? e_rel * strideB : BIGOFF;
#pragma unroll
        for (int r = 0; r < NV; r++) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2f, x[r]), rT, base + offs[r], 0, 0);
    }
    __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t rT, __amdgpu_buffer_rsrc_t rP, const int *kp_offsets, const int *kp_times, bool has_,
                                         size_t list, int E0_, int NE_, int n, bool pos_col, double eps2, double rinv2)
    {
        E0 = E0_; NE = NE_; has = has_;
        strideB = 3 * n * 8; strideR = (6 * n + 2) * 8; offM = 3 * n * 8; offMode = 6 * n * 8;
        bitA = pos_col ? 1 : 2;
        lo = has ? kp_offsets[list] : 0;
        const int hi = has ? kp_offsets[list + 1] : 0;
        idx = hi - 1;
        s = has ? kp_times[idx] : -1;
        nb = (has && idx - 1 >= lo) ? kp_times[idx - 1] : -1;
        nb2 = (has && idx - 2 >= lo) ? kp_times[idx - 2] : -1;
        int mo0;
        load_raw(rP, has ? idx - E0 : -1, sv, pm, mo0);
        diff(sv, pm, mo0, eps2, rinv2);
        store(rT, has ? idx - E0 : -1, sv);
        load_raw(rP, (has && idx - 1 >= lo) ? idx - 1 - E0 : -1, pv, pm, pmode);
        fresh = has && idx - 1 >= lo;                // (nothing below the list's first key-point: nothing to difference or store)
#pragma unroll
        for (int i = 0; i < NV; i++) av[i] = 0.0;
    }
    // the prefetched x+ / x- differenced (and stored) as soon as they have arrived -- a step behind their request, on a step
    // that has time for it -- so that the crossing itself is left with the slopes and the next requests
    __device__ __forceinline__ void settle(__amdgpu_buffer_rsrc_t rT, double eps2, double rinv2)
    {
        if (fresh) {
            diff(pv, pm, pmode, eps2, rinv2);
            store(rT, idx - 1 - E0, pv);
            fresh = false;
        }
    }
    bool need = false;
    __device__ __forceinline__ void cross(__amdgpu_buffer_rsrc_t rT, int t, double eps2, double rinv2)
    {
        need = t < s;
        if (need) {
            const double den = (double)(s - nb);
            const double rinv = kp_rcp(den);
            settle(rT, eps2, rinv2);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const double ev = sv[i];
                sv[i] = pv[i];
                av[i] = fdiv(ev - sv[i], den, rinv);
            }
            s = nb; idx--;
            nb = nb2;
        }
    }
    __device__ __forceinline__ void request(__amdgpu_buffer_rsrc_t rP, const int *kp_times)
    {
        if (need) {
            nb2 = (idx - 2 >= lo) ? kp_times[idx - 2] : -1;
            load_raw(rP, (idx - 1 >= lo) ? idx - 1 - E0 : -1, pv, pm, pmode);
            fresh = idx - 1 >= lo;
        }
    }
    __device__ __forceinline__ double value(int i, double dt) const { return lerp_nc(sv[i], dt, av[i]); }
};

// ---------------------------------------------------------------------------------------------------------
// Backward pass, PRODUCER / CONSUMER wave pair per trajectory.  Unlike the U/Z split above the dependency runs one
// way only: the producer wave evaluates what does not depend on V' -- this step's A, B columns (a4) and the cost
// tiles Lzz = Rz'WRz, [l_uu | l_u] = Ru'W[Ru | r] (a6) -- one step AHEAD of the consumer and hands them over through a
// two-slot LDS ring; the consumer wave runs the Riccati chain proper (backward_fused_body<.., PC = true>).  One
// s_barrier per step: at the end of step t the consumer has read slot t&1 and the producer has filled slot (t-1)&1.
// Two waves per SIMD at batch = #SIMDs: the producer's independent MFMAs and FP64 FMAs issue into the bubbles of the
// consumer's dependent chain.
// HELPER (consumer / helper pair, 256 < batch <= 512: two SIMDs per trajectory): ONE wave is the triple's producer AND its side
// wave -- at the top of step t the side products Tz = V Fz, Quz, Qzz (operands kept in registers since they were published), and,
// between the mid-step barrier and the end of the step (while the consumer forms the gains and V'), the tiles of step t-1.
// RU0 / RXC (helper only): r_u = 0 -- no r_u loads, no [l_uu | l_u] product, the ring's LU tiles stay zero; ONE constant r_x in
// registers (see backward_fused_body).
// SLP (helper only, differenced column store): per-DoF lists walked on the slope store (DownTrackerSlp).
template <int N, int M, bool TRIPLE = false, bool RAWP = false, bool HELPER = false, bool RU0 = false, bool RXC = false, bool SLP = false>
__device__ __forceinline__ void fusedpc_producer(double *pcbuf, int *sflag, RecLayout L, FusedArgs F, int T, const double *sh = nullptr)
{
    static_assert(HELPER && TRIPLE, "only the helper wave of the consumer / helper pair is left (two barriers per step)");
    static_assert(!SLP || (HELPER && !RAWP), "the slope store serves the helper on a differenced column store");
    static_assert(HELPER || (!RU0 && !RXC), "RU0 / RXC are the helper's instantiations");
    static_assert(!RXC || RU0, "RXC comes with RU0");
    constexpr int n = N, m = M;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int b = KP_BLOCK_TRAJ;
    const int nr = F.nr, ncr = (nr + 3) >> 2;
    const int strideB = 3 * L.n * 8;                               // bytes of one key-point entry of kpc: three columns
    typename std::conditional<RAWP, DownTrackerRaw, typename std::conditional<SLP, DownTrackerSlp, DownTracker<8>>::type>::type tr;       // A rows then B rows of column c
    int oRx[4], oR1[4], oRu[4];
    d4 Wt, Wr;
    {
        double wt[4], wr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + q;
            tr.offs[r] = (row < n && c < n) ? 8 * ((c < F.dof ? 0 : n) + row) : BIGOFF;
            tr.offs[4 + r] = (row < n && c < m) ? 8 * (2 * n + row) : BIGOFF;
            oRx[r] = (row < nr && c < n) ? 8 * (row * n + c) : OOBF;
            oR1[r] = (row < nr && (c == n || (KP_RXC_CXX && RXC))) ? 8 * row : OOBF;      // (RXC: r[k] in every column, see CXX in backward_fused_body)
            oRu[r] = (row < nr && c < m) ? 8 * (row * m + c) : OOBF;
            wr[r] = (row < nr) ? 2.0 * F.w_run[row] : 0.0;
            wt[r] = (row < nr) ? 2.0 * F.w_term[row] : 0.0;
        }
        Wt.x = wt[0]; Wt.y = wt[1]; Wt.z = wt[2]; Wt.w = wt[3];
        Wr.x = wr[0]; Wr.y = wr[1]; Wr.z = wr[2]; Wr.w = wr[3];
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const int E0 = F.kp_offsets[(size_t)b * F.dof], NE = F.kp_offsets[(size_t)(b + 1) * F.dof] - E0;      // this trajectory's key-point entries
    __amdgpu_buffer_rsrc_t rT = frsrc(F.kpc + (size_t)E0 * 3 * L.n, NE * strideB);
    const double *rb = F.r + (size_t)b * (T + 1) * nr;
    const double *rxb = F.r_x + (size_t)b * (T + 1) * nr * n;
    const double *rub = F.r_u + (size_t)b * (T + 1) * nr * m;
    d4 Rx, R1, Ru;
    // (HELPER: the steps are asked for in order, T-1 first -- running pointers, as in backward_fused_body; with the 64-bit
    // (step x size) products the compiler formed the descriptors on the VALU and wrapped every load in a waterfall loop)
    const double *pRx = rxb + (size_t)(T - 1) * nr * n, *pR1 = rb + (size_t)(T - 1) * nr, *pRu = rub + (size_t)(T - 1) * nr * m;
    (void)pRx; (void)pR1; (void)pRu;
    auto load_res = [&](int t) {
        const bool ok = t >= 0;
        __amdgpu_buffer_rsrc_t rA = frsrc(pRx, ok ? nr * n * 8 : 0);
        __amdgpu_buffer_rsrc_t rR = frsrc(pR1, ok ? nr * 8 : 0);
        __amdgpu_buffer_rsrc_t rU = frsrc(pRu, ok ? nr * m * 8 : 0);
        if (t > 0) { pRx -= nr * n; pR1 -= nr; pRu -= nr * m; }
        if constexpr (!RXC) { Rx.x = fbld(rA, oRx[0]); Rx.y = fbld(rA, oRx[1]); Rx.z = fbld(rA, oRx[2]); Rx.w = fbld(rA, oRx[3]); }
        R1.x = fbld(rR, oR1[0]); R1.y = fbld(rR, oR1[1]); R1.z = fbld(rR, oR1[2]); R1.w = fbld(rR, oR1[3]);
        if constexpr (!RU0) { Ru.x = fbld(rU, oRu[0]); Ru.y = fbld(rU, oRu[1]); Ru.z = fbld(rU, oRu[2]); Ru.w = fbld(rU, oRu[3]); }
    };
    // CXX: the constant block r_x' W r_x as a resident tile, Lzz~ = Cxx + 2 e_n (r_x' W r)' by ONE product (backward_fused_body)
    constexpr bool CXX = KP_RXC_CXX && RXC;
    d4 Cxx = zero, RxW = zero;
    const double en2 = (c == n) ? 2.0 : 0.0;
    const u64 mask_r1 = (c == n) ? ~0ull : 0ull;
    (void)Cxx; (void)RxW; (void)en2; (void)mask_r1;
    if constexpr (RXC) {                           // the one r_x of the task: in registers for the whole sweep
        __amdgpu_buffer_rsrc_t rA = frsrc(F.rx_const, nr * n * 8);
        Rx.x = fbld(rA, oRx[0]); Rx.y = fbld(rA, oRx[1]); Rx.z = fbld(rA, oRx[2]); Rx.w = fbld(rA, oRx[3]);
        if constexpr (CXX) { RxW = Rx * Wr; Cxx = PR(Rx, RxW, zero, ncr); }
    }
    if constexpr (RU0) {                           // [l_uu | l_u] = 0: both ring slots once (the consumer does not read them either)
        Ru = zero;
        lds_store4(pcbuf + 768, lane, zero); lds_store4(pcbuf + FPC_BUF + 768, lane, zero);
    }
    const int kd = (c < F.dof) ? c : c - F.dof;
    __amdgpu_buffer_rsrc_t rP = rT;
    if constexpr (RAWP) {
        rP = frsrc(F.fdk + (size_t)E0 * (6 * n + 2) * 8, NE * (6 * n + 2) * 8);
        tr.init(rT, rP, F.kp_offsets, F.kp_times, c < n, (size_t)b * F.dof + kd, E0, NE, n, c < F.dof, F.eps2, F.rinv_2eps);
    } else if constexpr (SLP) {
        rP = frsrc(F.kps + (size_t)E0 * 6 * L.n, NE * 2 * strideB);              // the trajectory's slice of the slope store
        tr.init(rP, F.kp_offsets, F.kp_times, c < n, (size_t)b * F.dof + kd, E0, NE, 2 * strideB);
    } else {
        tr.init(rT, F.kp_offsets, F.kp_times, c < n, (size_t)b * F.dof + kd, E0, NE, strideB);
    }
    (void)rP;
#pragma unroll
    for (int r = 0; r < 4; r++) if (c == n && 4 * r + q == n) tr.sv[r] = 1.0;       // Fz(n,n) = 1
    d4 hFz = zero, hFu = zero, hLzz = zero, hLU = zero;            // HELPER: the tiles published last, for the side products of their step
    (void)hFz; (void)hFu; (void)hLzz; (void)hLU;
    auto publish = [&](int t, const d4 &W2, auto terminal) __attribute__((always_inline)) {
        const double dt = (double)(t - tr.s);
        d4 Fz, Fu, Rz, Rur, Lzz;
        Fz.x = tr.value(0, dt); Fz.y = tr.value(1, dt); Fz.z = tr.value(2, dt); Fz.w = tr.value(3, dt);
        Fu.x = tr.value(4, dt); Fu.y = tr.value(5, dt); Fu.z = tr.value(6, dt); Fu.w = tr.value(7, dt);
        if constexpr (CXX && !decltype(terminal)::value) {
            double p = RxW.x * R1.x;
            p = __builtin_fma(RxW.y, R1.y, p); p = __builtin_fma(RxW.z, R1.z, p); p = __builtin_fma(RxW.w, R1.w, p);
            Lzz = MFMA(en2, p, Cxx);
        } else {
            if constexpr (CXX) {
                Rz.x = bits_or(Rx.x, bits_and(R1.x, mask_r1)); Rz.y = bits_or(Rx.y, bits_and(R1.y, mask_r1));
                Rz.z = bits_or(Rx.z, bits_and(R1.z, mask_r1)); Rz.w = bits_or(Rx.w, bits_and(R1.w, mask_r1));
            } else {
                Rz.x = bits_or(Rx.x, R1.x); Rz.y = bits_or(Rx.y, R1.y); Rz.z = bits_or(Rx.z, R1.z); Rz.w = bits_or(Rx.w, R1.w);
            }
            Lzz = PR(Rz, Rz * W2, zero, ncr);
        }
        d4 LU = zero;
        if constexpr (!RU0) {
            Rur.x = bits_or(Ru.x, R1.x); Rur.y = bits_or(Ru.y, R1.y); Rur.z = bits_or(Ru.z, R1.z); Rur.w = bits_or(Ru.w, R1.w);
            LU = PR(Ru, Rur * W2, zero, ncr);
        }
        load_res(t - 1);
        double *tb = pcbuf + (t & 1) * FPC_BUF;
        // (Fz is the side products' operand only: it stays in this wave's registers)
        lds_store4(tb + 256, lane, Fu);
        // (Lzz: the consumer of the triple / helper pair reads it at the terminal step only, V = Lzz(T-1); the side products take it
        // from registers here)
        if (t == T - 1) lds_store4(tb + 512, lane, Lzz);
        if constexpr (!RU0) lds_store4(tb + 768, lane, LU);
        if constexpr (HELPER) { hFz = Fz; hFu = Fu; hLzz = Lzz; hLU = LU; }
    };
    load_res(T - 1);
    if (lane == 0) sflag[0] = 0;
    publish(T - 1, Wt, std::true_type{});                 // terminal weights   (iLQR.cpp:537-539)
    __syncthreads();
    if constexpr (HELPER) {
        constexpr int NCZ = (N + 1 + 3) / 4;
        constexpr int REG_NN = n >> 2;
        const bool lane_nn = (c == n) && (q == (n & 3));
        const u64 mask_n = (c == n) ? ~0ull : 0ull;
        // everything of step t that hangs on V but not on the gains (fusedpc_side), from the tiles this wave published for it
        auto side = [&](int t) __attribute__((always_inline)) {
            d4 V = hLzz;                                   // V_xx = l_xx[T-1]   (iLQR.cpp:537-539)
            if (t < T - 1) {                               // (V' + V'')/2 from the consumer's unsymmetrised image, as the consumer forms it
                V.x = 0.5 * (sh[FLDS_V + (q) * FVS + c] + sh[FLDS_V + c * FVS + q]);
                V.y = 0.5 * (sh[FLDS_V + (4 + q) * FVS + c] + sh[FLDS_V + c * FVS + 4 + q]);
                V.z = 0.5 * (sh[FLDS_V + (8 + q) * FVS + c] + sh[FLDS_V + c * FVS + 8 + q]);
                V.w = 0.5 * (sh[FLDS_V + (12 + q) * FVS + c] + sh[FLDS_V + c * FVS + 12 + q]);
                if (lane_nn) fset_reg<REG_NN>(V, 0.0);
            }
            d4 Luz;
            Luz.x = bits_and(hLU.x, mask_n); Luz.y = bits_and(hLU.y, mask_n); Luz.z = bits_and(hLU.z, mask_n); Luz.w = bits_and(hLU.w, mask_n);
            const d4 Tz = PS<NCZ>(V, hFz, zero);
            const d4 Quz = PS<NCZ>(hFu, Tz, Luz);
            const d4 Qzz = PS<NCZ>(hFz, Tz, hLzz);
            lds_store4(pcbuf + FPC_SIDE_QUZ, lane, Quz);
            lds_store4(pcbuf + FPC_SIDE_QZZ, lane, Qzz);
        };
        for (int t = T - 1; t > 0; t--) {
            side(t);
            __syncthreads();                               // mid-step: the consumer takes Quz, Qzz
            // the tiles of step t-1, while the consumer forms the gains and V' of step t (slot (t-1)&1 was last read at the top of step t+1)
            // (the crossing's arithmetic in front of the wait for the residual tiles, its requests behind it)
            // (the prefetched x+ / x- are differenced BEHIND the publish, whose wait for the residual tiles has let them arrive)
            if constexpr (RAWP) tr.cross(rT, t - 1, F.eps2, F.rinv_2eps);
            else tr.cross(t - 1);
            publish(t - 1, Wr, std::false_type{});
            if constexpr (RAWP) { tr.settle(rT, F.eps2, F.rinv_2eps); tr.request(rP, F.kp_times); }
            else if constexpr (SLP) tr.request(rP, F.kp_times, 2 * strideB);
            else tr.request(rT, F.kp_times, strideB);
            __syncthreads();                               // end of step: V of step t-1 is there
            if (__builtin_amdgcn_readfirstlane(sflag[0])) return;     // (wave-uniform: a divergent exit makes t a per-lane value and every request a waterfall loop)
        }
        side(0);
        __syncthreads();
        __syncthreads();
        return;
    }
}

#define FPC_RING FLDS_TOTAL
#define FPC_FLAG (FPC_RING + 2 * FPC_BUF + 2 * 256)
#define FPC_TOTAL (FPC_FLAG + 2)
// guard: -1 run, 1 run only when the device flag says the key-point set is uniform, 0 only when it is not (the raw helper is
// launched for uniform sets; per-DoF lists go through k_fd_kp_difference and the plain / slope-store helper, launched behind it)
// Consumer / helper pair (2 x batch <= #SIMDs: a SIMD each): the triple's consumer, and ONE wave for its side and producer roles
template <int N, int M, bool RAWP, bool RU0, bool RXC, bool SLP = false>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_backward_fusedph(RecLayout L, FusedArgs F, int T, int role_shift, const double *__restrict__ lambda,
                   int pd_stride, double *__restrict__ Kout, double *__restrict__ kout,
                   double *__restrict__ delta_J, int *__restrict__ status, const int *__restrict__ kp_uniform, int guard)
{
    __shared__ __attribute__((aligned(16))) double sh[FPC_TOTAL];
    if (guard >= 0 && (*kp_uniform != 0) != (guard != 0)) return;
    const bool consumer = ((__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) ^ (blockIdx.x >> role_shift)) & 1) == 0;
    if (consumer)
        backward_fused_body<N, M, true, RU0, true>(sh, sh + FPC_RING, (int *)(sh + FPC_FLAG), L, F, T, lambda, pd_stride, Kout, kout,
                                                   delta_J, status);
    else
        fusedpc_producer<N, M, true, RAWP, true, RU0, RXC, SLP>(sh + FPC_RING, (int *)(sh + FPC_FLAG), L, F, T, sh);
}
// ---------------------------------------------------------------------------------------------------------
// Forward pass.  The column tracker walks UP in time; its tiles (row = A row, col = A column) are turned into
// the Y operands (row = contraction index) through a padded LDS transpose, off the Z dependency chain.
// The time loop of the forward sweep by register set: NS consecutive steps with compile-time set indices, and what is left
// below a full group at the end (MODE 1 ends in the final step, mode 3, at t = T-1; MODE 0 has no final step of its own).
template <int K, int NS, int MODE, class StepF>
__device__ __forceinline__ void fwd_steps(StepF &step, int t)
{
    step(t, std::integral_constant<int, MODE>{}, std::integral_constant<int, K>{});
    if constexpr (K + 1 < NS) fwd_steps<K + 1, NS, MODE>(step, t + 1);
}
template <int K, int NS, int MODE, class StepF>
__device__ __forceinline__ void fwd_tail(StepF &step, int t, int T)
{
    if constexpr (MODE == 0) {
        if (t < T) {
            step(t, std::integral_constant<int, 0>{}, std::integral_constant<int, K>{});
            if constexpr (K + 1 < NS) fwd_tail<K + 1, NS, 0>(step, t + 1, T);
        }
    } else {
        if (K + 1 == NS || t == T - 1) step(T - 1, std::integral_constant<int, 3>{}, std::integral_constant<int, K>{});
        else {
            step(t, std::integral_constant<int, MODE>{}, std::integral_constant<int, K>{});
            if constexpr (K + 1 < NS) fwd_tail<K + 1, NS, MODE>(step, t + 1, T);
        }
    }
}

// RXC: one r_x [nr][n] for every trajectory and step (see backward_fused_body): its transposed tile stays in registers, the
// sweep issues no r_x loads (4 of a step's 15) and the four tile sets shrink by a tile each.
template <int NCZ, int NCU, bool RU0 = false, bool UNI = false, bool RXC = false>
__device__ __forceinline__ void forward_fused_body(RecLayout L, FusedArgs F, int T, int n_alpha, const double *__restrict__ Kin, const double *__restrict__ kin,
               const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
               const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha)
{
    __shared__ __attribute__((aligned(16))) double shA[16 * 17], shB[16 * 17];
    // The chunk counts name the shape (KP_T1_SHAPES: (14,7) -> <4,2>, (4,1) -> <2,1>, (12,3) -> <4,1>, (10,3) -> <3,1>; the
    // launcher dispatches nothing else), so n and m are compile-time constants here as in the backward sweep: the row of k in
    // the gain operand, the chunk count of r_x dx and the row masks cost no selects or scalar branches per step.
    constexpr int n = (NCZ == 4 && NCU == 2) ? 14 : (NCZ == 2) ? 4 : (NCZ == 4) ? 12 : 10;
    constexpr int m = (NCZ == 4 && NCU == 2) ? 7 : (NCZ == 2) ? 1 : 3;
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const int b = KP_BLOCK_TRAJ;
    const int nr = F.nr;
    constexpr int strideB = 3 * n * 8;                             // bytes of one key-point entry of kpc: three columns
    constexpr int ncx = (n + 3) >> 2;

    // this trajectory's key-point entries [E0, E0 + NE) of kpc; KpU: entries per DoF list when all lists are the same (UNI)
    const int E0 = F.kp_offsets[(size_t)b * F.dof], NE = F.kp_offsets[(size_t)(b + 1) * F.dof] - E0;
    const int KpU = F.kp_offsets[(size_t)b * F.dof + 1] - E0;
    ColOffs co;
    if (!UNI) col_offsets(co, n, m, F.dof, c, q);
    int oK[4], oRxT[4], oRuT[4], oR[4], oub[4];
    double lo[NCU], hi[NCU], wcur[4];
    // RV2 (uniform key-point sets): the residuals sit in the rows of Jx, Ju in the order
    //     row 4r + q  <->  residual sig(4r + q) = 8 (r >> 1) + 2 q + (r & 1)
    // so that registers (0, 1) and (2, 3) of a lane are CONSECUTIVE residuals and r_t arrives with two 16-byte requests instead of four
    // 8-byte ones (a request costs this sweep ~24 cycles of its 1 330-cycle step whatever it carries: timing probe in
    // profiles/r05_headline_ab.txt).  The order is a relabelling of the rows of the resident r_x tile and of the per-row constants.
    constexpr bool RV2 = KP_FWD_RV2 && UNI;        // (the per-DoF list form measured 1.3 % SLOWER with it: 2.31 against 2.28 ms)
    auto sig = [](int i) { return RV2 ? 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1) : i; };
    const int cs = sig(c);                                                  // the residual in column c of the RxT operand
    const int oR2[2] = {(2 * q < nr) ? 16 * q : OOBF, (8 + 2 * q < nr) ? 64 + 16 * q : OOBF};
    (void)oR2;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        if (UNI) {
            // the Y operands themselves: Ya(p = row, o = c) = A(o, p), Yb(p, o) = B(o, p) -- element c of column `row`, whose
            // DoF list starts (row mod dof) * KpU entries into the trajectory's slice (equal lists: the same position in each)
            const int d = row < F.dof ? row : row - F.dof;
            co.a[r] = (row < n && c < n) ? 8 * ((d * KpU * 3 + (row < F.dof ? 0 : 1)) * n + c) : BIGOFF;
            co.b[r] = (row < m && c < n) ? 8 * ((row * KpU * 3 + 2) * n + c) : BIGOFF;
        }
        oK[r] = (row < n && c < m) ? 8 * (row * m + c) : OOBF;
        oRxT[r] = (row < n && cs < nr) ? 8 * (cs * n + row) : OOBF;   // RxT(p=row, k=c) = r_x[k][p]   (RV2: residual sig(c) in column c)
        oRuT[r] = (row < m && cs < nr) ? 8 * (cs * m + row) : OOBF;   // RuT(p=row, k=c) = r_u[k][p]
        oR[r] = (row < nr) ? 8 * row : OOBF;                          // r[k=row] (rows of Jx / Ju; RV2: oR2)
        oub[r] = (row < m) ? 8 * row : OOBF;
        if (r < NCU) {
            lo[r] = (row < m) ? ctrl_lim[2 * row] : -1.0e300;
            hi[r] = (row < m) ? ctrl_lim[2 * row + 1] : 1.0e300;
        }
        wcur[r] = (sig(row) < nr) ? F.w_run[sig(row)] : 0.0;
    }
    constexpr int rn = n >> 2;                                              // the register of row n (k)
    const int okn = (q == (n & 3) && c < m) ? 8 * c : OOBF;
    const double my_alpha = (c < n_alpha) ? alphas[c] : 0.0;
    d4 Z;
    {
        double zr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + q;
            zr[r] = (row == n) ? my_alpha : (row == n + 1) ? 1.0 : 0.0;
        }
        Z.x = zr[0]; Z.y = zr[1]; Z.z = zr[2]; Z.w = zr[3];
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double partial = 0.0;

    __amdgpu_buffer_rsrc_t rT = frsrc(F.kpc + (size_t)E0 * 3 * L.n, NE * strideB);
    const double *rb = F.r + (size_t)b * (T + 1) * nr;
    const double *rxb = F.r_x + (size_t)b * (T + 1) * nr * n;
    const double *rub = F.r_u + (size_t)b * (T + 1) * nr * m;

    struct Tiles { d4 YkK, RxT, RuT, rv, ub; double kk; };      // kk: this lane's element of k (row n of Yk lives in ONE register)
    auto load_tiles = [&](int t, Tiles &s) {
        __amdgpu_buffer_rsrc_t rK = frsrc(Kin + ((size_t)b * T + t) * m * n, m * n * 8);
        __amdgpu_buffer_rsrc_t rk = frsrc(kin + ((size_t)b * T + t) * m, m * 8);
        __amdgpu_buffer_rsrc_t ru = frsrc(u_nom + ((size_t)b * T + t) * m, m * 8);
        __amdgpu_buffer_rsrc_t rRx = frsrc(rxb + (size_t)t * nr * n, nr * n * 8);
        __amdgpu_buffer_rsrc_t rRu = frsrc(rub + (size_t)t * nr * m, nr * m * 8);
        __amdgpu_buffer_rsrc_t rR = frsrc(rb + (size_t)t * nr, nr * 8);
        s.YkK.x = fbld(rK, oK[0]); s.YkK.y = fbld(rK, oK[1]); s.YkK.z = fbld(rK, oK[2]); s.YkK.w = fbld(rK, oK[3]);
        s.kk = fbld(rk, okn);
        if constexpr (!RXC) { s.RxT.x = fbld(rRx, oRxT[0]); s.RxT.y = fbld(rRx, oRxT[1]); s.RxT.z = fbld(rRx, oRxT[2]); s.RxT.w = fbld(rRx, oRxT[3]); }
        if constexpr (!RU0) {
            s.RuT.x = fbld(rRu, oRuT[0]); s.RuT.y = NCU > 1 ? fbld(rRu, oRuT[1]) : 0.0;
            s.RuT.z = NCU > 2 ? fbld(rRu, oRuT[2]) : 0.0; s.RuT.w = NCU > 3 ? fbld(rRu, oRuT[3]) : 0.0;
        }
        if constexpr (RV2) {
            // (descriptor one element longer: the last pair of an odd residual count reaches into the next step's row -- the array has
            // T + 1 rows, the weights beyond nr are zero)
            __amdgpu_buffer_rsrc_t rR2 = frsrc(rb + (size_t)t * nr, nr * 8 + 8);
            double r0, r1, r2, r3;
            fbld2s(rR2, oR2[0], 0, r0, r1); fbld2s(rR2, oR2[1], 0, r2, r3);
            s.rv.x = r0; s.rv.y = r1; s.rv.z = r2; s.rv.w = r3;
        } else {
        s.rv.x = fbld(rR, oR[0]); s.rv.y = fbld(rR, oR[1]); s.rv.z = fbld(rR, oR[2]); s.rv.w = fbld(rR, oR[3]);
        }
        s.ub.x = fbld(ru, oub[0]); s.ub.y = NCU > 1 ? fbld(ru, oub[1]) : 0.0;
        s.ub.z = NCU > 2 ? fbld(ru, oub[2]) : 0.0; s.ub.w = NCU > 3 ? fbld(ru, oub[3]) : 0.0;
    };

    // ---- column tracker, walking UP in time --------------------------------------------------------------
    // UNI: every DoF of the trajectory has the SAME key-point list (set_interval, and whatever else comes out uniform), so
    // one list serves all four registers of a lane and the lanes interpolate the transposed operands Ya, Yb directly --
    // lane (q, c) holds A(c, 4r+q): row c of A, a different COLUMN (DoF list) per register otherwise, which is why the
    // general form interpolates column-wise and transposes through LDS
    const int kd = UNI ? 0 : (c < F.dof) ? c : c - F.dof;
    const bool has = c < n;
    const int klo = has ? F.kp_offsets[(size_t)b * F.dof + kd] : 0;
    const int khi = has ? F.kp_offsets[(size_t)b * F.dof + kd + 1] : 0;
    int idx = klo;
    int s = has ? F.kp_times[idx] : 0;                                  // == 0 for canonical key-points
    int e = (has && idx + 1 < khi) ? F.kp_times[idx + 1] : BIGT;
    int nb = (has && idx + 2 < khi) ? F.kp_times[idx + 2] : BIGT;
    // Walking up, a segment is entered at its START key-point, where the stored value is exact and the slope is
    // not needed yet: the end column is requested at the crossing and the slope formed one step later
    // (`pend`), so no second prefetch buffer is held.
    // General form (!UNI): the slope store of k_kp_slopes gives every segment's slope with its start column -- requested one
    // crossing ahead (ev, ea), so a crossing is register moves and nothing is waited for at the step behind it.
    double sv[8], ev[8], av[8], ea[8];
    (void)ea;
    __amdgpu_buffer_rsrc_t rS = rT;
    load_col(rT, co, has ? idx - E0 : -1, NE, strideB, sv);
    load_col(rT, co, (has && idx + 1 < khi) ? idx + 1 - E0 : -1, NE, strideB, ev);
    bool pend = true;
#pragma unroll
    for (int i = 0; i < 8; i++) av[i] = 0.0;
    constexpr bool FSLP = KP_SLOPES && !UNI;
    if constexpr (FSLP) {
        rS = frsrc(F.kps + (size_t)E0 * 6 * L.n, NE * 2 * strideB);
        load_col2(rS, co, has ? idx - E0 : -1, NE, 2 * strideB, sv, av);
        load_col2(rS, co, (has && idx + 1 < khi) ? idx + 1 - E0 : -1, NE, 2 * strideB, ev, ea);
        pend = false;
    }
    // the identity rows of Ya (alpha and the homogeneous 1 carry over): lanes c >= n walk no list
#pragma unroll
    for (int r = 0; r < 4; r++) if ((c == n || c == n + 1) && 4 * r + q == c) sv[r] = 1.0;

    // Two register sets: S0 holds the tiles of the even steps, S1 those of the odd ones; step t re-requests ITS set for step
    // t+2 right behind each tile's last use.  A wavefront's loads return in order and a step is shorter than a trip to HBM
    // (the sweep ran at one memory latency per step with a single set: 2.66 ms at B = 128 where its arithmetic needs 2.13,
    // -DKP_PROBE_SAMEB), so the requests have to be two steps ahead of their use.
    // (the general form with control residuals keeps four sets: with six its largest shape spills -- 100 bytes of scratch, tools/isa_lint.py)
    constexpr int NS = UNI ? KP_FWD_SETS : (RU0 || KP_FWD_SETS_GEN < 4) ? KP_FWD_SETS_GEN : 4;
    Tiles S[NS];
    d4 RxTc = {0.0, 0.0, 0.0, 0.0}, RxTt = RxTc;
    // SQW (round 5; constant r_x, no control residuals, uniform key-point sets -- the headline's forward sweep): the ROWS of the
    // resident r_x are scaled by sqrt(w_k) once (a second tile carries the terminal weights), so the product gives Js = sqrt(w) Jx and
    //     sum_k w_k Jx_k (2 r_k + Jx_k)  =  sum_k Js_k (2 sqrt(w_k) r_k + Js_k)
    // is two FMAs per register instead of an add, a multiply, an FMA and the doubling of r: 8 VALU instructions less of the step's
    // 79 (a lone wave pays for each).  sqrt(w) is correctly rounded; the costs move by ~1e-16 relative (held to the oracle at 1e-9).
    // A NEGATIVE weight (legal: the reference just forms w r^2) keeps its sign outside the root: the rows are scaled by sqrt|w_k|, every
    // register accumulates its own residual's terms (acc4: register r of lane (c, q) is residual 4r + q throughout), and the signs
    // join once, behind the sweep (the final step, whose terminal weights have signs of their own, is added with them directly).
    constexpr bool SQW = KP_FWD_SQW && RXC && RU0 && UNI;
    double s2run[4] = {0.0, 0.0, 0.0, 0.0}, s2term[4] = {0.0, 0.0, 0.0, 0.0}, sgrun[4] = {1.0, 1.0, 1.0, 1.0}, sgterm[4] = {1.0, 1.0, 1.0, 1.0};
    double acc4[4] = {0.0, 0.0, 0.0, 0.0};
    (void)s2run; (void)s2term; (void)RxTt; (void)sgrun; (void)sgterm; (void)acc4;
    if constexpr (RXC) {
        __amdgpu_buffer_rsrc_t rRxc = frsrc(F.rx_const, nr * n * 8);
        RxTc.x = fbld(rRxc, oRxT[0]); RxTc.y = fbld(rRxc, oRxT[1]); RxTc.z = fbld(rRxc, oRxT[2]); RxTc.w = fbld(rRxc, oRxT[3]);
        if constexpr (SQW) {
            // lane (c, q) of the operand tile holds r_x[k = c][p = 4r + q]: column c is residual c
            const double swr = (cs < nr) ? __builtin_sqrt(fabs(F.w_run[cs])) : 0.0, swt = (cs < nr) ? __builtin_sqrt(fabs(F.w_term[cs])) : 0.0;
            RxTt = RxTc * swt;
            RxTc = RxTc * swr;
#pragma unroll
            for (int r = 0; r < 4; r++) {           // the scoring side: residual k = 4r + q in register r
                const int kr = sig(4 * r + q);
                const double wr_ = (kr < nr) ? F.w_run[kr] : 0.0, wt_ = (kr < nr) ? F.w_term[kr] : 0.0;
                s2run[r] = 2.0 * __builtin_sqrt(fabs(wr_)); sgrun[r] = wr_ < 0.0 ? -1.0 : 1.0;
                s2term[r] = 2.0 * __builtin_sqrt(fabs(wt_)); sgterm[r] = wt_ < 0.0 ? -1.0 : 1.0;
            }
        }
    }
    (void)RxTc;
#pragma unroll
    for (int k = 0; k < NS; k++) load_tiles(k < T ? k : T - 1, S[k]);

    // One wavefront per workgroup: LDS accesses of the wave execute in order, so the write -> transposed read
    // pairs below need no barrier.  The Y operands of step t+1 are produced during step t (off the Z chain).
    auto stage_cols = [&](int t) {            // a4 in the column layout -> LDS
        const double dt = (double)(t - s);
        shA[(q) * 17 + c] = lerp_nc(sv[0], dt, av[0]);      shA[(4 + q) * 17 + c] = lerp_nc(sv[1], dt, av[1]);
        shA[(8 + q) * 17 + c] = lerp_nc(sv[2], dt, av[2]);  shA[(12 + q) * 17 + c] = lerp_nc(sv[3], dt, av[3]);
        shB[(q) * 17 + c] = lerp_nc(sv[4], dt, av[4]);      shB[(4 + q) * 17 + c] = lerp_nc(sv[5], dt, av[5]);
        shB[(8 + q) * 17 + c] = lerp_nc(sv[6], dt, av[6]);  shB[(12 + q) * 17 + c] = lerp_nc(sv[7], dt, av[7]);
    };
    auto fetch_Y = [&](d4 &Ya, d4 &Yb) {      // Ya(p, o) = A(o, p);  Yb(p, o) = B(o, p)
        Ya.x = shA[c * 17 + q];      Ya.y = shA[c * 17 + 4 + q];
        Ya.z = shA[c * 17 + 8 + q];  Ya.w = shA[c * 17 + 12 + q];
        Yb.x = shB[c * 17 + q];                Yb.y = NCU > 1 ? shB[c * 17 + 4 + q] : 0.0;
        Yb.z = NCU > 2 ? shB[c * 17 + 8 + q] : 0.0; Yb.w = NCU > 3 ? shB[c * 17 + 12 + q] : 0.0;
    };
    auto advance = [&](int t) {
        if constexpr (FSLP) {
            // A lane that reaches the end key-point of its segment takes the (value, slope) pairs requested a crossing ago -- register
            // moves -- and requests the pairs of the segment after.  The REQUESTS are issued by ALL lanes, behind a wave-uniform
            // branch ("some lane crosses"): a lane that does not cross re-requests the entry it already holds.  With the loads inside
            // the per-lane branch their results met the old values at the join of the branch -- copies of freshly loaded registers,
            // i.e. `s_waitcnt vmcnt(0)` on EVERY step: the wave drained its four tile sets in flight once per step and ran at one
            // trip to HBM per step (2.7 ms where the uniform form takes 1.8).  Now nothing waits for these loads before the
            // lane's next crossing.
            const bool cr = t >= e;
            if (__builtin_amdgcn_ballot_w64(cr) != 0) {
                if (cr) {
#pragma unroll
                    for (int i = 0; i < 8; i++) { sv[i] = ev[i]; av[i] = ea[i]; }
                    s = e; e = nb; idx++;
                }
                const int en = (idx + 1 < khi) ? idx + 1 - E0 : -1;
                load_col2(rS, co, en, NE, 2 * strideB, ev, ea);
                nb = (idx + 2 < khi) ? F.kp_times[idx + 2] : BIGT;
            }
            return;
        }
        if (pend) {                           // slope of the segment entered one step ago (its end column has landed)
            const double den = (double)(e - s);
            const double rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < 8; i++) av[i] = (e != BIGT) ? fdiv(ev[i] - sv[i], den, rinv) : 0.0;
            pend = false;
        }
        if (t >= e) {                         // per lane: reached the end key-point of the segment
#pragma unroll
            for (int i = 0; i < 8; i++) { sv[i] = ev[i]; av[i] = 0.0; }
            s = e; e = nb; idx++;
            load_col(rT, co, (idx + 1 < khi) ? idx + 1 - E0 : -1, NE, strideB, ev);
            nb = (idx + 2 < khi) ? F.kp_times[idx + 2] : BIGT;
            pend = true;
        }
    };
    // Y operands: step t uses (Ya, Yb); during step t the tiles of step t+1 are fetched from LDS (staged during
    // step t-1) and those of step t+2 are staged.
    d4 Ya, Yb;
    auto lerp_Y = [&](int t) {                // UNI: the operands of step t straight from the tracker
        const double dt = (double)(t - s);
        Ya.x = lerp_nc(sv[0], dt, av[0]); Ya.y = lerp_nc(sv[1], dt, av[1]); Ya.z = lerp_nc(sv[2], dt, av[2]); Ya.w = lerp_nc(sv[3], dt, av[3]);
        Yb.x = lerp_nc(sv[4], dt, av[4]); Yb.y = NCU > 1 ? lerp_nc(sv[5], dt, av[5]) : 0.0;
        Yb.z = NCU > 2 ? lerp_nc(sv[6], dt, av[6]) : 0.0; Yb.w = NCU > 3 ? lerp_nc(sv[7], dt, av[7]) : 0.0;
    };
    if (UNI) lerp_Y(0);
    else {
        stage_cols(0);
        fetch_Y(Ya, Yb);
        if (T > 1) { advance(1); stage_cols(1); }
    }

    // one descriptor per array for the whole trajectory; the step (t+1, or t again behind the last one: never used) goes
    // into the loads' scalar offset
    const __amdgpu_buffer_rsrc_t rK = frsrc(Kin + (size_t)b * T * m * n, T * m * n * 8), rk = frsrc(kin + (size_t)b * T * m, T * m * 8),
                                 ru = frsrc(u_nom + (size_t)b * T * m, T * m * 8), rRx = frsrc(rxb, (T + 1) * nr * n * 8),
                                 rRu = frsrc(rub, (T + 1) * nr * m * 8), rR = frsrc(rb, (T + 1) * nr * 8);
    // UNI: the uniform tracker of the segment loop below (positions and times are wave-uniform scalars)
    int up = 0, uks = 0, uke = 0, ukn_v = 0;            // segment's start position, its start / end time, the time after that (as loaded)
    double ev2[8];                                       // the column after the next one (requested a segment ahead)
    double wterm[4] = {0.0, 0.0, 0.0, 0.0};
    (void)up; (void)uks; (void)uke; (void)ukn_v; (void)ev2; (void)wterm;
    // MODE 0: the general form (per-lane tracker, crossing inside the step); UNI: 1 a step below T-1 -- when it is the last one
    // of its segment, the crossing into the next segment sits behind a wave-uniform scalar branch where the general form advances
    // its per-lane tracker; 3 the final step t = T-1 (terminal weights, nothing left to request).
    // The tile requests stay OUTSIDE every branch and the loop has one body (a pair of steps): a request in flight across a
    // join of paths that each issued it ends in register copies at the join, and a copy waits for the load (a form with the
    // crossing step peeled into straight-line code had `s_waitcnt vmcnt(0)` plus 16 moves at the end of every segment).
    // PAR: the parity of t (which register set the step works on)
    auto step = [&](int t, auto mode_tag, auto par_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        Tiles &cur = S[decltype(par_tag)::value];
        const bool more = MODE == 3 ? false : MODE != 0 ? true : t + 1 < T;
        const int tn = MODE == 3 ? t : (t + NS < T ? t + NS : T - 1);       // (behind the last steps: a valid address, never used)
        const int sK = tn * m * n * 8, sk = tn * m * 8, sRx = tn * nr * n * 8, sRu = tn * nr * m * 8, sR = tn * nr * 8;
        (void)sRu;
        // Order of the step: every product is a dependent MFMA chain whose result is usable ~100 cycles after its last
        // issue, so independent work is placed behind each chain before its consumer:
        //   U chain | A dx chain | clamp (U ready) | B du | r_u du | r_x dx | a4 of step t+1 (VALU + LDS) | cost (Jx ready)
        d4 Yk = cur.YkK;                               // + k in row n (the lanes of the other rows hold 0): ONE load, not a tile of four
        if constexpr (rn == 3) Yk.w += cur.kk; else if constexpr (rn == 2) Yk.z += cur.kk; else if constexpr (rn == 1) Yk.y += cur.kk; else Yk.x += cur.kk;
        const d4 ub = cur.ub;
        // TRIM (round 5, the uniform form): the product starts from zero and u_nom joins on the VALU in the registers that hold
        // controls -- as the accumulator's initial value the tile had to be assembled first: seven moves per step -- and the clamp is
        // v_min / v_max: 1.74 -> 1.68 ms.  (The general form measured 3 % SLOWER with the same change and keeps the old code.)
        constexpr bool TRIM = KP_FWD_TRIM && UNI;
        d4 U = PS<NCZ>(Yk, Z, TRIM ? zero : ub);       // (u_nom +) K dx + alpha k   (:879)
        __builtin_amdgcn_sched_barrier(0);
        cur.YkK.x = fblds(rK, oK[0], sK); cur.YkK.y = fblds(rK, oK[1], sK); cur.YkK.z = fblds(rK, oK[2], sK); cur.YkK.w = fblds(rK, oK[3], sK);
        cur.kk = fblds(rk, okn, sk);
        __builtin_amdgcn_sched_barrier(0);
        d4 Zn = PS<NCZ>(Ya, Z, zero);                  // A dx (does not need the controls)
        __builtin_amdgcn_sched_barrier(0);
        d4 dU;
        {
            double u;
            // rows >= 4*NCU of U hold no control (exact zeros): only the first NCU registers are clamped (:883-889)
            dU = zero;
            if constexpr (TRIM) {
            u = vmax64(vmin64(U.x + ub.x, hi[0]), lo[0]); U.x = u; dU.x = u - ub.x;
            if constexpr (NCU > 1) { u = vmax64(vmin64(U.y + ub.y, hi[1]), lo[1]); U.y = u; dU.y = u - ub.y; }
            if constexpr (NCU > 2) { u = vmax64(vmin64(U.z + ub.z, hi[2]), lo[2]); U.z = u; dU.z = u - ub.z; }
            if constexpr (NCU > 3) { u = vmax64(vmin64(U.w + ub.w, hi[3]), lo[3]); U.w = u; dU.w = u - ub.w; }
            } else {
            u = U.x; if (u > hi[0]) u = hi[0]; if (u < lo[0]) u = lo[0]; U.x = u; dU.x = u - ub.x;
            if constexpr (NCU > 1) { u = U.y; if (u > hi[1]) u = hi[1]; if (u < lo[1]) u = lo[1]; U.y = u; dU.y = u - ub.y; }
            if constexpr (NCU > 2) { u = U.z; if (u > hi[2]) u = hi[2]; if (u < lo[2]) u = lo[2]; U.z = u; dU.z = u - ub.z; }
            if constexpr (NCU > 3) { u = U.w; if (u > hi[3]) u = hi[3]; if (u < lo[3]) u = lo[3]; U.w = u; dU.w = u - ub.w; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur.ub.x = fblds(ru, oub[0], sk); cur.ub.y = NCU > 1 ? fblds(ru, oub[1], sk) : 0.0;
        cur.ub.z = NCU > 2 ? fblds(ru, oub[2], sk) : 0.0; cur.ub.w = NCU > 3 ? fblds(ru, oub[3], sk) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        Zn = PS<NCU>(Yb, dU, Zn);                      // + B du: the next state
        d4 Ju = zero;                                  // RU0: r_u = 0, the control residual term vanishes
        if constexpr (!RU0) Ju = PS<NCU>(cur.RuT, dU, zero);
        const d4 Jx = PS<ncx>(RXC ? ((SQW && MODE == 3) ? RxTt : RxTc) : cur.RxT, Z, zero);     // (SQW: sqrt(w) Jx; terminal weights at the final step)
        Z = Zn;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!RXC) { cur.RxT.x = fblds(rRx, oRxT[0], sRx); cur.RxT.y = fblds(rRx, oRxT[1], sRx); cur.RxT.z = fblds(rRx, oRxT[2], sRx); cur.RxT.w = fblds(rRx, oRxT[3], sRx); }
        if constexpr (!RU0) {
            cur.RuT.x = fblds(rRu, oRuT[0], sRu); cur.RuT.y = NCU > 1 ? fblds(rRu, oRuT[1], sRu) : 0.0;
            cur.RuT.z = NCU > 2 ? fblds(rRu, oRuT[2], sRu) : 0.0; cur.RuT.w = NCU > 3 ? fblds(rRu, oRuT[3], sRu) : 0.0;
        }
        if (!UNI && more) fetch_Y(Ya, Yb);             // Y operands of step t+1 (staged during step t-1 ... see below)
        __builtin_amdgcn_sched_barrier(0);
        if (U_alpha && c < n_alpha) {
            double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
            const double uv[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) { const int row = 4 * r + q; if (row < m) Ua[row] = uv[r]; }
        }
        // a4 for step t+2 (column layout -> LDS): fills the latency of the products above
        if constexpr (MODE == 0) {
            if (UNI) { if (more) { advance(t + 1); lerp_Y(t + 1); } }
            else if (t + 2 < T) { advance(t + 2); stage_cols(t + 2); }
        } else if constexpr (MODE == 1) {
            if (t + 1 == uke) {
                // crossing into segment up + 1: its start column (ev) and end column (ev2) were requested one and two segments
                // ago; the column after those and its time are requested now, the time BEHIND the columns (loads return in
                // order: a wait that leaves "the youngest loads" in flight must not have to sit out the time's latency)
                const int kn = __builtin_amdgcn_readfirstlane(ukn_v);
                uks = uke; uke = kn; up++;
                const int gap = uke - uks;
                const double den = (double)(gap > 0 ? gap : 1), rinv = kp_rcp(den);
#pragma unroll
                for (int i = 0; i < 8; i++) { sv[i] = ev[i]; ev[i] = ev2[i]; av[i] = fdiv(ev[i] - sv[i], den, rinv); }
                if (c == n || c == n + 1) {                                // the identity rows of Ya: lanes c >= n walk no list
#pragma unroll
                    for (int r = 0; r < 4; r++) if (4 * r + q == c) { sv[r] = 1.0; ev[r] = 1.0; av[r] = 0.0; }
                }
                const int pn = up + 2 < KpU ? up + 2 : KpU - 1;
                load_col(rT, co, pn, NE, strideB, ev2);
                __builtin_amdgcn_sched_barrier(0);
                ukn_v = F.kp_times[E0 + pn];
                __builtin_amdgcn_sched_barrier(0);
            }
            const double dt = (double)(t + 1 - uks);                       // 0 behind a crossing: the next step IS the key-point
            Ya.x = lerp_nc(sv[0], dt, av[0]); Ya.y = lerp_nc(sv[1], dt, av[1]); Ya.z = lerp_nc(sv[2], dt, av[2]); Ya.w = lerp_nc(sv[3], dt, av[3]);
            Yb.x = lerp_nc(sv[4], dt, av[4]); Yb.y = NCU > 1 ? lerp_nc(sv[5], dt, av[5]) : 0.0;
            Yb.z = NCU > 2 ? lerp_nc(sv[6], dt, av[6]) : 0.0; Yb.w = NCU > 3 ? lerp_nc(sv[7], dt, av[7]) : 0.0;
        }
        // quadratic cost model on the residuals: sum_k w_k [Jx_k (2 r_k + Jx_k) + Ju_k (2 r_k + Ju_k)]
        if constexpr (MODE == 0) {
            if (t == T - 1) {                     // terminal weights at the last step (Optimiser.cpp:209-211)
#pragma unroll
                for (int r = 0; r < 4; r++) wcur[r] = (sig(4 * r + q) < nr) ? F.w_term[sig(4 * r + q)] : 0.0;   // (hoisting these loads out of the loop measured 0.9 ms slower)
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 4; r++) wcur[r] = wterm[r];
        }
        const d4 r2 = cur.rv + cur.rv;
        if constexpr (SQW) {
            if constexpr (MODE == 3) {         // the final step: terminal weights, their signs applied on the spot
                partial += sgterm[0] * (Jx.x * __builtin_fma(s2term[0], cur.rv.x, Jx.x)) + sgterm[1] * (Jx.y * __builtin_fma(s2term[1], cur.rv.y, Jx.y))
                         + sgterm[2] * (Jx.z * __builtin_fma(s2term[2], cur.rv.z, Jx.z)) + sgterm[3] * (Jx.w * __builtin_fma(s2term[3], cur.rv.w, Jx.w));
            } else {
                acc4[0] = __builtin_fma(Jx.x, __builtin_fma(s2run[0], cur.rv.x, Jx.x), acc4[0]);
                acc4[1] = __builtin_fma(Jx.y, __builtin_fma(s2run[1], cur.rv.y, Jx.y), acc4[1]);
                acc4[2] = __builtin_fma(Jx.z, __builtin_fma(s2run[2], cur.rv.z, Jx.z), acc4[2]);
                acc4[3] = __builtin_fma(Jx.w, __builtin_fma(s2run[3], cur.rv.w, Jx.w), acc4[3]);
            }
        } else if constexpr (RU0)                      // Ju = 0 exactly: its term adds a signed zero
            partial += wcur[0] * (Jx.x * (r2.x + Jx.x)) + wcur[1] * (Jx.y * (r2.y + Jx.y))
                     + wcur[2] * (Jx.z * (r2.z + Jx.z)) + wcur[3] * (Jx.w * (r2.w + Jx.w));
        else
            partial += wcur[0] * (Jx.x * (r2.x + Jx.x) + Ju.x * (r2.x + Ju.x))
                     + wcur[1] * (Jx.y * (r2.y + Jx.y) + Ju.y * (r2.y + Ju.y))
                     + wcur[2] * (Jx.z * (r2.z + Jx.z) + Ju.z * (r2.z + Ju.z))
                     + wcur[3] * (Jx.w * (r2.w + Jx.w) + Ju.w * (r2.w + Ju.w));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RV2) {
            double r0, r1, r2, r3;
            fbld2s(rR, oR2[0], sR, r0, r1); fbld2s(rR, oR2[1], sR, r2, r3);
            cur.rv.x = r0; cur.rv.y = r1; cur.rv.z = r2; cur.rv.w = r3;
        } else {
        cur.rv.x = fblds(rR, oR[0], sR); cur.rv.y = fblds(rR, oR[1], sR); cur.rv.z = fblds(rR, oR[2], sR); cur.rv.w = fblds(rR, oR[3], sR);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (UNI) {
        // Segments from the bottom: segment p = steps k_p .. k_p+1 - 1; the last key-point T-1 is a segment of its own (the
        // final step).  On entry of the loop: sv = column(k_0), ev = column(k_1), ev2 = column(k_2) in flight, av = slope of
        // segment 0, Y(0) = sv.  (The per-lane tracker above did the loads of sv and ev; it is not used any further.)
#pragma unroll
        for (int r = 0; r < 4; r++) wterm[r] = (sig(4 * r + q) < nr) ? F.w_term[sig(4 * r + q)] : 0.0;
        uks = __builtin_amdgcn_readfirstlane(F.kp_times[E0]);
        uke = __builtin_amdgcn_readfirstlane(F.kp_times[E0 + (KpU > 1 ? 1 : 0)]);
        {
            const int p2 = KpU > 2 ? 2 : KpU - 1;
            load_col(rT, co, p2, NE, strideB, ev2);
            ukn_v = F.kp_times[E0 + p2];
            const int gap = uke - uks;
            const double den = (double)(gap > 0 ? gap : 1), rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < 8; i++) av[i] = fdiv(ev[i] - sv[i], den, rinv);
            if (c == n || c == n + 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (4 * r + q == c) { ev[r] = 1.0; av[r] = 0.0; }
            }
        }
        int t = 0;
        for (; t + NS < T; t += NS) fwd_steps<0, NS, 1>(step, t);         // (the crossing inside a step moves up, uks, uke on)
        fwd_tail<0, NS, 1>(step, t, T);
    } else {
        int t = 0;
        for (; t + NS <= T; t += NS) fwd_steps<0, NS, 0>(step, t);
        fwd_tail<0, NS, 0>(step, t, T);
    }
    if constexpr (SQW) partial += sgrun[0] * acc4[0] + sgrun[1] * acc4[1] + sgrun[2] * acc4[2] + sgrun[3] * acc4[3];
    partial += __shfl_xor(partial, 16);
    partial += __shfl_xor(partial, 32);
    if (q == 0 && c < n_alpha) cost_pred[(size_t)b * n_alpha + c] = partial;
}

// The one-wave forward sweep comes in two forms, launched back to back: UNI for key-point sets in which every DoF of a
// trajectory has the same list (the device flag of k_kp_uniform says so), the general form otherwise.  Each looks at the
// flag first and leaves if the set is not its kind, so the host never has to know.
template <int NCZ, int NCU, bool RU0, bool UNI, bool RXC = false>
__global__ void __launch_bounds__(64)
k_forward_fused(RecLayout L, FusedArgs F, int T, int n_alpha, const double *__restrict__ Kin,
                const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha,
                const int *__restrict__ kp_uniform)
{
    if ((*kp_uniform != 0) != UNI) return;
    forward_fused_body<NCZ, NCU, RU0, UNI, RXC>(L, F, T, n_alpha, Kin, kin, u_nom, ctrl_lim, alphas, cost_pred, U_alpha);
}
template <int NCZ, int NCU, bool RU0, bool UNI, bool RXC = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_forward_fused_excl(RecLayout L, FusedArgs F, int T, int n_alpha, const double *__restrict__ Kin,
                     const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                     const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha,
                     const int *__restrict__ kp_uniform)
{
    if ((*kp_uniform != 0) != UNI) return;
    forward_fused_body<NCZ, NCU, RU0, UNI, RXC>(L, F, T, n_alpha, Kin, kin, u_nom, ctrl_lim, alphas, cost_pred, U_alpha);
}

// ---------------------------------------------------------------------------------------------------------
// Forward pass, STATE / COST wave pair per trajectory (small batches: every wave has a SIMD to itself).  Only the state
// recursion is serial in time: wave S keeps Z and runs  U = u_nom + K dx + alpha k | clamp | Z' = A dx + B du  (10 MFMAs a
// step) and publishes (Z_t, dU_t); wave C, one step behind, scores the candidates (Jx = r_x dx, Ju = r_u du, the weighted
// quadratic: 6 MFMAs + the FP64 VALU work) and, one step ahead, interpolates the A, B columns of the next step (a4) into
// the transposing LDS tiles S reads.  One s_barrier per step, two ring slots each way.
#define FSC_Y 544                                 // one Y slot: A then B, [16][17] each
#define FSC_YS 0
#define FSC_ZS (2 * FSC_Y)                        // two (Z, dU) slots, D layout
#define FSC_TOTAL (FSC_ZS + 2 * 512)

// RU0: nobody reads the control change of a step (no control residuals to score): its LDS image is not written
// UNI: every DoF of the trajectory has the same key-point list -- the state wave interpolates its own operands Ya, Yb in the
// operand layout, straight from the column store (the segment tracker of forward_fused_body's UNI form: positions and times are
// wave-uniform scalars, the crossing sits behind a scalar branch at the end of a segment's last step); nobody stages columns
// through LDS, and the group needs no third wave.
template <int NCZ, int NCU, bool RU0 = false, bool UNI = false>
__device__ __forceinline__ void forward_sc_state(double *sh, RecLayout L, int T, int n_alpha, const double *__restrict__ Kin,
               const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
               const double *__restrict__ alphas, double *__restrict__ U_alpha, const FusedArgs *Fp = nullptr)
{
    const int n = L.n, m = L.m;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    int oK[4], oub[4];
    double lo[NCU], hi[NCU], kmask[4];
    // k rides in row n of the gain operand: ONE load (the lanes of that row), added into the register that holds the row
    const int okn = (q == (n & 3) && c < m) ? 8 * c : OOBF;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        oK[r] = (row < n && c < m) ? 8 * (row * m + c) : OOBF;
        kmask[r] = (r == (n >> 2)) ? 1.0 : 0.0;
        oub[r] = (row < m) ? 8 * row : OOBF;
        if (r < NCU) {
            lo[r] = (row < m) ? ctrl_lim[2 * row] : -1.0e300;
            hi[r] = (row < m) ? ctrl_lim[2 * row + 1] : 1.0e300;
        }
    }
    const double my_alpha = (c < n_alpha) ? alphas[c] : 0.0;
    d4 Z;
    {
        double zr[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * r + q;
            zr[r] = (row == n) ? my_alpha : (row == n + 1) ? 1.0 : 0.0;
        }
        Z.x = zr[0]; Z.y = zr[1]; Z.z = zr[2]; Z.w = zr[3];
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    // KP_FSC_SETS tile sets (step t works on set t mod KP_FSC_SETS): each is re-requested for step t + KP_FSC_SETS right behind
    // its use in step t.  The time loop is a body of KP_FSC_SETS steps WITHOUT conditions (the steps below a full group are
    // peeled behind it): a request inside `if (t + 1 < T)` makes every later wait conservative
    struct STiles { d4 YkK, ub; double kk; };
    // one descriptor per array for the whole trajectory, the step in the loads' scalar offset (behind the last step: step T-1
    // again, never used) -- as in forward_fused_body
    const __amdgpu_buffer_rsrc_t rK = frsrc(Kin + (size_t)b * T * m * n, T * m * n * 8), rk = frsrc(kin + (size_t)b * T * m, T * m * 8),
                                 ru = frsrc(u_nom + (size_t)b * T * m, T * m * 8);
    auto request = [&](int t, STiles &s_) {
        const int tn = t < T ? t : T - 1;
        const int sK = tn * m * n * 8, sk = tn * m * 8;
        s_.YkK.x = fblds(rK, oK[0], sK); s_.YkK.y = fblds(rK, oK[1], sK); s_.YkK.z = fblds(rK, oK[2], sK); s_.YkK.w = fblds(rK, oK[3], sK);
        s_.kk = fblds(rk, okn, sk);
        s_.ub.x = fblds(ru, oub[0], sk); s_.ub.y = NCU > 1 ? fblds(ru, oub[1], sk) : 0.0;
        s_.ub.z = NCU > 2 ? fblds(ru, oub[2], sk) : 0.0; s_.ub.w = NCU > 3 ? fblds(ru, oub[3], sk) : 0.0;
    };
    // ---- UNI: the segment tracker (forward_fused_body) ----
    ColOffs co;
    double sv[8], ev[8], ev2[8], av[8];
    int up = 0, uks = 0, uke = 0, ukn_v = 0, E0 = 0, NE = 0, KpU = 1;
    __amdgpu_buffer_rsrc_t rT = rK;
    d4 Yua = zero, Yub = zero;                         // the operands of the NEXT step
    (void)co; (void)sv; (void)ev; (void)ev2; (void)av; (void)up; (void)uks; (void)uke; (void)ukn_v; (void)E0; (void)NE; (void)KpU; (void)rT;
    constexpr int strideBc = 0;
    (void)strideBc;
    const int strideB = 3 * n * 8;
    if constexpr (UNI) {
        const FusedArgs &F = *Fp;
        E0 = F.kp_offsets[(size_t)b * F.dof]; NE = F.kp_offsets[(size_t)(b + 1) * F.dof] - E0;
        KpU = F.kp_offsets[(size_t)b * F.dof + 1] - E0;
        rT = frsrc(F.kpc + (size_t)E0 * 3 * n, NE * strideB);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // the Y operands themselves: Ya(p = row, o = c) = A(o, p), Yb(p, o) = B(o, p) -- element c of column `row`, whose
            // DoF list starts (row mod dof) * KpU entries into the trajectory's slice
            const int row = 4 * r + q;
            const int d = row < F.dof ? row : row - F.dof;
            co.a[r] = (row < n && c < n) ? 8 * ((d * KpU * 3 + (row < F.dof ? 0 : 1)) * n + c) : BIGOFF;
            co.b[r] = (row < m && c < n) ? 8 * ((row * KpU * 3 + 2) * n + c) : BIGOFF;
        }
        load_col(rT, co, 0, NE, strideB, sv);
        load_col(rT, co, KpU > 1 ? 1 : 0, NE, strideB, ev);
        uks = __builtin_amdgcn_readfirstlane(F.kp_times[E0]);
        uke = __builtin_amdgcn_readfirstlane(F.kp_times[E0 + (KpU > 1 ? 1 : 0)]);
        const int p2 = KpU > 2 ? 2 : KpU - 1;
        load_col(rT, co, p2, NE, strideB, ev2);
        ukn_v = F.kp_times[E0 + p2];
        const int gap = uke - uks;
        const double den = (double)(gap > 0 ? gap : 1), rinv = kp_rcp(den);
#pragma unroll
        for (int i = 0; i < 8; i++) av[i] = fdiv(ev[i] - sv[i], den, rinv);
        // the identity rows of Ya (alpha and the homogeneous 1 carry over): lanes c >= n walk no list
        if (c == n || c == n + 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) if (4 * r + q == c) { sv[r] = 1.0; ev[r] = 1.0; av[r] = 0.0; }
        }
        Yua.x = sv[0]; Yua.y = sv[1]; Yua.z = sv[2]; Yua.w = sv[3];            // Y(0): the first key-point is step 0
        Yub.x = sv[4]; Yub.y = NCU > 1 ? sv[5] : 0.0; Yub.z = NCU > 2 ? sv[6] : 0.0; Yub.w = NCU > 3 ? sv[7] : 0.0;
    }
    // the operands of step t + 1, formed behind the products of step t (UNI)
    auto next_Y = [&](int t) __attribute__((always_inline)) {
        const FusedArgs &F = *Fp;
        if (t + 1 == uke) {
            // crossing into segment up + 1: its start column (ev) and end column (ev2) were requested one and two segments ago;
            // the column after those and its time are requested now, the time BEHIND the columns (loads return in order)
            const int kn = __builtin_amdgcn_readfirstlane(ukn_v);
            uks = uke; uke = kn; up++;
            const int gap = uke - uks;
            const double den = (double)(gap > 0 ? gap : 1), rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < 8; i++) { sv[i] = ev[i]; ev[i] = ev2[i]; av[i] = fdiv(ev[i] - sv[i], den, rinv); }
            if (c == n || c == n + 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (4 * r + q == c) { sv[r] = 1.0; ev[r] = 1.0; av[r] = 0.0; }
            }
            const int pn = up + 2 < KpU ? up + 2 : KpU - 1;
            load_col(rT, co, pn, NE, strideB, ev2);
            __builtin_amdgcn_sched_barrier(0);
            ukn_v = F.kp_times[E0 + pn];
            __builtin_amdgcn_sched_barrier(0);
        }
        const double dt = (double)(t + 1 - uks);                       // 0 behind a crossing: the next step IS the key-point
        Yua.x = lerp_nc(sv[0], dt, av[0]); Yua.y = lerp_nc(sv[1], dt, av[1]); Yua.z = lerp_nc(sv[2], dt, av[2]); Yua.w = lerp_nc(sv[3], dt, av[3]);
        Yub.x = lerp_nc(sv[4], dt, av[4]); Yub.y = NCU > 1 ? lerp_nc(sv[5], dt, av[5]) : 0.0;
        Yub.z = NCU > 2 ? lerp_nc(sv[6], dt, av[6]) : 0.0; Yub.w = NCU > 3 ? lerp_nc(sv[7], dt, av[7]) : 0.0;
    };
    auto step = [&](int t, STiles &s_) {
        const double *shA = sh + FSC_YS + (t & 1) * FSC_Y, *shB = shA + 272;
        d4 Ya, Yb;                                     // Ya(p, o) = A(o, p);  Yb(p, o) = B(o, p)
        if constexpr (UNI) { Ya = Yua; Yb = Yub; }
        else {
        Ya.x = shA[c * 17 + q];      Ya.y = shA[c * 17 + 4 + q];
        Ya.z = shA[c * 17 + 8 + q];  Ya.w = shA[c * 17 + 12 + q];
        Yb.x = shB[c * 17 + q];                Yb.y = NCU > 1 ? shB[c * 17 + 4 + q] : 0.0;
        Yb.z = NCU > 2 ? shB[c * 17 + 8 + q] : 0.0; Yb.w = NCU > 3 ? shB[c * 17 + 12 + q] : 0.0;
        }
        d4 Yk = s_.YkK;
        {
            const double kk = s_.kk;                   // (a product with the 0 / 1 mask: no branch, no select on the register index)
            Yk.x = __builtin_fma(kmask[0], kk, Yk.x); Yk.y = __builtin_fma(kmask[1], kk, Yk.y);
            Yk.z = __builtin_fma(kmask[2], kk, Yk.z); Yk.w = __builtin_fma(kmask[3], kk, Yk.w);
        }
        const d4 ub = s_.ub;
        d4 U = PS<NCZ>(Yk, Z, ub);                     // u_nom + K dx + alpha k   (:879)
        __builtin_amdgcn_sched_barrier(0);
        request(t + KP_FSC_SETS, s_);
        __builtin_amdgcn_sched_barrier(0);
        d4 Zn = PS<NCZ>(Ya, Z, zero);                  // A dx
        d4 dU = zero;
        {
            double u;                                  // clamp (:883-889)
            u = U.x; if (u > hi[0]) u = hi[0]; if (u < lo[0]) u = lo[0]; U.x = u; dU.x = u - ub.x;
            if (NCU > 1) { u = U.y; if (u > hi[1]) u = hi[1]; if (u < lo[1]) u = lo[1]; U.y = u; dU.y = u - ub.y; }
            if (NCU > 2) { u = U.z; if (u > hi[2]) u = hi[2]; if (u < lo[2]) u = lo[2]; U.z = u; dU.z = u - ub.z; }
            if (NCU > 3) { u = U.w; if (u > hi[3]) u = hi[3]; if (u < lo[3]) u = lo[3]; U.w = u; dU.w = u - ub.w; }
        }
        double *zs = sh + FSC_ZS + (t & 1) * 512;
        lds_store4(zs, lane, Z);                       // the state this step started from, and its control change
        if constexpr (!RU0) lds_store4(zs + 256, lane, dU);
        Z = PS<NCU>(Yb, dU, Zn);                       // + B du: the next state
        if (U_alpha && c < n_alpha) {
            double *Ua = U_alpha + (((size_t)b * n_alpha + c) * T + t) * m;
            const double uv[4] = {U.x, U.y, U.z, U.w};
#pragma unroll
            for (int r = 0; r < NCU; r++) { const int row = 4 * r + q; if (row < m) Ua[row] = uv[r]; }
        }
        if constexpr (UNI) next_Y(t);                  // a4 of step t+1 under the products above
        __syncthreads();
    };
    STiles S[KP_FSC_SETS];
#pragma unroll
    for (int k = 0; k < KP_FSC_SETS; k++) request(k, S[k]);
    __syncthreads();                                   // Y(0) is staged
    int t = 0;
    for (; t + KP_FSC_SETS <= T; t += KP_FSC_SETS) {
#pragma unroll
        for (int k = 0; k < KP_FSC_SETS; k++) step(t + k, S[k]);
    }
#pragma unroll
    for (int k = 0; k < KP_FSC_SETS - 1; k++) if (t + k < T) step(t + k, S[k]);
}

// ROLE 0: score and stage (second wave of the pair); 1: score only; 2: stage the A, B columns only (third wave of the triple)
// RU0: r_u = 0 (no r_u tiles, no Ju product); RXC (with RU0): ONE constant r_x, its transposed tile in registers (forward_fused_body)
template <int NCZ, int NCU, int ROLE = 0, bool RU0 = false, bool RXC = false>
__device__ __forceinline__ void forward_sc_cost(double *sh, RecLayout L, FusedArgs F, int T, int n_alpha, double *__restrict__ cost_pred)
{
    const int n = L.n, m = L.m;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    const int nr = F.nr;
    const int strideB = 3 * L.n * 8;                               // bytes of one key-point entry of kpc: three columns
    const int ncx = (n + 3) >> 2;
    ColOffs co;
    col_offsets(co, n, m, F.dof, c, q);
    int oRxT[4], oRuT[4], oR[4];
    double wcur[4], wterm[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = 4 * r + q;
        oRxT[r] = (row < n && c < nr) ? 8 * (c * n + row) : OOBF;     // RxT(p=row, k=c) = r_x[k][p]
        oRuT[r] = (row < m && c < nr) ? 8 * (c * m + row) : OOBF;     // RuT(p=row, k=c) = r_u[k][p]
        oR[r] = (row < nr) ? 8 * row : OOBF;
        wcur[r] = (row < nr) ? F.w_run[row] : 0.0;
        wterm[r] = (row < nr) ? F.w_term[row] : 0.0;   // in registers from the start: a load inside the time loop makes the
                                                       // compiler drain every outstanding prefetch (vmcnt(0)) each step here
    }
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    double partial = 0.0;
    const int E0 = F.kp_offsets[(size_t)b * F.dof], NE = F.kp_offsets[(size_t)(b + 1) * F.dof] - E0;      // this trajectory's key-point entries
    __amdgpu_buffer_rsrc_t rT = frsrc(F.kpc + (size_t)E0 * 3 * L.n, NE * strideB);
    const double *rb = F.r + (size_t)b * (T + 1) * nr;
    const double *rxb = F.r_x + (size_t)b * (T + 1) * nr * n;
    const double *rub = F.r_u + (size_t)b * (T + 1) * nr * m;
    struct CTiles { d4 RxT, RuT, rv; };
    const __amdgpu_buffer_rsrc_t rNone = frsrc(rb, 0);
    d4 RxTc = {0.0, 0.0, 0.0, 0.0};
    if constexpr (RXC) {
        __amdgpu_buffer_rsrc_t rRxc = frsrc(F.rx_const, nr * n * 8);
        RxTc.x = fbld(rRxc, oRxT[0]); RxTc.y = fbld(rRxc, oRxT[1]); RxTc.z = fbld(rRxc, oRxT[2]); RxTc.w = fbld(rRxc, oRxT[3]);
    }
    (void)RxTc;
    auto request = [&](int t, CTiles &s_) {
        const bool ok = t < T;
        __amdgpu_buffer_rsrc_t rRx = ok ? frsrc(rxb + (size_t)t * nr * n, nr * n * 8) : rNone;
        __amdgpu_buffer_rsrc_t rRu = ok ? frsrc(rub + (size_t)t * nr * m, nr * m * 8) : rNone;
        __amdgpu_buffer_rsrc_t rR = ok ? frsrc(rb + (size_t)t * nr, nr * 8) : rNone;
        if constexpr (!RXC) { s_.RxT.x = fbld(rRx, oRxT[0]); s_.RxT.y = fbld(rRx, oRxT[1]); s_.RxT.z = fbld(rRx, oRxT[2]); s_.RxT.w = fbld(rRx, oRxT[3]); }
        if constexpr (!RU0) {
            s_.RuT.x = fbld(rRu, oRuT[0]); s_.RuT.y = NCU > 1 ? fbld(rRu, oRuT[1]) : 0.0;
            s_.RuT.z = NCU > 2 ? fbld(rRu, oRuT[2]) : 0.0; s_.RuT.w = NCU > 3 ? fbld(rRu, oRuT[3]) : 0.0;
        }
        s_.rv.x = fbld(rR, oR[0]); s_.rv.y = fbld(rR, oR[1]); s_.rv.z = fbld(rR, oR[2]); s_.rv.w = fbld(rR, oR[3]);
    };
    // ---- column tracker, walking UP in time (as in forward_fused_body) ----
    const int kd = (c < F.dof) ? c : c - F.dof;
    const bool has = c < n;
    const int klo = has ? F.kp_offsets[(size_t)b * F.dof + kd] : 0;
    const int khi = has ? F.kp_offsets[(size_t)b * F.dof + kd + 1] : 0;
    int idx = klo;
    int s = has ? F.kp_times[idx] : 0;
    int e = (has && idx + 1 < khi) ? F.kp_times[idx + 1] : BIGT;
    int nb = (has && idx + 2 < khi) ? F.kp_times[idx + 2] : BIGT;
    double sv[8], ev[8], av[8];
    if (ROLE != 1) {
        load_col(rT, co, has ? idx - E0 : -1, NE, strideB, sv);
        load_col(rT, co, (has && idx + 1 < khi) ? idx + 1 - E0 : -1, NE, strideB, ev);
    }
    bool pend = true;
#pragma unroll
    for (int i = 0; i < 8; i++) av[i] = 0.0;
#pragma unroll
    for (int r = 0; r < 4; r++) if ((c == n || c == n + 1) && 4 * r + q == c) sv[r] = 1.0;
    auto stage_cols = [&](int t) {
        double *shA = sh + FSC_YS + (t & 1) * FSC_Y, *shB = shA + 272;
        const double dt = (double)(t - s);
        shA[(q) * 17 + c] = lerp_nc(sv[0], dt, av[0]);      shA[(4 + q) * 17 + c] = lerp_nc(sv[1], dt, av[1]);
        shA[(8 + q) * 17 + c] = lerp_nc(sv[2], dt, av[2]);  shA[(12 + q) * 17 + c] = lerp_nc(sv[3], dt, av[3]);
        shB[(q) * 17 + c] = lerp_nc(sv[4], dt, av[4]);      shB[(4 + q) * 17 + c] = lerp_nc(sv[5], dt, av[5]);
        shB[(8 + q) * 17 + c] = lerp_nc(sv[6], dt, av[6]);  shB[(12 + q) * 17 + c] = lerp_nc(sv[7], dt, av[7]);
    };
    auto advance = [&](int t) {
        if (pend) {
            const double den = (double)(e - s);
            const double rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < 8; i++) av[i] = (e != BIGT) ? fdiv(ev[i] - sv[i], den, rinv) : 0.0;
            pend = false;
        }
        if (t >= e) {
#pragma unroll
            for (int i = 0; i < 8; i++) { sv[i] = ev[i]; av[i] = 0.0; }
            s = e; e = nb; idx++;
            load_col(rT, co, (idx + 1 < khi) ? idx + 1 - E0 : -1, NE, strideB, ev);
            nb = (idx + 2 < khi) ? F.kp_times[idx + 2] : BIGT;
            pend = true;
        }
    };
    // iteration t: score step t-1 (tiles re-requested for step t+1 right behind their use), then stage the A, B columns of
    // step t+1 for wave S
    auto iter = [&](int t, CTiles &s_) {               // s_ holds the tiles of step t-1
        if (ROLE != 2 && t >= 1) {
            const int tt = t - 1;
            const double *zs = sh + FSC_ZS + (tt & 1) * 512;
            const d4 Zt = lds_tile4(zs, lane);
            d4 Ju = zero;                              // RU0: the control residual term vanishes (and nobody reads the dU slot)
            if constexpr (!RU0) { const d4 dU = lds_tile4(zs + 256, lane); Ju = PS<NCU>(s_.RuT, dU, zero); }
            const d4 Jx = PR(RXC ? RxTc : s_.RxT, Zt, zero, ncx);
            const d4 r2 = s_.rv + s_.rv;
            __builtin_amdgcn_sched_barrier(0);
            request(tt + KP_FSC_SETS, s_);
            __builtin_amdgcn_sched_barrier(0);
            if (tt == T - 1) {                // terminal weights at the last step (Optimiser.cpp:209-211)
#pragma unroll
                for (int r = 0; r < 4; r++) wcur[r] = wterm[r];
            }
            // sum_k w_k [Jx_k (2 r_k + Jx_k) + Ju_k (2 r_k + Ju_k)]
            if constexpr (RU0)                         // (Ju = 0 exactly: its term adds a signed zero, as in forward_fused_body)
                partial += wcur[0] * (Jx.x * (r2.x + Jx.x)) + wcur[1] * (Jx.y * (r2.y + Jx.y))
                         + wcur[2] * (Jx.z * (r2.z + Jx.z)) + wcur[3] * (Jx.w * (r2.w + Jx.w));
            else
            partial += wcur[0] * (Jx.x * (r2.x + Jx.x) + Ju.x * (r2.x + Ju.x))
                     + wcur[1] * (Jx.y * (r2.y + Jx.y) + Ju.y * (r2.y + Ju.y))
                     + wcur[2] * (Jx.z * (r2.z + Jx.z) + Ju.z * (r2.z + Ju.z))
                     + wcur[3] * (Jx.w * (r2.w + Jx.w) + Ju.w * (r2.w + Ju.w));
        }
        if (ROLE != 1 && t + 1 < T) { advance(t + 1); stage_cols(t + 1); }
        if (t < T) __syncthreads();
    };
    CTiles S[KP_FSC_SETS];                             // set k: the steps congruent k mod KP_FSC_SETS
    if (ROLE != 2) {
#pragma unroll
        for (int k = 0; k < KP_FSC_SETS; k++) request(k, S[k]);
    }
    if (ROLE != 1) stage_cols(0);
    __syncthreads();
    iter(0, S[KP_FSC_SETS - 1]);                       // (scores nothing: stages the columns of step 1)
    int t = 1;
    for (; t + KP_FSC_SETS - 1 <= T; t += KP_FSC_SETS) {   // iteration t scores step t-1, whose tiles are in set (t-1) mod KP_FSC_SETS
#pragma unroll
        for (int k = 0; k < KP_FSC_SETS; k++) iter(t + k, S[k]);
    }
#pragma unroll
    for (int k = 0; k < KP_FSC_SETS - 1; k++) if (t + k <= T) iter(t + k, S[k]);
    partial += __shfl_xor(partial, 16);
    partial += __shfl_xor(partial, 32);
    if (ROLE != 2 && q == 0 && c < n_alpha) cost_pred[(size_t)b * n_alpha + c] = partial;
}

template <int NCZ, int NCU, bool RU0 = false, bool RXC = false>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_forward_fused_sc(RecLayout L, FusedArgs F, int T, int n_alpha, const double *__restrict__ Kin,
                   const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                   const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha,
                   const int *__restrict__ kp_uniform, int only_ragged)
{
    __shared__ __attribute__((aligned(16))) double sh[FSC_TOTAL];
    if (only_ragged && *kp_uniform != 0) return;       // (launched behind k_forward_fused_scu, which has run the uniform set)
    const bool state = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    if (state) forward_sc_state<NCZ, NCU, RU0>(sh, L, T, n_alpha, Kin, kin, u_nom, ctrl_lim, alphas, U_alpha);
    else       forward_sc_cost<NCZ, NCU, 0, RU0, RXC>(sh, L, F, T, n_alpha, cost_pred);
}

// UNIFORM key-point sets: state wave (with its own a4) + scoring wave -- no staging wave, whatever the batch up to #SIMDs / 2.
// Leaves at once if the device flag says the set is not uniform (the general forms are launched behind it and look at the same flag).
template <int NCZ, int NCU, bool RU0 = false, bool RXC = false>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_forward_fused_scu(RecLayout L, FusedArgs F, int T, int n_alpha, const double *__restrict__ Kin,
                    const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                    const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha,
                    const int *__restrict__ kp_uniform)
{
    __shared__ __attribute__((aligned(16))) double sh[FSC_TOTAL];
    if (*kp_uniform == 0) return;
    const bool state = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    if (state) forward_sc_state<NCZ, NCU, RU0, true>(sh, L, T, n_alpha, Kin, kin, u_nom, ctrl_lim, alphas, U_alpha, &F);
    else       forward_sc_cost<NCZ, NCU, 1, RU0, RXC>(sh, L, F, T, n_alpha, cost_pred);
}

// state + cost + staging waves (4 x batch <= #SIMDs: one 3-wave workgroup per CU): the cost wave of the pair is the longer
// one (1 360 vs 1 130 cycles per step); its a4 half goes to a third wave
template <int NCZ, int NCU, bool RU0 = false, bool RXC = false>
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_forward_fused_sc3(RecLayout L, FusedArgs F, int T, int n_alpha, const double *__restrict__ Kin,
                    const double *__restrict__ kin, const double *__restrict__ u_nom, const double *__restrict__ ctrl_lim,
                    const double *__restrict__ alphas, double *__restrict__ cost_pred, double *__restrict__ U_alpha,
                    const int *__restrict__ kp_uniform, int only_ragged)
{
    __shared__ __attribute__((aligned(16))) double sh[FSC_TOTAL];
    if (only_ragged && *kp_uniform != 0) return;       // (launched behind k_forward_fused_scu, which has run the uniform set)
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (role == 0) forward_sc_state<NCZ, NCU, RU0>(sh, L, T, n_alpha, Kin, kin, u_nom, ctrl_lim, alphas, U_alpha);
    else if (role == 1) forward_sc_cost<NCZ, NCU, 1, RU0, RXC>(sh, L, F, T, n_alpha, cost_pred);
    else forward_sc_cost<NCZ, NCU, 2>(sh, L, F, T, n_alpha, cost_pred);
}

#if KP_PART(1)
bool fused_supported(int n, int m, int nr, int dof, int T, int stride, int n_alpha)
{
    (void)stride;
    // one trajectory's slice of the key-point column store (at most T entries per DoF, 3n doubles each) behind one descriptor
    return kp_t1_shape(n, m) && nr >= 1 && nr <= 16 && m <= dof && n_alpha <= 16 && (long long)T * dof * (6 * n + 2) * 8 < (long long)BIGOFF;
}

#endif
static FusedArgs fused_args(const Ctx *c)
{
    FusedArgs F = {c->kp_offsets, c->kp_times, c->r, c->r_x, c->r_u, c->w_run, c->w_term, c->d.dof, c->d.nr,
                   c->kpc, c->fdk_dev, 2 * c->eps, 1.0 / (2 * c->eps), c->rx_const,
                   c->kps ? c->kps : c->kpc};      // (no slope store: lists known uniform, no general form runs -- never a null descriptor)
    return F;
}

#if KP_PART(1)
// Which wave organisation launch_backward_fused will pick: 1 one wave per trajectory, 5 the consumer / helper pair (both have RAW
// instantiations).  While a trajectory can have two SIMDs (2 x batch <= #SIMDs) the pair runs it: the consumer's chain is
// Tu | Quu | refresh | gains | V', the helper forms the side products Tz, Quz, Qzz, then a4 + a6 of the next step, and differences
// the payload of uniform sets.  Beyond that two waves take turns on a SIMD and one wave per trajectory wins.
// KPILQR_FUSED_WAVES = 1 | 5 forces a form (diagnostic, include/kpilqr.h).  Rounds 2-4 also had a control / state split (2), a
// producer / consumer pair (3) and a consumer / side / producer triple (4): all measured slower than the helper pair at every
// batch size (profiles/r04_helper_pair.txt) and were removed in round 5 (DESIGN_HISTORY.md keeps their numbers).
int backward_fused_form(const Ctx *c)
{
    const int f = c->tune.fused_bwd_waves;
    return (f == 1 || f == 5) ? f : (2 * c->d.batch <= c->n_simd ? 5 : 1);
}

// The wave organisation launch_forward_fused will pick: 1 one wave per trajectory, 2 state / cost + staging pair, 3 the state /
// cost / staging triple, 4 (the default while 2 x batch <= #SIMDs) the state / cost pair for UNIFORM key-point sets -- the state
// wave interpolates its own operands, no staging wave -- with the triple (4 x batch <= #SIMDs) or the one-wave general form
// behind it for per-DoF lists; the device flag decides which of the two runs.  Late round 4, with r_u = 0 / constant-r_x
// instantiations of the scoring wave and a state wave that issues 7 loads per step instead of 10 (profiles/r04_forward_forms.txt):
// 1.31 / 1.43 / 1.47 ms at 1 / 64 / 256 trajectories (one wave per trajectory: 1.72).
int forward_fused_form(const Ctx *c)
{
    if (c->tune.fused_fwd_waves) return c->tune.fused_fwd_waves;
    return 2 * c->d.batch <= c->n_simd ? 4 : 1;
}

// raw: difference the key-point ordered payload inside the sweep (one-wave form only; the caller checks backward_fused_form)
hipError_t launch_backward_fused(Ctx *c, int pd_stride, bool raw)
{
    const int n = c->n, m = c->d.m;
    dim3 grid(c->d.batch), block(64);
    const bool excl = c->d.batch <= c->n_simd;
    const FusedArgs F = fused_args(c);
    // Wave organisation of the backward sweep (backward_fused_form above).  While every wave of the consumer / helper pair can have
    // a SIMD to itself (2 x batch <= #SIMDs) the pair is the fastest form; beyond that two waves take turns on a SIMD and one wave
    // per trajectory wins (DESIGN.md sections 4.0, 4.4).  KPILQR_FUSED_WAVES forces a form: 1 = one wave, 2 = control/state split,
    // 3 = producer/consumer pair, 4 = the consumer / side / producer triple, 5 = the consumer / helper pair.
    const int form = backward_fused_form(c);
    if (raw && form != 1 && form != 5) return hipErrorInvalidValue;
    c->last_bwd_form = form; c->last_bwd_raw = raw; c->last_bwd_ru0 = form == 1 && c->ru_zero;       // kpilqr_last_launch
    const bool rxc = form == 1 && c->ru_zero && c->rx_const_on;     // (the caller has materialised r_x for every other form)
    c->last_bwd_rxc = rxc;
    c->last_bwd_slopes = form == 1 && c->kps != nullptr && !(raw && c->tune.fused_uni == 0);      // (what the general form walks, if it runs)
    if (form != 1) return launch_backward_fused_waves(c, pd_stride, raw, form);
#define LAUNCH5(NN, MM, RU, RW, UN, RX)                                                                       \
    do {                                                                                                     \
        if (excl)                                                                                            \
            hipLaunchKernelGGL((k_backward_fused_excl<NN, MM, RU, RW, UN, RX>), grid, block, 0, c->stream, c->L, F, c->d.T,  \
                               c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status, c->kp_uniform);      \
        else                                                                                                 \
            hipLaunchKernelGGL((k_backward_fused<NN, MM, RU, RW, UN, RX>), grid, block, 0, c->stream, c->L, F, c->d.T,   \
                               c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status, c->kp_uniform);      \
    } while (0)
// rxc: constant residual Jacobians (with r_u = 0 only: RU instantiations)
#define LAUNCH4(NN, MM, RU, RW, UN) do { if (RU && rxc) LAUNCH5(NN, MM, RU, RW, UN, RU); else LAUNCH5(NN, MM, RU, RW, UN, false); } while (0)
// both forms, back to back: the one whose kind of key-point set is not resident leaves at once.  raw: only UNIFORM sets are
// differenced inside the sweep; for per-DoF lists a lane's crossing is a divergent branch that every lane of the wave pays for
// (17 loads, 8 stores on most steps), and the streaming kernel + the plain general sweep are faster (iterative-error lists with
// a key-point on 99 % of the steps: 8.7 + 2.2 against 11.9 ms) -- k_fd_kp_difference is launched in between and looks at the
// same device flag.  (KPILQR_FUSED_UNI=0, diagnostic: the general raw form for every set.)
#define LAUNCH3(NN, MM, RU, RW) do { if (c->tune.fused_uni != 0) LAUNCH4(NN, MM, RU, RW, true); LAUNCH4(NN, MM, RU, RW, false); } while (0)
#define LAUNCH2(NN, MM, RU)                                                                                             \
    do {                                                                                                                \
        if (raw && c->tune.fused_uni != 0) {                                                                            \
            LAUNCH4(NN, MM, RU, true, true);                                                                            \
            hipError_t e_ = launch_fd_kp_difference(c, true);     /* (per-DoF lists: columns and their slopes) */       \
            if (e_ != hipSuccess) return e_;                                                                            \
            LAUNCH4(NN, MM, RU, false, false);                                                                          \
        } else if (raw) LAUNCH3(NN, MM, RU, true);                                                                      \
        else LAUNCH3(NN, MM, RU, false);                                                                                \
    } while (0)
#define LAUNCH(NN, MM) do { if (c->ru_zero) LAUNCH2(NN, MM, true); else LAUNCH2(NN, MM, false); } while (0)
#define KP_X(NN, MM) if (n == NN && m == MM) { LAUNCH(NN, MM); return hipGetLastError(); }
    KP_T1_SHAPES(KP_X)
#undef KP_X
#undef LAUNCH
#undef LAUNCH2
#undef LAUNCH3
#undef LAUNCH4
#undef LAUNCH5
    return hipErrorInvalidValue;
}
#endif

#if KP_PART(2)
// form 5 of the backward sweep, the consumer / helper pair; launch_backward_fused has filled in kpilqr_last_launch's fields
hipError_t launch_backward_fused_waves(Ctx *c, int pd_stride, bool raw, int form)
{
    const int n = c->n, m = c->d.m;
    dim3 grid(c->d.batch);
    const FusedArgs F = fused_args(c);
    const int role_shift = c->tune.role_shift;
    dim3 block2(128);
    // form 5, consumer / helper pair: the raw launch sequence of the pair (the helper differences the payload of uniform sets)
    if (form == 5) {
        const bool hru0 = c->ru_zero, hrxc = c->ru_zero && c->rx_const_on;
        c->last_bwd_ru0 = hru0; c->last_bwd_rxc = hrxc;
        // per-DoF lists: the helper walks the slope store (made by k_fd_kp_difference / k_kp_slopes for such sets only), uniform
        // sets the column store with its dividing tracker: two launches, the device flag decides
        const bool hslp = c->kps != nullptr;
        c->last_bwd_slopes = hslp;
#define LAUNCHPH2(NN, MM, RW, RU, RX, SL, GUARD) hipLaunchKernelGGL((k_backward_fusedph<NN, MM, RW, RU, RX, SL>), grid, block2, 0, c->stream, c->L, F, c->d.T, role_shift, c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status, c->kp_uniform, GUARD)
#define LAUNCHPH(NN, MM, RW, SL, GUARD) do { if (hrxc) LAUNCHPH2(NN, MM, RW, true, true, SL, GUARD); else if (hru0) LAUNCHPH2(NN, MM, RW, true, false, SL, GUARD); else LAUNCHPH2(NN, MM, RW, false, false, SL, GUARD); } while (0)
#define KP_X(NN, MM)                                                                                   \
        if (n == NN && m == MM) {                                                                      \
            if (raw) {                                                                                 \
                LAUNCHPH(NN, MM, true, false, 1);                                                      \
                hipError_t e_ = launch_fd_kp_difference(c, true);                                      \
                if (e_ != hipSuccess) return e_;                                                       \
                if (hslp) LAUNCHPH(NN, MM, false, true, 0); else LAUNCHPH(NN, MM, false, false, 0);    \
            } else if (hslp) { LAUNCHPH(NN, MM, false, false, 1); LAUNCHPH(NN, MM, false, true, 0); }  \
            else LAUNCHPH(NN, MM, false, false, -1);                                                   \
            return hipGetLastError();                                                                  \
        }
        KP_T1_SHAPES(KP_X)
#undef KP_X
#undef LAUNCHPH
#undef LAUNCHPH2
        return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}
#endif

#if KP_PART(1)
// diagnostic: the one-wave sweep (general form, reads kpc) with the refresh histogram
hipError_t launch_backward_fused_stats(Ctx *c, int pd_stride, int *hist_dev)
{
    const int n = c->n, m = c->d.m;
    const FusedArgs F = fused_args(c);
#define KP_X(NN, MM) if (n == NN && m == MM) { hipLaunchKernelGGL((k_backward_fused_stats<NN, MM>), dim3(c->d.batch), dim3(64), 0, c->stream, c->L, F, c->d.T, c->lambda, pd_stride, c->K, c->k, c->delta_J, c->status, hist_dev); return hipGetLastError(); }
    KP_T1_SHAPES(KP_X)
#undef KP_X
    return hipErrorInvalidValue;
}

#endif
#if KP_PART(3)
hipError_t launch_forward_fused(Ctx *c, double *U_alpha_dev)
{
    const int n = c->n, m = c->d.m;
    dim3 grid(c->d.batch), block(64);
    const bool excl = c->d.batch <= c->n_simd;
    const FusedArgs F = fused_args(c);
    // While a trajectory can have three SIMDs of a CU (4 x batch <= #SIMDs) the state / cost / staging triple runs it: 1.58 ms per
    // sweep at B = 1, 1.74 ... 1.76 at B = 128 ... 256 (four tile sets per wave, a time loop without conditions) against 1.80 for
    // one wave per trajectory with its tile requests four steps ahead, which takes over beyond.
    // KPILQR_FUSED_FWD_WAVES = 1 | 2 | 3 forces a form.
    int form = forward_fused_form(c);
    const bool src_ru0 = c->ru_zero, src_rxc = c->ru_zero && c->rx_const_on;      // the cost wave of the state / cost groups
    const int ncz = (n + 2 + 3) / 4, ncu = (m + 3) / 4;       // tile chunks of [dx; alpha; 1] and of the controls
    int only_ragged = 0;                                      // the general forms behind the uniform pair leave on a uniform set
    c->last_fwd_form_ragged = 0;
    if (form == 4) {
        dim3 block2(128);
        // per-DoF lists: the triple, or (KPILQR_FWD_RAGGED_PAIR=1) the state / cost+staging pair, or one wave per trajectory
        const int behind = 4 * c->d.batch <= c->n_simd ? 3 : c->tune.fwd_ragged_pair ? 2 : 1;
#define LAUNCHSCU(NCZ, NCU)                                                                                             \
        if (ncz == NCZ && ncu == NCU) {                                                                                 \
            if (src_rxc) hipLaunchKernelGGL((k_forward_fused_scu<NCZ, NCU, true, true>), grid, block2, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform);  \
            else if (src_ru0) hipLaunchKernelGGL((k_forward_fused_scu<NCZ, NCU, true, false>), grid, block2, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform);  \
            else hipLaunchKernelGGL((k_forward_fused_scu<NCZ, NCU>), grid, block2, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform);  \
        }
        LAUNCHSCU(4, 2) LAUNCHSCU(2, 1) LAUNCHSCU(4, 1) LAUNCHSCU(3, 1)
#undef LAUNCHSCU
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return e_;
        c->last_fwd_form = 3;                                 // kpilqr_last_launch: "pair" on a uniform set ...
        c->last_fwd_form_ragged = behind == 3 ? 4 : behind == 2 ? 3 : 1;        // ... the triple / pair / w1 otherwise
        c->last_fwd_ru0 = c->ru_zero; c->last_fwd_rxc = src_rxc;
        c->last_fwd_slopes = behind == 1 && c->kps != nullptr;
        if (c->kp_known_uniform) return hipSuccess;           // (the host placed the lists and saw them equal: nothing else can run)
        form = behind; only_ragged = 1;
    } else {
        c->last_fwd_form = form == 3 ? 4 : form == 2 ? 3 : 1;      // kpilqr_last_launch: triple / pair / w1
        c->last_fwd_ru0 = c->ru_zero; c->last_fwd_rxc = src_rxc;
        c->last_fwd_slopes = form == 1 && c->kps != nullptr;
    }
    const bool rxc = form == 1 && c->ru_zero && c->rx_const_on;
    if (form == 3) {
        dim3 block3(192);
#define LAUNCHSC3(NCZ, NCU)                                                                                             \
        if (ncz == NCZ && ncu == NCU) {                                                                                 \
            if (src_rxc) hipLaunchKernelGGL((k_forward_fused_sc3<NCZ, NCU, true, true>), grid, block3, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform, only_ragged); \
            else if (src_ru0) hipLaunchKernelGGL((k_forward_fused_sc3<NCZ, NCU, true, false>), grid, block3, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform, only_ragged); \
            else hipLaunchKernelGGL((k_forward_fused_sc3<NCZ, NCU>), grid, block3, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform, only_ragged); \
            return hipGetLastError();                                                                                   \
        }
        LAUNCHSC3(4, 2) LAUNCHSC3(2, 1) LAUNCHSC3(4, 1) LAUNCHSC3(3, 1)
#undef LAUNCHSC3
        return hipErrorInvalidValue;
    }
    if (form == 2) {
        dim3 block2(128);
#define LAUNCHSC(NCZ, NCU)                                                                                              \
        if (ncz == NCZ && ncu == NCU) {                                                                                 \
            if (src_rxc) hipLaunchKernelGGL((k_forward_fused_sc<NCZ, NCU, true, true>), grid, block2, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform, only_ragged); \
            else if (src_ru0) hipLaunchKernelGGL((k_forward_fused_sc<NCZ, NCU, true, false>), grid, block2, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform, only_ragged); \
            else hipLaunchKernelGGL((k_forward_fused_sc<NCZ, NCU>), grid, block2, 0, c->stream, c->L, F, c->d.T, c->d.n_alpha, \
                               c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred, U_alpha_dev, c->kp_uniform, only_ragged); \
            return hipGetLastError();                                                                                   \
        }
        LAUNCHSC(4, 2) LAUNCHSC(2, 1) LAUNCHSC(4, 1) LAUNCHSC(3, 1)
#undef LAUNCHSC
        return hipErrorInvalidValue;
    }
#define LAUNCH4(NCZ, NCU, RU, UNI, RX)                                                                            \
    do {                                                                                                          \
        if (excl)                                                                                                 \
            hipLaunchKernelGGL((k_forward_fused_excl<NCZ, NCU, RU, UNI, RX>), grid, block, 0, c->stream, c->L, F, c->d.T, \
                               c->d.n_alpha, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred,  \
                               U_alpha_dev, c->kp_uniform);                                                       \
        else                                                                                                      \
            hipLaunchKernelGGL((k_forward_fused<NCZ, NCU, RU, UNI, RX>), grid, block, 0, c->stream, c->L, F, c->d.T,  \
                               c->d.n_alpha, c->K, c->k, c->u_nom, c->ctrl_lim, c->alphas, c->cost_pred,  \
                               U_alpha_dev, c->kp_uniform);                                                       \
    } while (0)
#define LAUNCH3(NCZ, NCU, RU, UNI) do { if (RU && rxc) LAUNCH4(NCZ, NCU, RU, UNI, RU); else LAUNCH4(NCZ, NCU, RU, UNI, false); } while (0)
// both forms, back to back: the one whose kind of key-point set is not resident leaves at once (k_forward_fused)
#define LAUNCH2(NCZ, NCU, RU) do { if (!only_ragged) LAUNCH3(NCZ, NCU, RU, true); LAUNCH3(NCZ, NCU, RU, false); } while (0)
// r_u never uploaded (ru_zero): the instantiation without the r_u loads and the Ju product.  Round-2 history: it measured
// SLOWER at first (3.76 vs 3.22 ms at B = 1024: the compiler's wait placement left the latency shadow it sat in), and
// faster once the uniform-key-point form and the per-trajectory descriptors had changed the loop (2.81 vs 3.03 ms).
#define LAUNCH(NCZ, NCU) do { if (c->ru_zero) LAUNCH2(NCZ, NCU, true); else LAUNCH2(NCZ, NCU, false); } while (0)
#define KP_X(NCZ, NCU) if (ncz == NCZ && ncu == NCU) { LAUNCH(NCZ, NCU); return hipGetLastError(); }
    KP_X(4, 2) KP_X(2, 1) KP_X(4, 1) KP_X(3, 1)
#undef KP_X
#undef LAUNCH
#undef LAUNCH2
#undef LAUNCH3
#undef LAUNCH4
    return hipErrorInvalidValue;
}
#endif

}  // namespace kpilqr
