// tracker.h -- a4 (KeypointGenerator::InterpolateDerivatives, src/KeyPointGenerator/KeyPointGenerator.cpp:840-954) evaluated
// in registers by the tiled sweeps: a lane that holds NV elements of ONE column of A (or of B) walks the key-point list
// of that column's DoF and keeps (value at the segment start, slope):
//     x_t = x_s + (t - s) * ((x_e - x_s) / (e - s))                                   (:898-905, :933-948)
// in k_interpolate's operation order (un-contracted multiply-add, correctly rounded division), so the values are the
// ones the materialising kernel would have written.  The key-point columns themselves are read from the step records
// (k_fd_difference wrote them); the whole trajectory's records sit behind one buffer descriptor, and a lane with
// nothing to load carries an out-of-range offset and reads 0.  Same scheme as fused_mfma.hip (one-tile shapes).
#pragma once
#include "mfma_common.h"

namespace kpilqr {

#define KP_BIGT 0x3fffffff

// correctly rounded a/den for a normal-range quotient: one residual correction on top of a ~1 ulp reciprocal
// (the hardware division expansion without its scaling / fix-up tail; den is a small positive integer)
__device__ __forceinline__ double kp_fdiv(double a, double den, double rinv)
{
    const double q0 = a * rinv;
    const double rr = __builtin_fma(-den, q0, a);
    return __builtin_fma(rr, rinv, q0);
}
// start + dt*slope, never contracted (k_interpolate's operation order)
__device__ __forceinline__ double kp_lerp_nc(double sv, double dt, double av)
{
#pragma clang fp contract(off)
    const double p = dt * av;
    return sv + p;
}
// NV record elements of step tk (nothing when tk is not in [0, T): the base then points past every descriptor, and so does
// an element whose offset is KP_OOB -- sums stay below 2^32 because T * stride * 8 < 0x7ffffff0 is required).  The base is
// made opaque to the optimiser: it otherwise turns the select into a branch with a load in either arm, and loads inside
// divergent control flow cost a full drain of the memory pipe at the next wait.
template <int NV>
__device__ __forceinline__ void kp_load_vals(__amdgpu_buffer_rsrc_t rT, const int *offs, int tk, int T, int strideB, double *out)
{
    int base = ((unsigned)tk < (unsigned)T) ? tk * strideB : KP_OOB;
    asm volatile("" : "+v"(base));
#pragma unroll
    for (int r = 0; r < NV; r++) out[r] = kp_bld(rT, base + offs[r]);
}

// Both trackers are driven once per time-step by  consume(); step(t); <value(...)>; issue();
// Every global load sits in issue() and is UNCONDITIONAL (a lane that needs nothing carries an out-of-range offset and
// gets 0 back without a memory access), and its result is looked at one whole step later, in consume().  With loads
// inside the per-lane crossing branch the compiler cannot count the outstanding ones and drains the memory pipe
// (s_waitcnt vmcnt(0)) every step -- with ragged per-DoF lists some lane crosses at almost every step, so every step
// paid a full memory latency.

// Walking DOWN in time (backward sweep): a segment is entered from its END; its start column was requested when the
// previous segment was entered.
template <int NV>
struct KpDownTracker {
    int offs[NV];
    int lo, idx, s, nb, nb2, nb3, tnb3;
    bool fresh;                                  // tv holds the column at `nb`, requested by the last issue()
    bool tok;                                    // tnb3 (raw load of the last issue()) is a list entry
    double sv[NV], av[NV], pv[NV], tv[NV];
    const int *times;
    __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t rT, const int *kp_offsets, const int *kp_times, bool has, size_t list, int T, int strideB)
    {
        times = kp_times;
        lo = has ? kp_offsets[list] : 0;
        const int hi = has ? kp_offsets[list + 1] : 0;
        idx = hi - 1;
        s = has ? kp_times[idx] : -1;                    // == T-1 for canonical key-points
        nb = (has && idx - 1 >= lo) ? kp_times[idx - 1] : -1;
        nb2 = (has && idx - 2 >= lo) ? kp_times[idx - 2] : -1;
        nb3 = (has && idx - 3 >= lo) ? kp_times[idx - 3] : -1;
        tnb3 = nb3; tok = nb3 >= 0;
        kp_load_vals<NV>(rT, offs, s, T, strideB, sv);
        kp_load_vals<NV>(rT, offs, nb, T, strideB, pv);
        fresh = false;
#pragma unroll
        for (int i = 0; i < NV; i++) { av[i] = 0.0; tv[i] = 0.0; }
    }
    __device__ __forceinline__ void consume()
    {
#pragma unroll
        for (int i = 0; i < NV; i++) pv[i] = fresh ? tv[i] : pv[i];
        fresh = false;
        nb3 = tok ? tnb3 : -1;                       // the loaded value is first looked at here, a step after its request
    }
    __device__ __forceinline__ void step(int t)
    {
        if (t < s) {                                 // per lane: crossed the start of the current segment
            const double den = (double)(s - nb);
            const double rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const double ev = sv[i];
                sv[i] = pv[i];
                av[i] = kp_fdiv(ev - sv[i], den, rinv);
            }
            s = nb; idx--;
            nb = nb2; nb2 = nb3;
            fresh = true;                            // the column at the new `nb` is wanted
        }
    }
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rT, int T, int strideB)
    {
        kp_load_vals<NV>(rT, offs, fresh ? nb : -1, T, strideB, tv);
        const int j = idx - 3;
        int jj = j >= lo ? j : lo;                   // always a valid address
        asm volatile("" : "+v"(jj));
        tnb3 = times[jj];
        tok = j >= lo;
    }
    __device__ __forceinline__ double value(int i, double dt) const { return kp_lerp_nc(sv[i], dt, av[i]); }
};

// Walking UP in time (forward sweep): a segment is entered at its START key-point, where the stored value is exact;
// its end column is requested at the crossing and the slope formed one step later.
template <int NV>
struct KpUpTracker {
    int offs[NV];
    int hi, idx, s, e, nb, nb2, tnb2;
    bool pend;                                   // tv holds the column at `e`, requested by the last issue()
    bool tok;                                    // tnb2 (raw load of the last issue()) is a list entry
    double sv[NV], ev[NV], av[NV], tv[NV];
    const int *times;
    __device__ __forceinline__ void init(__amdgpu_buffer_rsrc_t rT, const int *kp_offsets, const int *kp_times, bool has, size_t list, int T, int strideB)
    {
        times = kp_times;
        const int lo = has ? kp_offsets[list] : 0;
        hi = has ? kp_offsets[list + 1] : 0;
        idx = lo;
        s = has ? kp_times[idx] : 0;                      // == 0 for canonical key-points
        e = (has && idx + 1 < hi) ? kp_times[idx + 1] : KP_BIGT;
        nb = (has && idx + 2 < hi) ? kp_times[idx + 2] : KP_BIGT;
        nb2 = (has && idx + 3 < hi) ? kp_times[idx + 3] : KP_BIGT;
        tnb2 = nb2; tok = nb2 != KP_BIGT;
        kp_load_vals<NV>(rT, offs, has ? s : KP_BIGT, T, strideB, sv);
        kp_load_vals<NV>(rT, offs, e, T, strideB, tv);
        pend = true;
#pragma unroll
        for (int i = 0; i < NV; i++) { av[i] = 0.0; ev[i] = 0.0; }
    }
    __device__ __forceinline__ void consume()
    {
        if (pend) {                           // slope of the segment entered one step ago (its end column has landed)
            const double den = (double)(e - s);
            const double rinv = kp_rcp(den);
#pragma unroll
            for (int i = 0; i < NV; i++) { ev[i] = tv[i]; av[i] = (e != KP_BIGT) ? kp_fdiv(ev[i] - sv[i], den, rinv) : 0.0; }
            pend = false;
        }
        nb2 = tok ? tnb2 : KP_BIGT;
    }
    __device__ __forceinline__ void step(int t)
    {
        if (t >= e) {                         // per lane: reached the end key-point of the segment
#pragma unroll
            for (int i = 0; i < NV; i++) { sv[i] = ev[i]; av[i] = 0.0; }
            s = e; e = nb; nb = nb2; idx++;
            pend = true;
        }
    }
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rT, int T, int strideB)
    {
        kp_load_vals<NV>(rT, offs, pend ? e : KP_BIGT, T, strideB, tv);
        const int j = idx + 3;
        int jj = j < hi ? j : (hi > 0 ? hi - 1 : 0);
        asm volatile("" : "+v"(jj));
        tnb2 = times[jj];
        tok = j < hi;
    }
    __device__ __forceinline__ double value(int i, double dt) const { return kp_lerp_nc(sv[i], dt, av[i]); }
};

}  // namespace kpilqr
