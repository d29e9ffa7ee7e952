// keypoints.hip -- key-point placement on the device for a whole batch:
// KeypointGenerator::GenerateKeyPointsSetInterval (src/KeyPointGenerator/KeyPointGenerator.cpp:319-339),
// GenerateJerkProfile / GenerateAccellerationProfile + GenerateKeyPointsAdaptive (:730-770 / :772-795, :341-382) and GenerateVelocityProfile +
// GenerateKeyPointsVelocityChange (:797-808, :642-728), producing the context's per-DoF CSR lists directly
// (what kpilqr_set_keypoints would have been given).  The placement of one DoF never looks at another DoF,
// so one thread owns one (trajectory, DoF) list and walks the horizon; the states of 64 steps at a time are
// staged through LDS with coalesced loads.  Decisions are bit-exact with the oracle: same operation order,
// IEEE division, no FMA contraction (this file is compiled -ffp-contract=off).
//
// Pass 1 writes a 64-step bitmask per list and chunk plus the list's count, pass 2 is an exclusive scan of
// the counts (kp_offsets), pass 3 expands the bitmasks into kp_times.  Duplicate entries the reference emits
// for the last row (velocity_change, :724-727) collapse: a DoF is a key-point at a step or it is not.
#include "common.h"

namespace kpilqr {

#define KP_CHUNK 64

// method: 0 set_interval, 1 adaptive_jerk, 2 velocity_change, 3 adaptive_accel
__global__ void __launch_bounds__(64)
k_kp_flags(int method, int dof, int T, int min_N, int max_N, double dt, const double *__restrict__ thr,
           const double *__restrict__ X, unsigned long long *__restrict__ mask, int *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) double sx[];      // [(KP_CHUNK + 2)][dof] velocities
    const int n = 2 * dof;
    const int b = blockIdx.x, j = threadIdx.x;
    const int nchunks = (T + KP_CHUNK - 1) / KP_CHUNK;
    const double *Xb = X + (size_t)b * T * n;
    const double th = (j < dof && thr) ? thr[j] : 0.0;
    // per-list state
    int last = 0;                        // adaptive_jerk: last key-point            (:356 last_indices)
    int counter = 0;                     // velocity_change                            (:660)
    double last_val = 0.0, last_dir = 0.0;
    int cnt = 0;
    for (int ch = 0; ch < nchunks; ch++) {
        const int t0 = ch * KP_CHUNK;
        // stage velocities of steps t0-1 .. t0+KP_CHUNK+1 (row 0 of sx <-> step t0-1)
        __syncthreads();
        for (int e = threadIdx.x; e < (KP_CHUNK + 3) * dof; e += blockDim.x) {
            const int row = e / dof, i = e - row * dof;
            const int t = t0 - 1 + row;
            sx[e] = (t >= 0 && t < T) ? Xb[(size_t)t * n + dof + i] : 0.0;
        }
        __syncthreads();
        if (j < dof) {
            unsigned long long w = 0;
            const int tend = min(KP_CHUNK, T - t0);
            for (int tt = 0; tt < tend; tt++) {
                const int t = t0 + tt;
                bool key = false;
                if (t == 0 || t == T - 1) {
                    key = true;                                             // rows 0 and T-1 are always full
                    if (method == 2 && t == T - 1 && T > 1) {
                        // velocity_change still runs its update at t = T-1 before the enforced last row (:724-727);
                        // the outcome is a key-point either way
                    }
                } else if (method == 0) {
                    key = (t % min_N) == 0;                                 // :328-338
                } else if (method == 1 || method == 3) {
                    // jerk[t] from the velocities of steps t, t+1, t+2; the last two rows of the profile are 0 (:757-761)
                    double jerk = 0.0;
                    if (method == 3) {
                        // adaptive_accel (:98-101): the profile is the signed velocity difference v[t+1] - v[t] (:785-788)
                        jerk = sx[(tt + 2) * dof + j] - sx[(tt + 1) * dof + j];
                    } else if (t < T - 2) {
                        const double v1 = sx[(tt + 1) * dof + j], v2 = sx[(tt + 2) * dof + j], v3 = sx[(tt + 3) * dof + j];
                        const double a1 = (v2 - v1) / dt;                  // :748
                        const double a2 = (v3 - v2) / dt;                  // :749
                        jerk = fabs((a2 - a1) / dt);                       // :752
                    }
                    if ((t - last) >= min_N) {                              // :359
                        if (jerk > th) { key = true; last = t; }
                    }
                    if ((t - last) >= max_N) { key = true; last = t; }     // :365
                }
                if (method == 2 && t >= 1) {                                // :660-716 (runs for t = 1 .. T-1)
                    counter++;
                    const double v = sx[(tt + 1) * dof + j], vp = sx[tt * dof + j];
                    const double dir = v - vp;                              // :671
                    last_val += fabs(v);                                    // :673
                    bool hit = false;
                    if (counter >= min_N) {                                 // :676
                        if (fabs(last_val) > th) hit = true;
                    }
                    if (!hit) {
                        if (counter >= min_N) {                             // :687
                            if (dir * last_dir < 0) hit = true;
                        } else {
                            last_dir = dir;                                 // :697-699
                        }
                    }
                    if (!hit && counter >= max_N) hit = true;               // :702
                    if (hit) { last_val = 0.0; counter = 0; key = true; }
                }
                if (key) { w |= 1ull << tt; cnt++; }
            }
            mask[((size_t)b * dof + j) * nchunks + ch] = w;
        }
    }
    if (j < dof) count[(size_t)b * dof + j] = cnt;
}

// exclusive scan of counts[0..nlists) -> offsets[0..nlists]; one workgroup
__global__ void __launch_bounds__(1024)
k_kp_scan(int nlists, const int *__restrict__ count, int *__restrict__ offsets)
{
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (nlists + 1023) / 1024;
    const int lo = min(tid * per, nlists), hi = min(lo + per, nlists);
    int s = 0;
    for (int i = lo; i < hi; i++) s += count[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (tid >= d) ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; i++) { offsets[i] = run; run += count[i]; }
    if (tid == 1023) offsets[nlists] = part[1023];
}

__global__ void __launch_bounds__(256)
k_kp_fill(int nlists, int nchunks, const unsigned long long *__restrict__ mask, const int *__restrict__ offsets,
          int *__restrict__ times)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlists) return;
    int at = offsets[l];
    for (int ch = 0; ch < nchunks; ch++) {
        unsigned long long w = mask[(size_t)l * nchunks + ch];
        while (w) {
            const int tz = __builtin_ctzll(w);
            times[at++] = ch * KP_CHUNK + tz;
            w &= w - 1;
        }
    }
}

hipError_t launch_generate_keypoints(Ctx *c, int method, int min_N, int max_N, double dt, const double *thr_dev,
                                     const double *X_dev, unsigned long long *mask_dev, int *count_dev)
{
    const int dof = c->d.dof, T = c->d.T, nlists = c->d.batch * dof;
    const int nchunks = (T + KP_CHUNK - 1) / KP_CHUNK;
    const size_t lds = sizeof(double) * (size_t)(KP_CHUNK + 3) * dof;
    hipLaunchKernelGGL(k_kp_flags, dim3(c->d.batch), dim3(64), lds, c->stream, method, dof, T, min_N, max_N, dt, thr_dev,
                       X_dev, mask_dev, count_dev);
    hipLaunchKernelGGL(k_kp_scan, dim3(1), dim3(1024), 0, c->stream, nlists, count_dev, c->kp_offsets);
    hipLaunchKernelGGL(k_kp_fill, dim3((nlists + 255) / 256), dim3(256), 0, c->stream, nlists, nchunks, mask_dev,
                       c->kp_offsets, c->kp_times);
    return hipGetLastError();
}

// ---- iterative_error: the error test of one bisection level, for the whole batch -------------------------------------------
// KeypointGenerator::CheckDOFColumnError (src/KeyPointGenerator/KeyPointGenerator.cpp:550-640): an interval [s, e] of DoF i
// is good when it is no longer than min_N (:562-564) or when the mean, over the velocity rows of the DoF's two A columns,
// of the squared difference between the column at the midpoint and the mean of the columns at the ends stays below the
// threshold (:608-639).  The three key-point columns are read from the step records (the host differenced them at
// s, (s+e)/2 and e and k_fd_difference wrote them).  One thread per interval, the reference's summation order (compiled
// without contraction): decisions are bit-identical with the oracle.
__global__ void __launch_bounds__(256)
k_kp_error_test(RecLayout L, int T, int batch, int dof, int n_iv, const int *__restrict__ iv, int min_N, double threshold,
                const double *__restrict__ rec, unsigned char *__restrict__ good)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_iv) return;
    const int b = iv[4 * k], i = iv[4 * k + 1], s = iv[4 * k + 2], e = iv[4 * k + 3];
    if ((unsigned)b >= (unsigned)batch || (unsigned)i >= (unsigned)dof || s < 0 || e >= T || e < s) { good[k] = 2; return; }   // bad argument
    if (e - s <= min_N) { good[k] = 1; return; }
    const int mid = (s + e) / 2, n = L.n;
    const double *Rs = rec + ((size_t)b * T + s) * L.stride + L.off_A, *Rm = rec + ((size_t)b * T + mid) * L.stride + L.off_A,
                 *Re = rec + ((size_t)b * T + e) * L.stride + L.off_A;
    double error_sum = 0.0;
    int counter = 0;
    for (int cc = 0; cc < 2; cc++) {
        const int col = cc == 0 ? i : i + dof;
        for (int j = dof; j < 2 * dof; j++) {
            const double approx = (Rs[col * n + j] + Re[col * n + j]) / 2;
            const double d = Rm[col * n + j] - approx;
            error_sum += d * d;
            counter++;
        }
    }
    const double average_error = counter > 0 ? error_sum / counter : 0.0;
    good[k] = average_error < threshold ? 1 : 0;
}

hipError_t launch_kp_error_test(Ctx *c, int n_iv, const int *iv_dev, int min_N, double threshold, unsigned char *good_dev)
{
    if (n_iv == 0) return hipSuccess;
    hipLaunchKernelGGL(k_kp_error_test, dim3((n_iv + 255) / 256), dim3(256), 0, c->stream, c->L, c->d.T, c->d.batch, c->d.dof, n_iv,
                       iv_dev, min_N, threshold, c->rec, good_dev);
    return hipGetLastError();
}

}  // namespace kpilqr
