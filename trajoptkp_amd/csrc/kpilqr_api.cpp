// kpilqr_api.cpp -- the C ABI of libkpilqr.so (include/kpilqr.h): context lifetime, pinned
// staging, uploads/downloads and kernel dispatch.  No CPU fallback lives here: without a HIP
// device kpilqr_create fails with KPILQR_ERR_NO_DEVICE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

using namespace kpilqr;

static thread_local std::string g_err;

struct kpilqr_ctx : public Ctx {};

static int set_err(kpilqr_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    g_err = msg;
    return code;
}

template <class T>
static hipError_t dalloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    return hipMalloc((void **)p, count * sizeof(T));
}

template <class T>
static int regrow(kpilqr_ctx *c, T **p, size_t count)
{
    if (*p) KP_HIP(c, hipFree(*p));
    *p = nullptr;
    KP_HIP(c, dalloc(p, count));
    return KPILQR_OK;
}

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

namespace kpilqr {
Ctx::Tuning read_tuning_from_env()
{
    Ctx::Tuning t;
    t.fused_bwd_waves = env_int("KPILQR_FUSED_WAVES", 0);
    t.fused_fwd_waves = env_int("KPILQR_FUSED_FWD_WAVES", 0);
    t.role_shift = env_int("KPILQR_ROLE_SHIFT", 9);
    t.tiled_nt_min = env_int("KPILQR_TILED_NT_MIN", 0);
    t.tiled_a6 = env_int("KPILQR_TILED_A6", -1);
    t.tiled_a4 = env_int("KPILQR_TILED_A4", -1);
    return t;
}
}  // namespace kpilqr

extern "C" {

int kpilqr_version(void) { return KPILQR_VERSION; }

const char *kpilqr_strerror(kpilqr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int kpilqr_create(const kpilqr_dims *dims, void *stream, kpilqr_ctx **out)
{
    if (!dims || !out) return set_err(nullptr, KPILQR_ERR_ARG, "null argument");
    *out = nullptr;
    if (dims->dof < 1 || dims->m < 1 || dims->T < 2 || dims->nr < 1 || dims->batch < 1 || dims->n_alpha < 1)
        return set_err(nullptr, KPILQR_ERR_ARG, "dims out of range");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return set_err(nullptr, KPILQR_ERR_NO_DEVICE,
                       std::string("no HIP device available (libkpilqr has no CPU fallback): ") +
                           (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    if (dims->device < 0 || dims->device >= ndev) return set_err(nullptr, KPILQR_ERR_ARG, "device ordinal out of range");
    if (fd_difference_waves(2 * dims->dof, dims->m) < 1)
        return set_err(nullptr, KPILQR_ERR_ARG, "state/control dimensions too large: one key-point's FD columns ((n+m)(n+1) doubles) exceed the CU's 160 KB of LDS");
    if (hipSetDevice(dims->device) != hipSuccess) return set_err(nullptr, KPILQR_ERR_NO_DEVICE, "hipSetDevice failed");

    kpilqr_ctx *c = new (std::nothrow) kpilqr_ctx();
    if (!c) return set_err(nullptr, KPILQR_ERR_ALLOC, "host allocation failed");
    c->tune = read_tuning_from_env();           // the only place the environment is looked at
    c->d = *dims;
    c->n = 2 * dims->dof;
    c->L = RecLayout(c->n, dims->m);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dims->device) == hipSuccess && prop.multiProcessorCount > 0)
            c->n_simd = prop.multiProcessorCount * 4;
    }

    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c; return set_err(nullptr, KPILQR_ERR_HIP, "hipStreamCreate failed");
        }
        c->own_stream = true;
    }

    const size_t B = dims->batch, T = dims->T, n = c->n, m = dims->m, nr = dims->nr;
    hipError_t rc = hipSuccess;
#define TRY(x) do { if (rc == hipSuccess) rc = (x); } while (0)
    TRY(dalloc(&c->rec, B * T * c->L.stride));
    TRY(dalloc(&c->K, B * T * n * m));
    TRY(dalloc(&c->k, B * T * m));
    TRY(dalloc(&c->r, B * (T + 1) * nr));
    TRY(dalloc(&c->r_x, B * (T + 1) * nr * n));
    TRY(dalloc(&c->r_u, B * (T + 1) * nr * m));
    TRY(dalloc(&c->w_run, nr));
    TRY(dalloc(&c->w_term, nr));
    TRY(dalloc(&c->u_nom, B * T * m));
    TRY(dalloc(&c->ctrl_lim, 2 * m));
    TRY(dalloc(&c->lambda, B));
    TRY(dalloc(&c->alphas, (size_t)dims->n_alpha));
    TRY(dalloc(&c->cost_pred, B * dims->n_alpha));
    TRY(dalloc(&c->delta_J, B));
    TRY(dalloc(&c->traj_cost, B));
    TRY(dalloc(&c->status, B));
    TRY(dalloc(&c->segmap, B * dims->dof * T));
    TRY(dalloc(&c->kp_offsets, B * dims->dof + 1));
#undef TRY
    if (rc != hipSuccess) {
        std::string msg = std::string("hipMalloc failed: ") + hipGetErrorString(rc);
        kpilqr_destroy(c);
        return set_err(nullptr, KPILQR_ERR_ALLOC, msg);
    }
    // records start zeroed so that padding / never-written columns are defined
    (void)hipMemsetAsync(c->rec, 0, B * T * c->L.stride * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->K, 0, B * T * n * m * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->k, 0, B * T * m * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->r, 0, B * (T + 1) * nr * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->r_x, 0, B * (T + 1) * nr * n * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->r_u, 0, B * (T + 1) * nr * m * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->u_nom, 0, B * T * m * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->status, 0, B * sizeof(int), c->stream);

    const bool generic = (dims->flags & KPILQR_FLAG_GENERIC_KERNELS) != 0;
    const bool force_tiled = (dims->flags & KPILQR_FLAG_TILED_KERNELS) != 0;
    c->bwd_variant = (!generic && !force_tiled && backward_mfma_supported(c->n, dims->m)) ? "mfma_f64_t1"
                   : (!generic && backward_tiled_supported(c->n, dims->m, c->tune.tiled_nt_min)) ? "mfma_f64_tiled" : "generic_lds";
    c->fwd_variant = (!generic && !force_tiled && forward_mfma_supported(c->n, dims->m, dims->n_alpha)) ? "mfma_f64_t1"
                   : (!generic && forward_tiled_supported(c->n, dims->m, dims->n_alpha, c->tune.tiled_nt_min)) ? "mfma_f64_tiled" : "generic_lds";
    if ((dims->flags & KPILQR_FLAG_FUSED) && !generic && !force_tiled &&
        fused_supported(c->n, dims->m, dims->nr, dims->dof, dims->T, c->L.stride, dims->n_alpha)) {
        c->fused = true;
        c->bwd_variant = c->fwd_variant = "mfma_f64_t1_fused";
    }
    // the same flag on a tiled shape: a6 (cost derivatives) inside the sweeps, A and B still materialised by k_interpolate
    // It replaces k_cost_derivs (HBM-bound: n^2 doubles written per step, so its time grows with batch x n^2) by
    // NT*ceil(nr/4) + 6 MFMAs per wave-step of the latency-bound backward sweep (+9 % at four tiles, whatever the batch):
    // a gain from ~100 trajectories of a four-tile state up (n = 62, B = 128, T = 5000: 74.4 -> 70.6 ms), a loss for two or
    // three tiles at the batches measured.  KPILQR_TILED_A6 = 0 | 1 overrides the choice.
    if ((dims->flags & KPILQR_FLAG_FUSED) && !c->fused && dims->nr <= 16 &&
        strcmp(c->bwd_variant, "mfma_f64_tiled") == 0 && strcmp(c->fwd_variant, "mfma_f64_tiled") == 0) {
        const bool want = c->tune.tiled_a6 >= 0 ? c->tune.tiled_a6 != 0
                                                : (tiled_tiles(c->n, c->tune.tiled_nt_min) == 4 && dims->batch >= 96);
        if (want) {
            c->tiled_a6 = true;
            c->bwd_variant = c->fwd_variant = "mfma_f64_tiled_a6";
        }
    }
    if (strcmp(c->bwd_variant, "generic_lds") == 0 && backward_generic_lds_bytes(c->n, dims->m) > 160 * 1024) {
        kpilqr_destroy(c);
        return set_err(nullptr, KPILQR_ERR_ARG, "state dimension too large for the generic backward kernel (LDS)");
    }
    *out = c;
    return KPILQR_OK;
}

void kpilqr_destroy(kpilqr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->d.device);
    (void)hipStreamSynchronize(c->stream);
    comm_destroy(c);
    void *ptrs[] = {c->rec, c->K, c->k, c->r, c->r_x, c->r_u, c->w_run, c->w_term, c->u_nom, c->ctrl_lim,
                    c->lambda, c->alphas, c->cost_pred, c->delta_J, c->traj_cost, c->status, c->segmap,
                    c->kp_offsets, c->kp_times, c->X_states, c->kp_thr, c->kp_mask, c->kp_count, c->ls8, c->job_b, c->job_t, c->job_col, c->job_nom, c->job_mode,
                    c->xplus, c->xminus, c->xnom, c->stage, c->slot_start};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int kpilqr_get_dims(kpilqr_ctx *c, kpilqr_dims *out)
{
    if (!c || !out) return KPILQR_ERR_ARG;
    *out = c->d;
    return KPILQR_OK;
}

int kpilqr_host_alloc(kpilqr_ctx *c, size_t bytes, void **pinned)
{
    if (!c || !pinned) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    KP_HIP(c, hipHostMalloc(pinned, bytes ? bytes : 1, hipHostMallocDefault));
    return KPILQR_OK;
}

int kpilqr_host_free(kpilqr_ctx *c, void *pinned)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (pinned) KP_HIP(c, hipHostFree(pinned));
    return KPILQR_OK;
}

int kpilqr_sync(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_device_ptr(kpilqr_ctx *c, int which, void **dptr, size_t *bytes)
{
    if (!c || !dptr) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t B = c->d.batch, T = c->d.T, n = c->n, m = c->d.m, nr = c->d.nr;
    void *p = nullptr; size_t sz = 0;
    switch (which) {
    case KPILQR_BUF_STEP_RECORDS: p = c->rec; sz = B * T * c->L.stride * 8; break;
    case KPILQR_BUF_K: p = c->K; sz = B * T * n * m * 8; break;
    case KPILQR_BUF_k: p = c->k; sz = B * T * m * 8; break;
    case KPILQR_BUF_RESIDUALS: p = c->r; sz = B * (T + 1) * nr * 8; break;
    case KPILQR_BUF_R_X: p = c->r_x; sz = B * (T + 1) * nr * n * 8; break;
    case KPILQR_BUF_R_U: p = c->r_u; sz = B * (T + 1) * nr * m * 8; break;
    case KPILQR_BUF_U_NOM: p = c->u_nom; sz = B * T * m * 8; break;
    case KPILQR_BUF_FD_XPLUS: p = c->xplus; sz = (size_t)c->njobs * n * 8; break;
    case KPILQR_BUF_FD_XMINUS: p = c->xminus; sz = (size_t)c->njobs * n * 8; break;
    case KPILQR_BUF_COST_PRED: p = c->cost_pred; sz = B * c->d.n_alpha * 8; break;
    case KPILQR_BUF_DELTA_J: p = c->delta_J; sz = B * 8; break;
    case KPILQR_BUF_STATUS: p = c->status; sz = B * 4; break;
    default: return set_err(c, KPILQR_ERR_ARG, "unknown buffer id");
    }
    *dptr = p;
    if (bytes) *bytes = sz;
    return KPILQR_OK;
}

// ---- STEP 1b ------------------------------------------------------------------------------------
int kpilqr_set_keypoints(kpilqr_ctx *c, const int *kp_offsets, const int *kp_times)
{
    if (!c || !kp_offsets || !kp_times) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t nlists = (size_t)c->d.batch * c->d.dof;
    const int total = kp_offsets[nlists];
    if (kp_offsets[0] != 0 || total < 0) return set_err(c, KPILQR_ERR_ARG, "kp_offsets must start at 0");
    for (size_t i = 0; i < nlists; i++)
        if (kp_offsets[i + 1] < kp_offsets[i]) return set_err(c, KPILQR_ERR_ARG, "kp_offsets not monotone");
    for (int i = 0; i < total; i++)
        if (kp_times[i] < 0 || kp_times[i] >= c->d.T) return set_err(c, KPILQR_ERR_ARG, "kp_times out of [0,T)");
    bool canonical = true;
    for (size_t i = 0; i < nlists && canonical; i++) {
        const int a = kp_offsets[i], e = kp_offsets[i + 1];
        if (e <= a || kp_times[a] != 0 || kp_times[e - 1] != c->d.T - 1) { canonical = false; break; }
        for (int j = a + 1; j < e; j++) if (kp_times[j] <= kp_times[j - 1]) { canonical = false; break; }
    }
    c->kp_canonical = canonical;
    if ((size_t)total > c->kp_cap) {
        if (c->kp_times) { KP_HIP(c, hipStreamSynchronize(c->stream)); KP_HIP(c, hipFree(c->kp_times)); c->kp_times = nullptr; }
        c->kp_cap = (size_t)total + (size_t)total / 4 + 64;
        KP_HIP(c, dalloc(&c->kp_times, c->kp_cap));
    }
    KP_HIP(c, hipMemcpyAsync(c->kp_offsets, kp_offsets, (nlists + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, hipMemcpyAsync(c->kp_times, kp_times, (size_t)total * sizeof(int), hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, launch_build_segmap(c));
    // the host arrays may be pageable: make the copies complete before returning control
    KP_HIP(c, hipStreamSynchronize(c->stream));
    c->have_kp = true;
    return KPILQR_OK;
}

static int ensure_stage(kpilqr_ctx *c, size_t bytes);

// ---- key-point placement on the device ----------------------------------------------------------------
int kpilqr_upload_states(kpilqr_ctx *c, const double *X)
{
    if (!c || !X) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t count = (size_t)c->d.batch * c->d.T * c->n;
    if (!c->X_states) KP_HIP(c, hipMalloc((void **)&c->X_states, count * sizeof(double)));
    KP_HIP(c, hipMemcpyAsync(c->X_states, X, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->have_states = true;
    return KPILQR_OK;
}

int kpilqr_generate_keypoints(kpilqr_ctx *c, const char *method, int min_N, int max_N, const double *thresholds, double dt)
{
    if (!c || !method) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    int mth = -1;
    if (strcmp(method, "set_interval") == 0) mth = 0;
    else if (strcmp(method, "adaptive_jerk") == 0) mth = 1;
    else if (strcmp(method, "velocity_change") == 0) mth = 2;
    else return set_err(c, KPILQR_ERR_ARG, "kpilqr_generate_keypoints: method must be set_interval, adaptive_jerk or velocity_change "
                                           "(iterative_error interleaves host finite differences and stays on the host)");
    if (min_N < 1 || max_N < 1) return set_err(c, KPILQR_ERR_ARG, "min_N and max_N must be >= 1");
    if (c->d.dof > 64) return set_err(c, KPILQR_ERR_ARG, "on-device key-point placement supports dof <= 64");
    if (mth != 0 && (!thresholds || !(dt > 0.0))) return set_err(c, KPILQR_ERR_ARG, "thresholds and a positive dt are required");
    if (mth != 0 && !c->have_states) return set_err(c, KPILQR_ERR_STATE, "kpilqr_generate_keypoints before kpilqr_upload_states");
    const size_t nlists = (size_t)c->d.batch * c->d.dof, T = c->d.T, nchunks = (T + 63) / 64;
    if (!c->kp_thr) KP_HIP(c, hipMalloc((void **)&c->kp_thr, sizeof(double) * c->d.dof));
    if (!c->kp_mask) KP_HIP(c, hipMalloc((void **)&c->kp_mask, sizeof(unsigned long long) * nlists * nchunks));
    if (!c->kp_count) KP_HIP(c, hipMalloc((void **)&c->kp_count, sizeof(int) * nlists));
    if (!c->X_states) KP_HIP(c, hipMalloc((void **)&c->X_states, sizeof(double) * c->d.batch * T * c->n));   // set_interval never reads it
    if (nlists * T > c->kp_cap) {            // worst case: every step a key-point
        KP_HIP(c, hipStreamSynchronize(c->stream));
        if (c->kp_times) KP_HIP(c, hipFree(c->kp_times));
        c->kp_times = nullptr; c->kp_cap = 0;
        KP_HIP(c, dalloc(&c->kp_times, nlists * T));
        c->kp_cap = nlists * T;
    }
    if (thresholds) {
        KP_HIP(c, hipMemcpyAsync(c->kp_thr, thresholds, sizeof(double) * c->d.dof, hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipStreamSynchronize(c->stream));       // the host array may be pageable
    }
    KP_HIP(c, launch_generate_keypoints(c, mth, min_N, max_N, dt, thresholds ? c->kp_thr : nullptr, c->X_states, c->kp_mask, c->kp_count));
    KP_HIP(c, launch_build_segmap(c));
    c->have_kp = true;
    c->kp_canonical = true;      // rows 0 and T-1 are always full and the lists are strictly increasing by construction
    return KPILQR_OK;
}

int kpilqr_get_keypoints(kpilqr_ctx *c, int *kp_offsets, int *kp_times, int times_capacity)
{
    if (!c || !kp_offsets) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "no key-points set");
    const size_t nlists = (size_t)c->d.batch * c->d.dof;
    KP_HIP(c, hipMemcpyAsync(kp_offsets, c->kp_offsets, (nlists + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    const int total = kp_offsets[nlists];
    if (kp_times) {
        if (total > times_capacity) return set_err(c, KPILQR_ERR_ARG, "kp_times capacity too small");
        KP_HIP(c, hipMemcpyAsync(kp_times, c->kp_times, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        KP_HIP(c, hipStreamSynchronize(c->stream));
    }
    return total;
}

int kpilqr_upload_fd(kpilqr_ctx *c, int njobs, const int *job_b, const int *job_t, const int *job_col,
                     const unsigned char *job_mode, const int *job_nom, const double *xplus,
                     const double *xminus, int nnom, const double *xnom, double eps)
{
    if (!c || njobs < 0 || nnom < 0) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (njobs > 0 && (!job_b || !job_t || !job_col || !job_mode || !xplus || !xminus))
        return set_err(c, KPILQR_ERR_ARG, "null FD job array");
    if (!(eps > 0.0)) return set_err(c, KPILQR_ERR_ARG, "eps must be positive");
    const int n = c->n;
    // validate indices on the host: a bad index would be an out-of-bounds device write
    for (int j = 0; j < njobs; j++) {
        if (job_b[j] < 0 || job_b[j] >= c->d.batch || job_t[j] < 0 || job_t[j] >= c->d.T ||
            job_col[j] < 0 || job_col[j] >= n + c->d.m || job_mode[j] > 2)
            return set_err(c, KPILQR_ERR_ARG, "FD job index out of range");
        if (job_mode[j] != 0 && (!job_nom || !xnom || job_nom[j] < 0 || job_nom[j] >= nnom))
            return set_err(c, KPILQR_ERR_ARG, "one-sided FD job without a valid nominal state");
    }
    if ((size_t)njobs > c->job_cap) {
        KP_HIP(c, hipStreamSynchronize(c->stream));
        const size_t cap = (size_t)njobs + (size_t)njobs / 8 + 64;
        int rc;
        if ((rc = regrow(c, &c->job_b, cap)) || (rc = regrow(c, &c->job_t, cap)) || (rc = regrow(c, &c->job_col, cap)) ||
            (rc = regrow(c, &c->job_nom, cap)) || (rc = regrow(c, &c->job_mode, cap)) ||
            (rc = regrow(c, &c->xplus, cap * n)) || (rc = regrow(c, &c->xminus, cap * n)))
            return rc;
        c->job_cap = cap;
    }
    if ((size_t)nnom > c->nom_cap || !c->xnom) {
        KP_HIP(c, hipStreamSynchronize(c->stream));
        const size_t cap = (size_t)nnom + (size_t)nnom / 8 + 64;
        int rc;
        if ((rc = regrow(c, &c->xnom, cap * n))) return rc;
        c->nom_cap = cap;
    }
    // slot table: maximal runs of consecutive jobs with the same (trajectory, time)
    std::vector<int> slots;
    slots.reserve((size_t)njobs / 8 + 2);
    for (int j = 0; j < njobs; j++)
        if (j == 0 || job_b[j] != job_b[j - 1] || job_t[j] != job_t[j - 1]) slots.push_back(j);
    const int nslots = (int)slots.size();
    slots.push_back(njobs);
    if ((size_t)nslots + 1 > c->slot_cap) {
        KP_HIP(c, hipStreamSynchronize(c->stream));
        const size_t cap = (size_t)nslots + (size_t)nslots / 8 + 64;
        int rc = regrow(c, &c->slot_start, cap + 1);
        if (rc) return rc;
        c->slot_cap = cap + 1;
    }
    KP_HIP(c, hipMemcpyAsync(c->slot_start, slots.data(), ((size_t)nslots + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));      // `slots` is a local, pageable vector
    c->nslots = nslots;
    const size_t J = njobs;
    if (njobs) {
        KP_HIP(c, hipMemcpyAsync(c->job_b, job_b, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->job_t, job_t, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->job_col, job_col, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->job_mode, job_mode, J, hipMemcpyHostToDevice, c->stream));
        if (job_nom) KP_HIP(c, hipMemcpyAsync(c->job_nom, job_nom, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        else KP_HIP(c, hipMemsetAsync(c->job_nom, 0, J * sizeof(int), c->stream));
        KP_HIP(c, hipMemcpyAsync(c->xplus, xplus, J * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->xminus, xminus, J * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (nnom) KP_HIP(c, hipMemcpyAsync(c->xnom, xnom, (size_t)nnom * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->njobs = njobs; c->nnom = nnom; c->eps = eps;
    return KPILQR_OK;
}

int kpilqr_fd_difference(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    KP_HIP(c, launch_fd_difference(c));
    return KPILQR_OK;
}

int kpilqr_interpolate(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_interpolate before kpilqr_set_keypoints");
    KP_HIP(c, launch_interpolate(c));
    return KPILQR_OK;
}

// Optimiser::FilterDynamicsMatrices (Optimiser.cpp:340-406) on the materialised A sequence
int kpilqr_filter_dynamics(kpilqr_ctx *c, const char *method, const double *coefs, int ncoef)
{
    if (!c || !method || !coefs) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    int mth = strcmp(method, "low_pass") == 0 ? 0 : strcmp(method, "FIR") == 0 ? 1 : -1;
    if (mth < 0) return set_err(c, KPILQR_ERR_ARG, "Filtering method not recognised (low_pass, FIR)");
    if (ncoef < 1 || ncoef > 16) return set_err(c, KPILQR_ERR_ARG, "1..16 filter coefficients");
    if (c->fused)
        return set_err(c, KPILQR_ERR_STATE, "the A filters act on the materialised sequence: create the context without KPILQR_FLAG_FUSED");
    int rc = ensure_stage(c, 16 * sizeof(double));
    if (rc) return rc;
    KP_HIP(c, hipMemcpyAsync(c->stage, coefs, sizeof(double) * ncoef, hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    KP_HIP(c, launch_filter_dynamics(c, mth, c->stage, ncoef));
    return KPILQR_OK;
}

// ---- STEP 1c ------------------------------------------------------------------------------------
int kpilqr_upload_residuals(kpilqr_ctx *c, const double *r, const double *r_x, const double *r_u,
                            const double *w_run, const double *w_term)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t B = c->d.batch, T1 = c->d.T + 1, n = c->n, m = c->d.m, nr = c->d.nr;
    if (r) KP_HIP(c, hipMemcpyAsync(c->r, r, B * T1 * nr * 8, hipMemcpyHostToDevice, c->stream));
    if (r_x) KP_HIP(c, hipMemcpyAsync(c->r_x, r_x, B * T1 * nr * n * 8, hipMemcpyHostToDevice, c->stream));
    if (r_u) KP_HIP(c, hipMemcpyAsync(c->r_u, r_u, B * T1 * nr * m * 8, hipMemcpyHostToDevice, c->stream));
    if (w_run) KP_HIP(c, hipMemcpyAsync(c->w_run, w_run, nr * 8, hipMemcpyHostToDevice, c->stream));
    if (w_term) KP_HIP(c, hipMemcpyAsync(c->w_term, w_term, nr * 8, hipMemcpyHostToDevice, c->stream));
    return KPILQR_OK;
}

int kpilqr_cost_derivs(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    KP_HIP(c, launch_cost_derivs(c));
    return KPILQR_OK;
}

int kpilqr_trajectory_cost(kpilqr_ctx *c, double *cost)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    KP_HIP(c, launch_trajectory_cost(c));
    if (cost) KP_HIP(c, hipMemcpyAsync(cost, c->traj_cost, (size_t)c->d.batch * 8, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// ---- STEP 2 -------------------------------------------------------------------------------------
static int check_fused(kpilqr_ctx *c)
{
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "fused sweeps need kpilqr_set_keypoints first");
    if (!c->kp_canonical)
        return set_err(c, KPILQR_ERR_STATE, "fused sweeps need canonical key-points (per DoF: strictly increasing, first 0, last T-1)");
    return KPILQR_OK;
}

static int run_backward(kpilqr_ctx *c, int pd_stride)
{
    if (c->fused) {
        int rc = check_fused(c);
        if (rc) return rc;
        KP_HIP(c, launch_backward_fused(c, pd_stride));
        return KPILQR_OK;
    }
    if (strcmp(c->bwd_variant, "mfma_f64_t1") == 0) KP_HIP(c, launch_backward_mfma(c, pd_stride));
    else if (strncmp(c->bwd_variant, "mfma_f64_tiled", 14) == 0) KP_HIP(c, launch_backward_tiled(c, pd_stride));
    else KP_HIP(c, launch_backward_generic(c, pd_stride));
    return KPILQR_OK;
}

int kpilqr_backward(kpilqr_ctx *c, const double *lambda, int pd_check_stride, int *status, double *delta_J)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (pd_check_stride < 1) return set_err(c, KPILQR_ERR_ARG, "pd_check_stride must be >= 1");
    if (lambda) KP_HIP(c, hipMemcpyAsync(c->lambda, lambda, (size_t)c->d.batch * 8, hipMemcpyHostToDevice, c->stream));
    int rc = run_backward(c, pd_check_stride);
    if (rc) return rc;
    if (status) KP_HIP(c, hipMemcpyAsync(status, c->status, (size_t)c->d.batch * 4, hipMemcpyDeviceToHost, c->stream));
    if (delta_J) KP_HIP(c, hipMemcpyAsync(delta_J, c->delta_J, (size_t)c->d.batch * 8, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

int kpilqr_download_gains(kpilqr_ctx *c, double *K, double *k)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t B = c->d.batch, T = c->d.T, n = c->n, m = c->d.m;
    if (K) KP_HIP(c, hipMemcpyAsync(K, c->K, B * T * n * m * 8, hipMemcpyDeviceToHost, c->stream));
    if (k) KP_HIP(c, hipMemcpyAsync(k, c->k, B * T * m * 8, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// iLQR_SVR::LeastImportantDofs, summing branch (iLQR_SVR.cpp:952-968), over the gains of the last backward pass
int kpilqr_dof_importance(kpilqr_ctx *c, int sampling_k_interval, double *sums)
{
    if (!c || !sums) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (sampling_k_interval < 1) return set_err(c, KPILQR_ERR_ARG, "sampling_k_interval must be >= 1");
    const size_t bytes = (size_t)c->d.batch * c->d.dof * sizeof(double);
    int rc = ensure_stage(c, bytes);
    if (rc) return rc;
    KP_HIP(c, launch_dof_importance(c, sampling_k_interval, c->stage));
    KP_HIP(c, hipMemcpyAsync(sums, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// ---- STEP 3 -------------------------------------------------------------------------------------
int kpilqr_upload_nominal(kpilqr_ctx *c, const double *u_nom, const double *ctrl_lim)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t B = c->d.batch, T = c->d.T, m = c->d.m;
    if (u_nom) KP_HIP(c, hipMemcpyAsync(c->u_nom, u_nom, B * T * m * 8, hipMemcpyHostToDevice, c->stream));
    if (ctrl_lim) KP_HIP(c, hipMemcpyAsync(c->ctrl_lim, ctrl_lim, 2 * m * 8, hipMemcpyHostToDevice, c->stream));
    return KPILQR_OK;
}

static int ensure_stage(kpilqr_ctx *c, size_t bytes)
{
    if (bytes <= c->stage_cap) return KPILQR_OK;
    KP_HIP(c, hipStreamSynchronize(c->stream));
    if (c->stage) KP_HIP(c, hipFree(c->stage));
    c->stage = nullptr; c->stage_cap = 0;
    KP_HIP(c, hipMalloc((void **)&c->stage, bytes));
    c->stage_cap = bytes;
    return KPILQR_OK;
}

static int run_forward(kpilqr_ctx *c, double *U_dev)
{
    if (c->fused) {
        int rc = check_fused(c);
        if (rc) return rc;
        KP_HIP(c, launch_forward_fused(c, U_dev));
        return KPILQR_OK;
    }
    if (strcmp(c->fwd_variant, "mfma_f64_t1") == 0) KP_HIP(c, launch_forward_mfma(c, U_dev));
    else if (strncmp(c->fwd_variant, "mfma_f64_tiled", 14) == 0) KP_HIP(c, launch_forward_tiled(c, U_dev));
    else KP_HIP(c, launch_forward_generic(c, U_dev));
    return KPILQR_OK;
}

int kpilqr_forward_linear(kpilqr_ctx *c, const double *alphas, double *cost_pred, double *U_alpha)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t B = c->d.batch, T = c->d.T, m = c->d.m, na = c->d.n_alpha;
    if (alphas) KP_HIP(c, hipMemcpyAsync(c->alphas, alphas, na * 8, hipMemcpyHostToDevice, c->stream));
    double *U_dev = nullptr;
    if (U_alpha) {
        int rc = ensure_stage(c, B * na * T * m * 8);
        if (rc) return rc;
        U_dev = c->stage;
    }
    int rc = run_forward(c, U_dev);
    if (rc) return rc;
    if (cost_pred) KP_HIP(c, hipMemcpyAsync(cost_pred, c->cost_pred, B * na * 8, hipMemcpyDeviceToHost, c->stream));
    if (U_alpha) KP_HIP(c, hipMemcpyAsync(U_alpha, U_dev, B * na * T * m * 8, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// ---- whole iteration ------------------------------------------------------------------------------
int kpilqr_iterate(kpilqr_ctx *c, const double *lambda, int pd_check_stride, const double *alphas)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_iterate before kpilqr_set_keypoints");
    if (pd_check_stride < 1) return set_err(c, KPILQR_ERR_ARG, "pd_check_stride must be >= 1");
    if (lambda) KP_HIP(c, hipMemcpyAsync(c->lambda, lambda, (size_t)c->d.batch * 8, hipMemcpyHostToDevice, c->stream));
    if (alphas) KP_HIP(c, hipMemcpyAsync(c->alphas, alphas, (size_t)c->d.n_alpha * 8, hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, launch_fd_difference(c));
    if (!c->fused) {              // the fused sweeps interpolate A,B and form l_* themselves
        KP_HIP(c, launch_interpolate(c));
        if (!c->tiled_a6) KP_HIP(c, launch_cost_derivs(c));      // tiled + flag: l_* are formed inside the sweeps
    }
    int rc = run_backward(c, pd_check_stride);
    if (rc) return rc;
    return run_forward(c, nullptr);
}

// ---- multi-GPU: the line-search cost reduction -----------------------------------------------------------
int kpilqr_comm_unique_id(char id[128])
{
    if (!id) return KPILQR_ERR_ARG;
    if (const char *e = comm_unique_id(id)) return set_err(nullptr, KPILQR_ERR_HIP, std::string("RCCL: ") + e);
    return KPILQR_OK;
}

int kpilqr_comm_init(kpilqr_ctx *c, int nranks, int rank, const char id[128])
{
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return KPILQR_ERR_ARG;
    if (c->comm) return set_err(c, KPILQR_ERR_STATE, "communicator already initialised");
    KP_HIP(c, hipSetDevice(c->d.device));
    if (const char *e = comm_init(c, nranks, rank, id)) return set_err(c, KPILQR_ERR_HIP, std::string("RCCL: ") + e);
    return KPILQR_OK;
}

int kpilqr_allreduce_linesearch(kpilqr_ctx *c, double vec8[8])
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    if (!c->ls8) KP_HIP(c, hipMalloc((void **)&c->ls8, 8 * sizeof(double)));
    KP_HIP(c, launch_pack_linesearch(c, c->ls8));
    if (const char *e = comm_allreduce8(c, c->ls8)) return set_err(c, KPILQR_ERR_HIP, std::string("RCCL: ") + e);
    if (vec8) KP_HIP(c, hipMemcpyAsync(vec8, c->ls8, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// ---- debug / oracle hooks -------------------------------------------------------------------------
int kpilqr_set_AB(kpilqr_ctx *c, const double *A, const double *B)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t BT = (size_t)c->d.batch * c->d.T, n = c->n, m = c->d.m;
    const size_t szA = BT * n * n * 8, szB = BT * n * m * 8;
    int rc = ensure_stage(c, szA + szB);
    if (rc) return rc;
    double *dA = c->stage, *dB = (double *)((char *)c->stage + szA);
    if (A) KP_HIP(c, hipMemcpyAsync(dA, A, szA, hipMemcpyHostToDevice, c->stream));
    if (B) KP_HIP(c, hipMemcpyAsync(dB, B, szB, hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, launch_pack_AB(c, A ? dA : nullptr, B ? dB : nullptr));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_get_AB(kpilqr_ctx *c, double *A, double *B)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t BT = (size_t)c->d.batch * c->d.T, n = c->n, m = c->d.m;
    const size_t szA = BT * n * n * 8, szB = BT * n * m * 8;
    int rc = ensure_stage(c, szA + szB);
    if (rc) return rc;
    double *dA = c->stage, *dB = (double *)((char *)c->stage + szA);
    KP_HIP(c, launch_unpack_AB(c, A ? dA : nullptr, B ? dB : nullptr));
    if (A) KP_HIP(c, hipMemcpyAsync(A, dA, szA, hipMemcpyDeviceToHost, c->stream));
    if (B) KP_HIP(c, hipMemcpyAsync(B, dB, szB, hipMemcpyDeviceToHost, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_set_cost_derivs(kpilqr_ctx *c, const double *l_x, const double *l_xx, const double *l_u, const double *l_uu)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t BT = (size_t)c->d.batch * c->d.T, n = c->n, m = c->d.m;
    const size_t s1 = BT * n * 8, s2 = BT * n * n * 8, s3 = BT * m * 8, s4 = BT * m * m * 8;
    int rc = ensure_stage(c, s1 + s2 + s3 + s4);
    if (rc) return rc;
    char *base = (char *)c->stage;
    double *d1 = (double *)base, *d2 = (double *)(base + s1), *d3 = (double *)(base + s1 + s2), *d4 = (double *)(base + s1 + s2 + s3);
    if (l_x) KP_HIP(c, hipMemcpyAsync(d1, l_x, s1, hipMemcpyHostToDevice, c->stream));
    if (l_xx) KP_HIP(c, hipMemcpyAsync(d2, l_xx, s2, hipMemcpyHostToDevice, c->stream));
    if (l_u) KP_HIP(c, hipMemcpyAsync(d3, l_u, s3, hipMemcpyHostToDevice, c->stream));
    if (l_uu) KP_HIP(c, hipMemcpyAsync(d4, l_uu, s4, hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, launch_pack_cost(c, l_x ? d1 : nullptr, l_xx ? d2 : nullptr, l_u ? d3 : nullptr, l_uu ? d4 : nullptr));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_get_cost_derivs(kpilqr_ctx *c, double *l_x, double *l_xx, double *l_u, double *l_uu)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));          // one context = one device; the caller may drive several
    const size_t BT = (size_t)c->d.batch * c->d.T, n = c->n, m = c->d.m;
    const size_t s1 = BT * n * 8, s2 = BT * n * n * 8, s3 = BT * m * 8, s4 = BT * m * m * 8;
    int rc = ensure_stage(c, s1 + s2 + s3 + s4);
    if (rc) return rc;
    char *base = (char *)c->stage;
    double *d1 = (double *)base, *d2 = (double *)(base + s1), *d3 = (double *)(base + s1 + s2), *d4 = (double *)(base + s1 + s2 + s3);
    KP_HIP(c, launch_unpack_cost(c, l_x ? d1 : nullptr, l_xx ? d2 : nullptr, l_u ? d3 : nullptr, l_uu ? d4 : nullptr));
    if (l_x) KP_HIP(c, hipMemcpyAsync(l_x, d1, s1, hipMemcpyDeviceToHost, c->stream));
    if (l_xx) KP_HIP(c, hipMemcpyAsync(l_xx, d2, s2, hipMemcpyDeviceToHost, c->stream));
    if (l_u) KP_HIP(c, hipMemcpyAsync(l_u, d3, s3, hipMemcpyDeviceToHost, c->stream));
    if (l_uu) KP_HIP(c, hipMemcpyAsync(l_uu, d4, s4, hipMemcpyDeviceToHost, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

const char *kpilqr_backward_variant(kpilqr_ctx *c) { return c ? c->bwd_variant : ""; }
const char *kpilqr_forward_variant(kpilqr_ctx *c) { return c ? c->fwd_variant : ""; }

}  // extern "C"
