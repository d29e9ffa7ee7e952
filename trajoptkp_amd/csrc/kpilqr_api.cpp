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

// memory the DMA engines can read in place: an async copy from it never has to be waited for before the call returns
static bool is_pinned(const void *p)
{
    if (!p) return true;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// chunk streams of kpilqr_iterate_streamed -> the context's stream: everything enqueued by a streamed iteration is
// ordered before whatever the caller enqueues next
static int join_pipeline(kpilqr_ctx *c)
{
    if (!c->pipe_dirty) return KPILQR_OK;
    for (int i = 0; i < Ctx::kPipeStreams; i++) {
        KP_HIP(c, hipEventRecord(c->pipe_done[i], c->pipe_stream[i]));
        KP_HIP(c, hipStreamWaitEvent(c->stream, c->pipe_done[i], 0));
    }
    c->pipe_dirty = false;
    return KPILQR_OK;
}

// every enqueueing entry point: select the context's device (several contexts per host thread), join the chunk streams
#define KP_ENTER(c)                                                   \
    do {                                                              \
        KP_HIP(c, hipSetDevice((c)->d.device));                       \
        if ((c)->pipe_dirty) { int rcj_ = join_pipeline(c); if (rcj_) return rcj_; } \
    } while (0)

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
    t.fwd_ragged_pair = env_int("KPILQR_FWD_RAGGED_PAIR", 0);
    t.fused_fwd_waves = env_int("KPILQR_FUSED_FWD_WAVES", 0);
    t.role_shift = env_int("KPILQR_ROLE_SHIFT", 9);
    t.tiled_nt_min = env_int("KPILQR_TILED_NT_MIN", 0);
    t.tiled_a6 = env_int("KPILQR_TILED_A6", -1);
    t.tiled_uw = env_int("KPILQR_TILED_UW", -1);
    t.tiled_fsc = env_int("KPILQR_TILED_FSC", -1);
    t.pipe_copy = env_int("KPILQR_PIPE_COPY", -1);
    t.fused_raw = env_int("KPILQR_FUSED_RAW", -1);
    t.fused_uni = env_int("KPILQR_FUSED_UNI", -1);
    return t;
}
}  // namespace kpilqr

// (Re)sizes every dimension-dependent device buffer of the context for c->d: a buffer is re-allocated only when it has to
// GROW (capacities are remembered), so shrinking the state vector and growing it back -- what iLQR_SVR does between
// optimisations -- re-uses the allocations.  Records, gains, residual and nominal-control buffers are zeroed.
static int size_buffers(kpilqr_ctx *c)
{
    const kpilqr_dims *dims = &c->d;
    c->n = 2 * dims->dof;
    c->L = RecLayout(c->n, dims->m);
    const size_t B = dims->batch, T = dims->T, n = c->n, m = dims->m, nr = dims->nr;
    struct Want { void **p; size_t bytes; size_t *cap; bool zero; };
    const Want want[] = {
        // a fused (one-tile) context keeps key-point columns only (Ctx::kpc); its records appear on demand (ensure_records)
        {(void **)&c->rec, c->fused ? 0 : B * T * c->L.stride * 8, &c->cap[0], true},
        {(void **)&c->K, B * T * n * m * 8, &c->cap[1], true},
        {(void **)&c->k, B * T * m * 8, &c->cap[2], true},
        {(void **)&c->r, B * (T + 1) * nr * 8, &c->cap[3], true},
        {(void **)&c->r_x, B * (T + 1) * nr * n * 8, &c->cap[4], true},
        {(void **)&c->r_u, B * (T + 1) * nr * m * 8, &c->cap[5], true},
        {(void **)&c->w_run, nr * 8, &c->cap[6], false},
        {(void **)&c->w_term, nr * 8, &c->cap[7], false},
        {(void **)&c->u_nom, B * T * m * 8, &c->cap[8], true},
        {(void **)&c->ctrl_lim, 2 * m * 8, &c->cap[9], false},
        {(void **)&c->lambda, B * 8, &c->cap[10], false},
        {(void **)&c->alphas, (size_t)dims->n_alpha * 8, &c->cap[11], false},
        {(void **)&c->cost_pred, B * dims->n_alpha * 8, &c->cap[12], false},
        {(void **)&c->delta_J, B * 8, &c->cap[13], false},
        {(void **)&c->traj_cost, B * 8, &c->cap[14], false},
        {(void **)&c->status, B * 4, &c->cap[15], true},
        {(void **)&c->segmap, B * dims->dof * T * sizeof(int2), &c->cap[16], false},
        {(void **)&c->kp_offsets, (B * dims->dof + 1) * 4, &c->cap[17], false},
    };
    for (const Want &w : want) {
        if (w.bytes > *w.cap) {
            if (*w.p) KP_HIP(c, hipFree(*w.p));
            *w.p = nullptr; *w.cap = 0;
            hipError_t e = hipMalloc(w.p, w.bytes ? w.bytes : 8);
            if (e != hipSuccess) { c->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return KPILQR_ERR_ALLOC; }
            *w.cap = w.bytes;
        }
        // records start zeroed so that padding / never-written columns are defined
        if (w.zero && w.bytes) KP_HIP(c, hipMemsetAsync(*w.p, 0, w.bytes, c->stream));
    }
    c->have_rec = !c->fused;
    c->rec_fd_base = c->rec;
    c->fd_batch_total = dims->batch;
    return KPILQR_OK;
}

// ---- fused contexts: key-point column store, entry tables, records on demand ----------------------------------------------
static int grow_dev(kpilqr_ctx *c, void **p, size_t *cap, size_t bytes, bool zero)
{
    if (bytes > *cap) {
        KP_HIP(c, hipStreamSynchronize(c->stream));              // nothing in flight still uses the old allocation
        if (*p) KP_HIP(c, hipFree(*p));
        *p = nullptr; *cap = 0;
        hipError_t e = hipMalloc(p, bytes ? bytes : 8);
        if (e != hipSuccess) { c->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return KPILQR_ERR_ALLOC; }
        *cap = bytes;
        if (zero) KP_HIP(c, hipMemsetAsync(*p, 0, bytes, c->stream));
        return 1;                                                // re-allocated: the old contents are gone
    }
    return KPILQR_OK;
}

// number of CSR entries of the current lists (known to the host since kpilqr_set_keypoints / kpilqr_generate_keypoints,
// which reads the total back); the capacity of kp_times before any key-points exist
static size_t kp_entries(const kpilqr_ctx *c)
{
    return c->kp_total_host >= 0 ? (size_t)c->kp_total_host : c->kp_cap;
}

// kpc for the current key-points: 3n doubles per CSR entry (a quarter of slack, so that lists whose counts move a little
// from one linearisation to the next -- the adaptive methods -- do not re-allocate every time)
static int ensure_kpc(kpilqr_ctx *c)
{
    const size_t need = kp_entries(c) * 3 * (size_t)c->n * 8;
    if (need <= c->kpc_cap) return KPILQR_OK;
    const int rc = grow_dev(c, (void **)&c->kpc, &c->kpc_cap, need + need / 4 + 4096, true);
    if (rc < 0) return rc;
    if (rc > 0) c->kpc_valid = c->kpc_touched = c->kps_valid = false;
    return KPILQR_OK;
}

// the slope store beside kpc, same size -- only when the lists may be per-DoF (ragged): uniform sets never read it
static int ensure_kps(kpilqr_ctx *c, bool force = false)
{
    if (c->kp_known_uniform && !force) return KPILQR_OK;
    const size_t need = kp_entries(c) * 6 * (size_t)c->n * 8;              // (value, slope) pairs
    if (need <= c->kps_cap) return KPILQR_OK;
    const int rc = grow_dev(c, (void **)&c->kps, &c->kps_cap, need + need / 4 + 4096, false);
    if (rc < 0) return rc;
    c->kps_valid = false;
    return KPILQR_OK;
}

// the slopes of the columns kpc holds (per-DoF lists: the general forms of the one-wave sweeps walk them; k_kp_slopes leaves at
// once when the device says the set is uniform)
static int slopes_for_kpc(kpilqr_ctx *c)
{
    if (c->kps_valid || c->kp_known_uniform || !c->kps) return KPILQR_OK;
    KP_HIP(c, launch_kp_slopes(c, true));
    c->kps_valid = true;
    return KPILQR_OK;
}

// kp_entry [list][t] and kp_entry_list [entry] for the current lists
static int ensure_entry_tables(kpilqr_ctx *c)
{
    if (c->entry_tables_valid) return KPILQR_OK;
    int rc = grow_dev(c, (void **)&c->kp_entry, &c->kp_entry_cap, (size_t)c->d.batch * c->d.dof * c->d.T * sizeof(int), false);
    if (rc < 0) return rc;
    {
        const size_t need = (kp_entries(c) ? kp_entries(c) : 1) * sizeof(int);
        rc = need <= c->kp_entry_list_cap ? KPILQR_OK : grow_dev(c, (void **)&c->kp_entry_list, &c->kp_entry_list_cap, need + need / 4, false);
    }
    if (rc < 0) return rc;
    KP_HIP(c, launch_build_entry_tables(c));
    c->entry_tables_valid = true;
    return KPILQR_OK;
}

// The resident FD payload differenced into kpc (explicitly: the raw backward sweep does the same on the fly)
static int difference_to_kpc(kpilqr_ctx *c)
{
    if (c->fd_kind == 0 || !c->have_kp) return KPILQR_OK;
    if (c->fd_kind == 3) {                       // the columns ARE the payload
        if (!c->kpc_valid) return set_err(c, KPILQR_ERR_STATE, "the key-point columns are gone (new key-points): upload them again");
        return KPILQR_OK;
    }
    int rc = ensure_kpc(c);
    if (rc) return rc;
    if (c->fused) { rc = ensure_kps(c); if (rc) return rc; }       // (only the fused sweeps' per-DoF list forms read the slope store)
    c->kps_valid = false;
    if (c->fd_kind == 1) {
        rc = ensure_entry_tables(c);
        if (rc) return rc;
        KP_HIP(c, launch_fd_difference_kpc(c));
    } else {
        KP_HIP(c, launch_fd_kp_difference(c));
        if (c->kps && !c->kp_known_uniform) c->kps_valid = true;         // (the slope store of per-DoF lists is written in the same pass)
    }
    c->kpc_valid = true;
    return KPILQR_OK;
}

// The key-point columns of the resident FD payload written into the step records (what kpilqr_fd_difference means on a
// context that has records)
static int records_from_payload(kpilqr_ctx *c)
{
    if (c->fd_kind == 1) { KP_HIP(c, launch_fd_difference(c)); return KPILQR_OK; }
    if (c->fd_kind == 2 || c->fd_kind == 3) {
        if (!c->have_kp) return KPILQR_OK;
        int rc = KPILQR_OK;
        if (!c->kpc_valid) rc = difference_to_kpc(c);
        if (rc) return rc;
        rc = ensure_entry_tables(c);
        if (rc) return rc;
        KP_HIP(c, launch_kpc_to_records(c));
    }
    return KPILQR_OK;
}

// A fused context has no step records until something asks for the materialised sequence (kpilqr_interpolate, get_AB /
// set_AB, the cost-derivative hooks, the key-point error test, KPILQR_BUF_STEP_RECORDS): then they are allocated, zeroed
// and given the key-point columns of the resident payload.
static int ensure_records(kpilqr_ctx *c)
{
    if (!c->have_rec) {
        const size_t bytes = (size_t)c->d.batch * c->d.T * c->L.stride * 8;
        const int rc = grow_dev(c, (void **)&c->rec, &c->cap[0], bytes, false);
        if (rc < 0) return rc;
        KP_HIP(c, hipMemsetAsync(c->rec, 0, bytes, c->stream));
        c->rec_fd_base = c->rec;
        c->have_rec = true;
        c->rec_synced = false;
    }
    if (!c->rec_synced) {
        const int rc = records_from_payload(c);
        if (rc) return rc;
        c->rec_synced = true;
    }
    return KPILQR_OK;
}

// a new FD payload or new key-points: whatever was derived from the old ones is stale
static void payload_changed(kpilqr_ctx *c)
{
    c->kpc_valid = c->kpc_touched = c->kps_valid = false;
    c->rec_synced = false;
}

// Constant residual Jacobians (kpilqr_upload_residual_jacobians_const): the one-wave fused sweeps keep r_x in registers; every
// other kernel family streams r_x per step from the context's buffer, which then receives the broadcast copy -- once, on demand.
static int ensure_rx_buffer(kpilqr_ctx *c)
{
    if (!c->rx_const_on || c->rx_buf_valid) return KPILQR_OK;
    KP_HIP(c, launch_broadcast_rx(c));
    c->rx_buf_valid = true;
    return KPILQR_OK;
}
// whether the sweep about to be launched reads the r_x buffer (every form but the one-wave fused instantiations without r_u)
static bool backward_reads_rx_buffer(const kpilqr_ctx *c)
{
    const int form = c->fused ? backward_fused_form(c) : 0;
    return !((form == 1 || form == 5) && c->ru_zero);       // (the forms with a constant-Jacobian instantiation)
}
static bool forward_reads_rx_buffer(const kpilqr_ctx *c) { return !(c->fused && c->ru_zero); }      // (every fused forward form has its constant-Jacobian instantiation)

// kernel families for c->d (names: kpilqr_backward_variant)
static int select_variants(kpilqr_ctx *c)
{
    const kpilqr_dims *dims = &c->d;
    c->fused = c->tiled_a6 = false;
    const bool generic = (dims->flags & KPILQR_FLAG_GENERIC_KERNELS) != 0;
    const bool force_tiled = (dims->flags & KPILQR_FLAG_TILED_KERNELS) != 0;
    c->bwd_variant = (!generic && !force_tiled && backward_mfma_supported(c->n, dims->m)) ? "mfma_f64_t1"
                   : (!generic && backward_tiled_supported(c->n, dims->m, c->tune.tiled_nt_min)) ? "mfma_f64_tiled"
                   : (!generic && backward_wide_supported(c->n, dims->m, c->tune.tiled_nt_min)) ? "mfma_f64_wide" : "generic_lds";
    c->fwd_variant = (!generic && !force_tiled && forward_mfma_supported(c->n, dims->m, dims->n_alpha)) ? "mfma_f64_t1"
                   : (!generic && forward_tiled_supported(c->n, dims->m, dims->n_alpha, c->tune.tiled_nt_min)) ? "mfma_f64_tiled"
                   : (!generic && forward_wide_supported(c->n, dims->m, dims->n_alpha, c->tune.tiled_nt_min)) ? "mfma_f64_wide" : "generic_lds";
    if ((dims->flags & KPILQR_FLAG_FUSED) && !generic && !force_tiled &&
        fused_supported(c->n, dims->m, dims->nr, dims->dof, dims->T, c->L.stride, dims->n_alpha)) {
        c->fused = true;
        c->bwd_variant = c->fwd_variant = "mfma_f64_t1_fused";
    }
    // The same flag on a tiled shape (n + 2 > 16): a6 (variant "..._a6"), cost derivatives formed from the residuals inside the
    // sweeps.  It replaces k_cost_derivs (HBM-bound: n^2 doubles written per step) by NT*ceil(nr/4) + 6 MFMAs per wave-step of the
    // latency-bound backward sweep (+9 % at four tiles, whatever the batch): a gain from ~100 trajectories of a four-tile state up
    // (n = 62, B = 128, T = 5000: 74.4 -> 70.6 ms), a loss for two or three tiles at the batches measured.  KPILQR_TILED_A6 = 0 | 1.
    // (a4 inside the tiled sweeps existed in rounds 2-4, parity-green and slower by more than the k_interpolate it removed; removed
    // in round 5: tiled_mfma.hip.)
    if ((dims->flags & KPILQR_FLAG_FUSED) && !c->fused &&
        strcmp(c->bwd_variant, "mfma_f64_tiled") == 0 && strcmp(c->fwd_variant, "mfma_f64_tiled") == 0) {
        const bool want_a6 = dims->nr <= 16 && (c->tune.tiled_a6 >= 0 ? c->tune.tiled_a6 != 0
                                                : (tiled_tiles(c->n, c->tune.tiled_nt_min) == 4 && dims->batch >= 96));
        c->tiled_a6 = want_a6;
        if (want_a6) c->bwd_variant = c->fwd_variant = "mfma_f64_tiled_a6";
    }
    if (strcmp(c->bwd_variant, "generic_lds") == 0 && backward_generic_lds_bytes(c->n, dims->m) > 160 * 1024) {
        c->err = "state dimension too large for the generic backward kernel (LDS)";
        return KPILQR_ERR_ARG;
    }
    return KPILQR_OK;
}

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

    {
        int rcs = select_variants(c);                        // first: a fused context allocates no step records
        if (rcs == KPILQR_OK) rcs = size_buffers(c);
        if (rcs != KPILQR_OK) { const std::string msg = c->err; kpilqr_destroy(c); return set_err(nullptr, rcs, msg); }
    }
    hipError_t rc = hipSuccess;
#define TRY(x) do { if (rc == hipSuccess) rc = (x); } while (0)
    TRY(dalloc(&c->err_flag, 1));
    TRY(dalloc(&c->kp_uniform, 1));
    TRY(hipHostMalloc((void **)&c->err_flag_host, sizeof(int), hipHostMallocDefault));
#undef TRY
    if (rc != hipSuccess) {
        std::string msg = std::string("hipMalloc failed: ") + hipGetErrorString(rc);
        kpilqr_destroy(c);
        return set_err(nullptr, KPILQR_ERR_ALLOC, msg);
    }
    (void)hipMemsetAsync(c->err_flag, 0, sizeof(int), c->stream);
    (void)hipMemsetAsync(c->kp_uniform, 0, sizeof(int), c->stream);
    *out = c;
    return KPILQR_OK;
}

void kpilqr_destroy(kpilqr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->d.device);
    (void)hipStreamSynchronize(c->stream);
    if (c->pipe_ready) {
        for (int i = 0; i < Ctx::kPipeStreams; i++) {
            if (c->pipe_stream[i]) { (void)hipStreamSynchronize(c->pipe_stream[i]); (void)hipStreamDestroy(c->pipe_stream[i]); }
            if (c->pipe_done[i]) (void)hipEventDestroy(c->pipe_done[i]);
        }
        if (c->pipe_in) (void)hipEventDestroy(c->pipe_in);
    }
    comm_destroy(c);
    void *ptrs[] = {c->rec, c->K, c->k, c->r, c->r_x, c->r_u, c->w_run, c->w_term, c->u_nom, c->ctrl_lim,
                    c->lambda, c->alphas, c->cost_pred, c->delta_J, c->traj_cost, c->status, c->segmap,
                    c->kp_offsets, c->kp_times, c->X_states, c->kp_thr, c->kp_mask, c->kp_count, c->ls8, c->fd_dev,
                    c->stage, c->err_flag, c->kp_uniform, c->kpc, c->kp_entry, c->kp_entry_list, c->fdk_dev, c->rx_const, c->kps};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->err_flag_host) (void)hipHostFree(c->err_flag_host);
    if (c->kp_traj_first_host) free(c->kp_traj_first_host);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int kpilqr_get_dims(kpilqr_ctx *c, kpilqr_dims *out)
{
    if (!c || !out) return KPILQR_ERR_ARG;
    *out = c->d;
    return KPILQR_OK;
}

// iLQR_SVR::Resize (src/Optimiser/iLQR_SVR.cpp:38-193): the optimiser changes the size of its state vector between
// optimisations.  Re-sizes the context in place; everything uploaded before (key-points, FD payload, residuals, nominal
// controls, weights, limits, alphas, lambda) is forgotten, the kernel families are re-selected.
int kpilqr_resize(kpilqr_ctx *c, int new_dof, int new_num_ctrl, int new_horizon)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (new_dof < 1 || new_num_ctrl < 1 || new_horizon < 2) return set_err(c, KPILQR_ERR_ARG, "dims out of range");
    KP_HIP(c, hipStreamSynchronize(c->stream));          // nothing in flight still uses the old layout
    const kpilqr_dims old = c->d;
    c->d.dof = new_dof; c->d.m = new_num_ctrl; c->d.T = new_horizon;
    c->n = 2 * new_dof; c->L = RecLayout(c->n, new_num_ctrl);
    int rc = select_variants(c);
    if (rc == KPILQR_OK) rc = size_buffers(c);
    if (rc != KPILQR_OK) {                              // leave a usable context behind
        const std::string msg = c->err;
        c->d = old;
        c->n = 2 * old.dof; c->L = RecLayout(c->n, old.m);
        (void)select_variants(c); (void)size_buffers(c);
        return set_err(c, rc, msg);
    }
    c->have_kp = c->kp_canonical = c->have_states = c->kp_known_uniform = false;
    c->njobs = c->nnom = 0;
    c->fd_kind = 0; c->fdk_entries = 0; c->entry_tables_valid = false; c->kp_total_host = -1;
    if (c->kp_traj_first_host) { free(c->kp_traj_first_host); c->kp_traj_first_host = nullptr; }
    payload_changed(c);
    c->ru_zero = true;                                   // size_buffers zeroed r_u
    c->rx_const_on = false; c->rx_buf_valid = true;
    if (c->X_states) { KP_HIP(c, hipFree(c->X_states)); c->X_states = nullptr; }
    if (c->kp_mask) { KP_HIP(c, hipFree(c->kp_mask)); c->kp_mask = nullptr; }
    if (c->kp_count) { KP_HIP(c, hipFree(c->kp_count)); c->kp_count = nullptr; }
    if (c->kp_thr) { KP_HIP(c, hipFree(c->kp_thr)); c->kp_thr = nullptr; }
    return KPILQR_OK;
}

int kpilqr_host_alloc(kpilqr_ctx *c, size_t bytes, void **pinned)
{
    if (!c || !pinned) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    KP_HIP(c, hipHostMalloc(pinned, bytes ? bytes : 1, hipHostMallocDefault));
    return KPILQR_OK;
}

int kpilqr_host_free(kpilqr_ctx *c, void *pinned)
{
    if (!c) {
        // the allocation may outlive the context it was made through (a binding whose arrays are still referenced after the
        // engine was closed): pinned host memory is not tied to a device or a stream
        if (pinned && hipHostFree(pinned) != hipSuccess) return set_err(nullptr, KPILQR_ERR_HIP, "hipHostFree failed");
        return KPILQR_OK;
    }
    KP_ENTER(c);
    if (pinned) KP_HIP(c, hipHostFree(pinned));
    return KPILQR_OK;
}

// Waits for the context's stream and reports what the device-side argument checks raised since the last report.  Used by
// kpilqr_sync and by every other entry point that synchronises (blocking downloads): a caller that never calls kpilqr_sync
// still sees a skipped FD job.
static int sync_and_report(kpilqr_ctx *c)
{
    KP_HIP(c, hipMemcpyAsync(c->err_flag_host, c->err_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    if (*c->err_flag_host) {
        const int bits = *c->err_flag_host;
        KP_HIP(c, hipMemsetAsync(c->err_flag, 0, sizeof(int), c->stream));
        return set_err(c, KPILQR_ERR_ARG, (bits & 1) ? "FD job index out of range (trajectory, time, column, mode or nominal row): the job was skipped"
                                                     : "device-side argument check failed");
    }
    return KPILQR_OK;
}

int kpilqr_sync(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    return sync_and_report(c);
}

int kpilqr_device_ptr(kpilqr_ctx *c, int which, void **dptr, size_t *bytes)
{
    if (!c || !dptr) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    const size_t B = c->d.batch, T = c->d.T, n = c->n, m = c->d.m, nr = c->d.nr;
    void *p = nullptr; size_t sz = 0;
    switch (which) {
    case KPILQR_BUF_STEP_RECORDS: { const int rcr = ensure_records(c); if (rcr) return rcr; } p = c->rec; sz = B * T * c->L.stride * 8; break;
    case KPILQR_BUF_K: p = c->K; sz = B * T * n * m * 8; break;
    case KPILQR_BUF_k: p = c->k; sz = B * T * m * 8; break;
    case KPILQR_BUF_RESIDUALS: p = c->r; sz = B * (T + 1) * nr * 8; break;
    case KPILQR_BUF_R_X:
        // a writable pointer leaves the library: the buffer gets the constant Jacobians' broadcast copy and is what the
        // sweeps read from here on (the constant mode is off, as for r_u below)
        { const int rcx = ensure_rx_buffer(c); if (rcx) return rcx; }
        c->rx_const_on = false; c->rx_buf_valid = true;
        p = c->r_x; sz = B * (T + 1) * nr * n * 8; break;
    case KPILQR_BUF_R_U:
        // a writable pointer leaves the library: from here on r_u may be non-zero without kpilqr_upload_residuals having
        // seen it, so the r_u-free instantiations of the fused sweeps (Ctx::ru_zero) are off for this context
        p = c->r_u; sz = B * (T + 1) * nr * m * 8; c->ru_zero = false; break;
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
    KP_ENTER(c);
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
    // every trajectory's DoFs share one list (set_interval): the device flag of k_kp_uniform will say the same, only the
    // segment-loop forms of the sweeps run and no slope store is needed
    bool uniform = true;
    for (int b = 0; b < c->d.batch && uniform; b++) {
        const int *o = kp_offsets + (size_t)b * c->d.dof;
        const int len0 = o[1] - o[0];
        for (int i = 1; i < c->d.dof && uniform; i++)
            uniform = (o[i + 1] - o[i] == len0) && memcmp(kp_times + o[i], kp_times + o[0], sizeof(int) * (size_t)len0) == 0;
    }
    c->kp_known_uniform = uniform && c->tune.fused_uni != 0;     // (KPILQR_FUSED_UNI=0: the general forms run on every set)
    if ((size_t)total > c->kp_cap) {
        if (c->kp_times) { KP_HIP(c, hipStreamSynchronize(c->stream)); KP_HIP(c, hipFree(c->kp_times)); c->kp_times = nullptr; }
        c->kp_cap = (size_t)total + (size_t)total / 4 + 64;
        KP_HIP(c, dalloc(&c->kp_times, c->kp_cap));
    }
    KP_HIP(c, hipMemcpyAsync(c->kp_offsets, kp_offsets, (nlists + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, hipMemcpyAsync(c->kp_times, kp_times, (size_t)total * sizeof(int), hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, launch_build_segmap(c));
    // pageable host arrays: make the copies complete before returning control (pinned ones are read in place)
    if (!(is_pinned(kp_offsets) && is_pinned(kp_times))) KP_HIP(c, hipStreamSynchronize(c->stream));
    c->have_kp = true;
    // host copy of the first CSR entry of every trajectory (chunking of a key-point ordered payload), and everything that
    // was derived from the old lists is stale; a key-point ordered payload is laid out BY the lists: it has to follow them
    c->kp_total_host = total;
    if (!c->kp_traj_first_host) c->kp_traj_first_host = (int *)malloc(sizeof(int) * ((size_t)c->d.batch + 1));
    if (!c->kp_traj_first_host) return set_err(c, KPILQR_ERR_ALLOC, "host allocation failed");
    for (int b = 0; b <= c->d.batch; b++) c->kp_traj_first_host[b] = kp_offsets[(size_t)b * c->d.dof];
    c->entry_tables_valid = false;
    if (c->fd_kind == 2 || c->fd_kind == 3) { c->fd_kind = 0; c->fdk_entries = 0; }
    payload_changed(c);
    return KPILQR_OK;
}

static int ensure_stage(kpilqr_ctx *c, size_t bytes);

// ---- key-point placement on the device ----------------------------------------------------------------
int kpilqr_upload_states(kpilqr_ctx *c, const double *X)
{
    if (!c || !X) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    const size_t count = (size_t)c->d.batch * c->d.T * c->n;
    if (!c->X_states) KP_HIP(c, hipMalloc((void **)&c->X_states, count * sizeof(double)));
    KP_HIP(c, hipMemcpyAsync(c->X_states, X, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->have_states = true;
    return KPILQR_OK;
}

int kpilqr_generate_keypoints(kpilqr_ctx *c, const char *method, int min_N, int max_N, const double *thresholds, double dt)
{
    if (!c || !method) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    int mth = -1;
    if (strcmp(method, "set_interval") == 0) mth = 0;
    else if (strcmp(method, "adaptive_jerk") == 0) mth = 1;
    else if (strcmp(method, "velocity_change") == 0) mth = 2;
    else if (strcmp(method, "adaptive_accel") == 0) mth = 3;
    else return set_err(c, KPILQR_ERR_ARG, "kpilqr_generate_keypoints: method must be set_interval, adaptive_jerk, adaptive_accel or velocity_change "
                                           "(iterative_error interleaves host finite differences and stays on the host)");
    if (min_N < 1 || max_N < 1) return set_err(c, KPILQR_ERR_ARG, "min_N and max_N must be >= 1");
    if (c->d.dof > 64) return set_err(c, KPILQR_ERR_ARG, "on-device key-point placement supports dof <= 64");
    if (mth != 0 && (!thresholds || (mth == 1 && !(dt > 0.0)))) return set_err(c, KPILQR_ERR_ARG, "thresholds (and, for adaptive_jerk, a positive dt) are required");
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
    c->kp_known_uniform = mth == 0 && c->tune.fused_uni != 0;    // set_interval: one list for all DoFs; the other methods place per DoF
    c->kp_canonical = true;      // rows 0 and T-1 are always full and the lists are strictly increasing by construction
    // The lists exist on the device only (kpilqr_get_keypoints brings them to the host), but their TOTAL is read back here --
    // one int: the column store and the entry tables are sized from it (not from the worst case batch * dof * T: 7.2 GB
    // against 1.4 GB at the headline shape), and the `entries` of a key-point ordered upload is checked against it.
    {
        int total = 0;
        KP_HIP(c, hipMemcpyAsync(&total, c->kp_offsets + nlists, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        KP_HIP(c, hipStreamSynchronize(c->stream));
        if (total < 0 || (size_t)total > nlists * T) return set_err(c, KPILQR_ERR_HIP, "kpilqr_generate_keypoints: implausible key-point count read back");
        c->kp_total_host = total;
    }
    if (c->kp_traj_first_host) { free(c->kp_traj_first_host); c->kp_traj_first_host = nullptr; }
    c->entry_tables_valid = false;
    if (c->fd_kind == 2 || c->fd_kind == 3) { c->fd_kind = 0; c->fdk_entries = 0; }
    payload_changed(c);
    return KPILQR_OK;
}

int kpilqr_keypoint_error_test(kpilqr_ctx *c, int n_iv, const int *intervals, int min_N, double threshold, unsigned char *good)
{
    if (!c || n_iv < 0 || (n_iv > 0 && (!intervals || !good))) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (n_iv == 0) return KPILQR_OK;
    const size_t iv_bytes = (size_t)n_iv * 4 * sizeof(int), off = (iv_bytes + 15) & ~(size_t)15;
    int rc = ensure_stage(c, off + (size_t)n_iv);
    if (rc) return rc;
    rc = ensure_records(c);                  // a fused context: records on demand, with the resident payload's columns
    if (rc) return rc;
    int *iv_dev = (int *)c->stage;
    unsigned char *good_dev = (unsigned char *)c->stage + off;
    KP_HIP(c, hipMemcpyAsync(iv_dev, intervals, iv_bytes, hipMemcpyHostToDevice, c->stream));
    KP_HIP(c, launch_kp_error_test(c, n_iv, iv_dev, min_N, threshold, good_dev));
    KP_HIP(c, hipMemcpyAsync(good, good_dev, (size_t)n_iv, hipMemcpyDeviceToHost, c->stream));
    { const int rcs = sync_and_report(c); if (rcs) return rcs; }
    for (int k = 0; k < n_iv; k++) if (good[k] > 1) return set_err(c, KPILQR_ERR_ARG, "kpilqr_keypoint_error_test: interval out of range");
    return KPILQR_OK;
}

int kpilqr_get_keypoints(kpilqr_ctx *c, int *kp_offsets, int *kp_times, int times_capacity)
{
    if (!c || !kp_offsets) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "no key-points set");
    const size_t nlists = (size_t)c->d.batch * c->d.dof;
    KP_HIP(c, hipMemcpyAsync(kp_offsets, c->kp_offsets, (nlists + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    KP_HIP(c, hipStreamSynchronize(c->stream));
    const int total = kp_offsets[nlists];
    c->kp_total_host = total;
    if (!c->kp_traj_first_host) c->kp_traj_first_host = (int *)malloc(sizeof(int) * ((size_t)c->d.batch + 1));
    if (c->kp_traj_first_host) for (int b = 0; b <= c->d.batch; b++) c->kp_traj_first_host[b] = kp_offsets[(size_t)b * c->d.dof];
    if (kp_times) {
        if (total > times_capacity) return set_err(c, KPILQR_ERR_ARG, "kp_times capacity too small");
        KP_HIP(c, hipMemcpyAsync(kp_times, c->kp_times, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        KP_HIP(c, hipStreamSynchronize(c->stream));
    }
    return total;
}

static size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

static void fd_layout(int n, int njobs, int nnom, kpilqr_fd_layout *L)
{
    const size_t J = (size_t)njobs, N = (size_t)nnom;
    size_t o = 0;
    L->xplus = o; o = al16(o + J * n * sizeof(double));
    L->xminus = o; o = al16(o + J * n * sizeof(double));
    L->xnom = o; o = al16(o + N * n * sizeof(double));
    L->job_b = o; o = al16(o + J * sizeof(int));
    L->job_t = o; o = al16(o + J * sizeof(int));
    L->job_col = o; o = al16(o + J * sizeof(int));
    L->job_nom = o; o = al16(o + J * sizeof(int));
    L->job_mode = o; o = al16(o + J);
    L->bytes = o;
}

int kpilqr_fd_slab_layout(kpilqr_ctx *c, int njobs, int nnom, kpilqr_fd_layout *out)
{
    if (!c || !out || njobs < 0 || nnom < 0) return KPILQR_ERR_ARG;
    fd_layout(c->n, njobs, nnom, out);
    return KPILQR_OK;
}

// device slab with the layout of (njobs, nnom); grows (with a stream sync) only when it has to
static int fd_bind(kpilqr_ctx *c, int njobs, int nnom, kpilqr_fd_layout *L)
{
    fd_layout(c->n, njobs, nnom, L);
    if (L->bytes > c->fd_dev_cap) {
        KP_HIP(c, hipStreamSynchronize(c->stream));
        if (c->fd_dev) KP_HIP(c, hipFree(c->fd_dev));
        c->fd_dev = nullptr; c->fd_dev_cap = 0;
        const size_t cap = L->bytes + L->bytes / 8 + 4096;
        KP_HIP(c, hipMalloc((void **)&c->fd_dev, cap));
        c->fd_dev_cap = cap;
    }
    char *base = c->fd_dev;
    c->xplus = (double *)(base + L->xplus); c->xminus = (double *)(base + L->xminus); c->xnom = (double *)(base + L->xnom);
    c->job_b = (int *)(base + L->job_b); c->job_t = (int *)(base + L->job_t); c->job_col = (int *)(base + L->job_col);
    c->job_nom = (int *)(base + L->job_nom);
    c->job_mode = (unsigned char *)(base + L->job_mode);
    return KPILQR_OK;
}

int kpilqr_upload_fd(kpilqr_ctx *c, int njobs, const int *job_b, const int *job_t, const int *job_col,
                     const unsigned char *job_mode, const int *job_nom, const double *xplus,
                     const double *xminus, int nnom, const double *xnom, double eps)
{
    if (!c || njobs < 0 || nnom < 0) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (njobs > 0 && (!job_b || !job_t || !job_col || !job_mode || !xplus || !xminus))
        return set_err(c, KPILQR_ERR_ARG, "null FD job array");
    if (nnom > 0 && !xnom) return set_err(c, KPILQR_ERR_ARG, "nnom > 0 without xnom");
    if (!(eps > 0.0)) return set_err(c, KPILQR_ERR_ARG, "eps must be positive");
    const int n = c->n;
    kpilqr_fd_layout L;
    int rc = fd_bind(c, njobs, nnom, &L);
    if (rc) return rc;
    const size_t J = njobs;
    if (njobs) {
        KP_HIP(c, hipMemcpyAsync(c->job_b, job_b, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->job_t, job_t, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->job_col, job_col, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->job_mode, job_mode, J, hipMemcpyHostToDevice, c->stream));
        // no nominal rows given: every job_nom becomes -1, so a one-sided job (mode 1, 2) fails the device-side range check
        // whatever nnom is, is skipped and reported by the next synchronising call
        if (job_nom) KP_HIP(c, hipMemcpyAsync(c->job_nom, job_nom, J * sizeof(int), hipMemcpyHostToDevice, c->stream));
        else KP_HIP(c, hipMemsetAsync(c->job_nom, 0xff, J * sizeof(int), c->stream));
        KP_HIP(c, hipMemcpyAsync(c->xplus, xplus, J * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, hipMemcpyAsync(c->xminus, xminus, J * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (nnom) KP_HIP(c, hipMemcpyAsync(c->xnom, xnom, (size_t)nnom * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->njobs = njobs; c->nnom = nnom; c->eps = eps;
    c->fd_kind = 1; payload_changed(c);
    // pageable sources: the caller may free them on return, so wait for the copies; pinned ones are read in place
    if (!(is_pinned(job_b) && is_pinned(job_t) && is_pinned(job_col) && is_pinned(job_mode) && is_pinned(job_nom) &&
          is_pinned(xplus) && is_pinned(xminus) && is_pinned(xnom)))
        KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_upload_fd_slab(kpilqr_ctx *c, const void *slab, int njobs, int nnom, double eps)
{
    if (!c || njobs < 0 || nnom < 0 || (njobs > 0 && !slab)) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (!(eps > 0.0)) return set_err(c, KPILQR_ERR_ARG, "eps must be positive");
    kpilqr_fd_layout L;
    int rc = fd_bind(c, njobs, nnom, &L);
    if (rc) return rc;
    if (njobs) KP_HIP(c, hipMemcpyAsync(c->fd_dev, slab, L.bytes, hipMemcpyHostToDevice, c->stream));   // the one DMA
    c->njobs = njobs; c->nnom = nnom; c->eps = eps;
    c->fd_kind = 1; payload_changed(c);
    if (!is_pinned(slab)) KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

// ---- key-point ordered FD payload ---------------------------------------------------------------------------------------
static void fdkp_layout(int n, int entries, kpilqr_fdkp_layout *L)
{
    L->entry_stride = (size_t)(6 * n + 2) * 8;
#if KP_RAW_PAIRS
    L->xplus = 0; L->xminus = 8; L->elem_stride = 16; L->mode = (size_t)6 * n * 8;       // (x+, x-) pairs, element by element
#else
    L->xplus = 0; L->xminus = (size_t)3 * n * 8; L->elem_stride = 8; L->mode = (size_t)6 * n * 8;
#endif
    L->bytes = (size_t)entries * L->entry_stride;
}

int kpilqr_fd_kp_layout(kpilqr_ctx *c, int entries, kpilqr_fdkp_layout *out)
{
    if (!c || !out || entries < 0) return KPILQR_ERR_ARG;
    fdkp_layout(c->n, entries, out);
    return KPILQR_OK;
}

// device slab of the key-point ordered payload for `entries` entries; grows (after a stream sync) only when it has to
static int fdk_bind(kpilqr_ctx *c, int entries, kpilqr_fdkp_layout *L)
{
    fdkp_layout(c->n, entries, L);
    const int rc = grow_dev(c, (void **)&c->fdk_dev, &c->fdk_dev_cap, L->bytes + L->bytes / 8 + 4096, false);
    if (rc < 0) return rc;
    return KPILQR_OK;
}

int kpilqr_upload_fd_kp(kpilqr_ctx *c, const void *slab, int entries, double eps)
{
    if (!c || entries < 0 || (entries > 0 && !slab)) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (!(eps > 0.0)) return set_err(c, KPILQR_ERR_ARG, "eps must be positive");
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_upload_fd_kp before the key-points it is ordered by (kpilqr_set_keypoints / kpilqr_generate_keypoints)");
    // (the sweeps index the slab by the device CSR, not by `entries`: a smaller slab would be read past its end)
    if (entries != c->kp_total_host)
        return set_err(c, KPILQR_ERR_ARG, "kpilqr_upload_fd_kp: `entries` is not the number of key-point entries (kp_offsets[batch*dof])");
    kpilqr_fdkp_layout L;
    int rc = fdk_bind(c, entries, &L);
    if (rc) return rc;
    if (entries) KP_HIP(c, hipMemcpyAsync(c->fdk_dev, slab, L.bytes, hipMemcpyHostToDevice, c->stream));   // the one DMA
    c->fdk_entries = entries; c->fdk_first = 0; c->eps = eps;
    c->fd_kind = 2; payload_changed(c);
    if (!is_pinned(slab)) KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

// The differenced key-point columns as the payload (fd_kind 3): straight into the column store
int kpilqr_upload_kp_columns(kpilqr_ctx *c, const double *columns, int entries)
{
    if (!c || entries < 0 || (entries > 0 && !columns)) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_upload_kp_columns before the key-points it is ordered by (kpilqr_set_keypoints / kpilqr_generate_keypoints)");
    if (entries != c->kp_total_host)
        return set_err(c, KPILQR_ERR_ARG, "kpilqr_upload_kp_columns: `entries` is not the number of key-point entries (kp_offsets[batch*dof])");
    int rc = ensure_kpc(c);
    if (rc) return rc;
    if (entries) KP_HIP(c, hipMemcpyAsync(c->kpc, columns, (size_t)entries * 3 * c->n * 8, hipMemcpyHostToDevice, c->stream));
    c->fd_kind = 3; c->fdk_entries = entries; c->fdk_first = 0;      // (the entry range, as for the key-point ordered payload)
    payload_changed(c);
    c->kpc_valid = true;
    if (!is_pinned(columns)) KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_fd_difference(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (c->fused) {
        // the sweeps read the key-point column store; the records, if something has asked for them, follow
        int rc = difference_to_kpc(c);
        if (rc) return rc;
        if (c->have_rec) { rc = records_from_payload(c); if (rc) return rc; c->rec_synced = true; }
        return KPILQR_OK;
    }
    return records_from_payload(c);
}

int kpilqr_interpolate(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_interpolate before kpilqr_set_keypoints");
    if (c->fused) { const int rc = ensure_records(c); if (rc) return rc; }
    KP_HIP(c, launch_interpolate(c));
    return KPILQR_OK;
}

// Optimiser::FilterDynamicsMatrices (Optimiser.cpp:340-406) on the materialised A sequence
int kpilqr_filter_dynamics(kpilqr_ctx *c, const char *method, const double *coefs, int ncoef)
{
    if (!c || !method || !coefs) return KPILQR_ERR_ARG;
    KP_ENTER(c);
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
    KP_ENTER(c);
    const size_t B = c->d.batch, T1 = c->d.T + 1, n = c->n, m = c->d.m, nr = c->d.nr;
    if (r) KP_HIP(c, hipMemcpyAsync(c->r, r, B * T1 * nr * 8, hipMemcpyHostToDevice, c->stream));
    if (r_x) { KP_HIP(c, hipMemcpyAsync(c->r_x, r_x, B * T1 * nr * n * 8, hipMemcpyHostToDevice, c->stream)); c->rx_const_on = false; c->rx_buf_valid = true; }
    if (r_u) { KP_HIP(c, hipMemcpyAsync(c->r_u, r_u, B * T1 * nr * m * 8, hipMemcpyHostToDevice, c->stream)); c->ru_zero = false; }
    if (w_run) KP_HIP(c, hipMemcpyAsync(c->w_run, w_run, nr * 8, hipMemcpyHostToDevice, c->stream));
    if (w_term) KP_HIP(c, hipMemcpyAsync(c->w_term, w_term, nr * 8, hipMemcpyHostToDevice, c->stream));
    return KPILQR_OK;
}

// One r_x [nr][n] (and r_u [nr][m], or none) for every trajectory and step: a task whose residuals are affine in the state
// (reaching: r = [q - q*, qdot], Reaching.cpp:43-54).  Uploaded once; the fused one-wave sweeps then issue no r_x loads.
int kpilqr_upload_residual_jacobians_const(kpilqr_ctx *c, const double *r_x, const double *r_u)
{
    if (!c || !r_x) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    const size_t n = c->n, m = c->d.m, nr = c->d.nr, reps = (size_t)c->d.batch * (c->d.T + 1);
    { const int rcg = grow_dev(c, (void **)&c->rx_const, &c->rx_const_cap, nr * n * 8, false); if (rcg < 0) return rcg; }
    KP_HIP(c, hipMemcpyAsync(c->rx_const, r_x, nr * n * 8, hipMemcpyHostToDevice, c->stream));
    c->rx_const_on = true; c->rx_buf_valid = false;
    if (r_u) {                                   // dense control residuals: streamed from the (broadcast) buffer like r_x then
        const int rcs = ensure_stage(c, nr * m * 8);
        if (rcs) return rcs;
        KP_HIP(c, hipMemcpyAsync(c->stage, r_u, nr * m * 8, hipMemcpyHostToDevice, c->stream));
        KP_HIP(c, launch_broadcast(c->stream, c->stage, (int)(nr * m), c->r_u, reps));
        c->ru_zero = false;
    } else if (!c->ru_zero) {
        KP_HIP(c, hipMemsetAsync(c->r_u, 0, reps * nr * m * 8, c->stream));
        c->ru_zero = true;
    }
    if (!(is_pinned(r_x) && is_pinned(r_u))) KP_HIP(c, hipStreamSynchronize(c->stream));
    return KPILQR_OK;
}

int kpilqr_cost_derivs(kpilqr_ctx *c)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    { const int rcx = ensure_rx_buffer(c); if (rcx) return rcx; }
    if (c->fused) { const int rc = ensure_records(c); if (rc) return rc; }
    KP_HIP(c, launch_cost_derivs(c));
    return KPILQR_OK;
}

int kpilqr_trajectory_cost(kpilqr_ctx *c, double *cost)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
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
    if (c->rx_const_on && (c->fused || c->tiled_a6) && backward_reads_rx_buffer(c)) { const int rcx = ensure_rx_buffer(c); if (rcx) return rcx; }
    if (c->fused) {
        int rc = check_fused(c);
        if (rc) return rc;
        rc = ensure_kpc(c);
        if (rc) return rc;
        rc = ensure_kps(c);
        if (rc) return rc;
        // Key-point ordered payload, one wave per trajectory or the producer / consumer pair: the sweep (its producer wave)
        // differences the payload itself and leaves kpc
        // behind for the forward sweep -- no differencing kernel.  (It may stop at a failed PD check, so it never marks
        // kpc valid: another backward pass on the same payload differences again.)  Otherwise the payload is differenced
        // into kpc first, once, and the sweeps read kpc.
        const int bform = backward_fused_form(c);
        if (!c->kpc_valid && c->fd_kind == 2 && (bform == 1 || bform == 3 || bform == 5) && c->tune.fused_raw != 0) {
            KP_HIP(c, launch_backward_fused(c, pd_stride, true));
            c->kpc_touched = true;
            // (KPILQR_FUSED_UNI=0, diagnostic: the GENERAL raw sweep has differenced every set inside the sweep -- dividing at its
            // crossings -- and left the columns; the forward sweep's general form walks the slope store, made from them here)
            if (c->tune.fused_uni == 0 && bform == 1 && c->kps) KP_HIP(c, launch_kp_slopes(c, false));
            return KPILQR_OK;
        }
        if (!c->kpc_valid) { rc = difference_to_kpc(c); if (rc) return rc; }
        if (c->kpc_valid) { rc = slopes_for_kpc(c); if (rc) return rc; }
        KP_HIP(c, launch_backward_fused(c, pd_stride, false));
        return KPILQR_OK;
    }
    c->last_bwd_form = 0;
    if (strcmp(c->bwd_variant, "mfma_f64_t1") == 0) KP_HIP(c, launch_backward_mfma(c, pd_stride));
    else if (strncmp(c->bwd_variant, "mfma_f64_tiled", 14) == 0) KP_HIP(c, launch_backward_tiled(c, pd_stride));
    else if (strcmp(c->bwd_variant, "mfma_f64_wide") == 0) KP_HIP(c, launch_backward_wide(c, pd_stride));
    else KP_HIP(c, launch_backward_generic(c, pd_stride));
    return KPILQR_OK;
}

int kpilqr_backward(kpilqr_ctx *c, const double *lambda, int pd_check_stride, int *status, double *delta_J)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (pd_check_stride < 1) return set_err(c, KPILQR_ERR_ARG, "pd_check_stride must be >= 1");
    if (lambda) KP_HIP(c, hipMemcpyAsync(c->lambda, lambda, (size_t)c->d.batch * 8, hipMemcpyHostToDevice, c->stream));
    int rc = run_backward(c, pd_check_stride);
    if (rc) return rc;
    if (status) KP_HIP(c, hipMemcpyAsync(status, c->status, (size_t)c->d.batch * 4, hipMemcpyDeviceToHost, c->stream));
    if (delta_J) KP_HIP(c, hipMemcpyAsync(delta_J, c->delta_J, (size_t)c->d.batch * 8, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// Diagnostic (bench's lambda sweep): the backward pass of a fused context in its instrumented form.  hist [batch][6] = number
// of steps whose (Quu + lambda I)^-1 came from: the third-order Newton-Schulz refresh alone, that plus 1 / 2 / 3 second-order
// steps, the LDL' factorisation (first step, checked steps, re-seeds), the pivoted slow path.  Gains, delta_J and status
// are written as by kpilqr_backward.  Uses the lambda already resident.  Synchronous.
int kpilqr_backward_stats(kpilqr_ctx *c, int pd_check_stride, int *hist)
{
    if (!c || !hist) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (!c->fused) return set_err(c, KPILQR_ERR_STATE, "kpilqr_backward_stats: fused contexts only");
    if (pd_check_stride < 1) return set_err(c, KPILQR_ERR_ARG, "pd_check_stride must be >= 1");
    int rc = check_fused(c);
    if (rc) return rc;
    rc = ensure_kpc(c);
    if (rc) return rc;
    if (!c->kpc_valid) { rc = difference_to_kpc(c); if (rc) return rc; }
    // (the instrumented sweep is the GENERAL form whatever the lists are: it walks the slope store, made here unconditionally)
    rc = ensure_kps(c, true);
    if (rc) return rc;
    KP_HIP(c, launch_kp_slopes(c, false));
    c->kps_valid = true;
    rc = ensure_rx_buffer(c);
    if (rc) return rc;
    const size_t bytes = (size_t)c->d.batch * 6 * sizeof(int);
    rc = ensure_stage(c, bytes);
    if (rc) return rc;
    KP_HIP(c, launch_backward_fused_stats(c, pd_check_stride, (int *)c->stage));
    KP_HIP(c, hipMemcpyAsync(hist, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    return sync_and_report(c);
}

int kpilqr_download_gains(kpilqr_ctx *c, double *K, double *k)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    const size_t B = c->d.batch, T = c->d.T, n = c->n, m = c->d.m;
    if (K) KP_HIP(c, hipMemcpyAsync(K, c->K, B * T * n * m * 8, hipMemcpyDeviceToHost, c->stream));
    if (k) KP_HIP(c, hipMemcpyAsync(k, c->k, B * T * m * 8, hipMemcpyDeviceToHost, c->stream));
    return KPILQR_OK;
}

// iLQR_SVR::LeastImportantDofs, summing branch (iLQR_SVR.cpp:952-968), over the gains of the last backward pass
int kpilqr_dof_importance(kpilqr_ctx *c, int sampling_k_interval, double *sums)
{
    if (!c || !sums) return KPILQR_ERR_ARG;
    KP_ENTER(c);
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
    KP_ENTER(c);
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
    if (c->rx_const_on && (c->fused || c->tiled_a6) && forward_reads_rx_buffer(c)) { const int rcx = ensure_rx_buffer(c); if (rcx) return rcx; }
    if (c->fused) {
        int rc = check_fused(c);
        if (rc) return rc;
        rc = ensure_kpc(c);
        if (rc) return rc;
        // kpc: differenced explicitly, or left behind by the raw backward sweep of this payload
        if (!c->kpc_valid && !c->kpc_touched) { rc = difference_to_kpc(c); if (rc) return rc; }
        rc = ensure_kps(c);
        if (rc) return rc;
        if (c->kpc_valid) { rc = slopes_for_kpc(c); if (rc) return rc; }       // (behind a raw backward sweep: its launch sequence made them)
        KP_HIP(c, launch_forward_fused(c, U_dev));
        return KPILQR_OK;
    }
    c->last_fwd_form = 0;
    if (strcmp(c->fwd_variant, "mfma_f64_t1") == 0) KP_HIP(c, launch_forward_mfma(c, U_dev));
    else if (strncmp(c->fwd_variant, "mfma_f64_tiled", 14) == 0) KP_HIP(c, launch_forward_tiled(c, U_dev));
    else if (strcmp(c->fwd_variant, "mfma_f64_wide") == 0) KP_HIP(c, launch_forward_wide(c, U_dev));
    else KP_HIP(c, launch_forward_generic(c, U_dev));
    return KPILQR_OK;
}

int kpilqr_forward_linear(kpilqr_ctx *c, const double *alphas, double *cost_pred, double *U_alpha)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
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
    KP_ENTER(c);
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_iterate before kpilqr_set_keypoints");
    if (pd_check_stride < 1) return set_err(c, KPILQR_ERR_ARG, "pd_check_stride must be >= 1");
    if (lambda) KP_HIP(c, hipMemcpyAsync(c->lambda, lambda, (size_t)c->d.batch * 8, hipMemcpyHostToDevice, c->stream));
    if (alphas) KP_HIP(c, hipMemcpyAsync(c->alphas, alphas, (size_t)c->d.n_alpha * 8, hipMemcpyHostToDevice, c->stream));
    if (!c->fused) {              // the fused sweeps difference (or read kpc), interpolate A, B and form l_* themselves
        { const int rcp = records_from_payload(c); if (rcp) return rcp; }
        KP_HIP(c, launch_interpolate(c));
        if (!c->tiled_a6) { const int rcx = ensure_rx_buffer(c); if (rcx) return rcx; KP_HIP(c, launch_cost_derivs(c)); }      // tiled + flag: l_* are formed inside the sweeps
    }
    int rc = run_backward(c, pd_check_stride);
    if (rc) return rc;
    return run_forward(c, nullptr);
}

// ---- whole iteration, pipelined over chunks of trajectories ------------------------------------------------------------
static int pipe_setup(kpilqr_ctx *c)
{
    if (c->pipe_ready) return KPILQR_OK;
    for (int i = 0; i < Ctx::kPipeStreams; i++) {
        KP_HIP(c, hipStreamCreateWithFlags(&c->pipe_stream[i], hipStreamNonBlocking));
        KP_HIP(c, hipEventCreateWithFlags(&c->pipe_done[i], hipEventDisableTiming));
    }
    KP_HIP(c, hipEventCreateWithFlags(&c->pipe_in, hipEventDisableTiming));
    c->pipe_ready = true;
    return KPILQR_OK;
}

// A view of trajectories [b0, b0+nb) of context c on stream s: every per-trajectory pointer shifted, so the ordinary
// launchers run unchanged on the chunk.  n_simd is the chunk's SHARE of the chip: the launchers pick their wave
// organisation (wave pairs / triples per trajectory) as if the whole batch were in flight, which it is.
static void make_view(const kpilqr_ctx *c, int b0, int nb, hipStream_t s, kpilqr_ctx *v)
{
    *v = *c;
    const size_t T = c->d.T, n = c->n, m = c->d.m, nr = c->d.nr, na = c->d.n_alpha, dof = c->d.dof, o = (size_t)b0;
    v->d.batch = nb; v->stream = s; v->own_stream = false;
    if (v->rec) v->rec += o * T * c->L.stride;
    v->K += o * T * n * m; v->k += o * T * m;
    v->r += o * (T + 1) * nr; v->r_x += o * (T + 1) * nr * n; v->r_u += o * (T + 1) * nr * m;
    v->u_nom += o * T * m; v->lambda += o; v->cost_pred += o * na; v->delta_J += o; v->traj_cost += o; v->status += o;
    v->segmap += o * dof * T; v->kp_offsets += o * dof;
    if (c->kp_traj_first_host) { v->fdk_first = c->kp_traj_first_host[b0]; v->kp_view_entries = c->kp_traj_first_host[b0 + nb] - v->fdk_first; }
    long long share = (long long)c->n_simd * nb / c->d.batch;
    v->n_simd = share < 4 ? 4 : (int)share;
}

int kpilqr_iterate_streamed(kpilqr_ctx *c, const kpilqr_stream_io *io, int pd_check_stride, int nchunks)
{
    if (!c || !io) return KPILQR_ERR_ARG;
    KP_HIP(c, hipSetDevice(c->d.device));
    if (!c->have_kp) return set_err(c, KPILQR_ERR_STATE, "kpilqr_iterate_streamed before kpilqr_set_keypoints");
    if (pd_check_stride < 1) return set_err(c, KPILQR_ERR_ARG, "pd_check_stride must be >= 1");
    const int B = c->d.batch;
    if (nchunks < 1) nchunks = Ctx::kPipeStreams;          // 0: one chunk per pipeline stream
    if (nchunks > B) nchunks = B;
    if (io->fd_slab && (io->njobs < 1 || !io->traj_job_first || (io->nnom > 0 && !io->traj_nom_first)))
        return set_err(c, KPILQR_ERR_ARG, "streamed FD payload needs the per-trajectory job / nominal-row offsets");
    if ((io->fd_slab != nullptr) + (io->fd_kp_slab != nullptr) + (io->kp_columns != nullptr) > 1)
        return set_err(c, KPILQR_ERR_ARG, "one FD payload per iteration: job lists OR key-point ordered OR key-point columns");
    if (io->fd_kp_slab || io->kp_columns) {
        if (!c->kp_traj_first_host || c->kp_total_host < 0)
            return set_err(c, KPILQR_ERR_STATE, "streamed key-point ordered payload: the lists must be known to the host (kpilqr_set_keypoints, or kpilqr_get_keypoints after generating them)");
        if (io->entries != c->kp_total_host) return set_err(c, KPILQR_ERR_ARG, "fd_kp_slab / kp_columns: `entries` is not the number of key-point entries");
    }
    const void *hostp[] = {io->kp_columns, io->fd_kp_slab, io->fd_slab, io->r, io->r_x, io->r_u, io->u_nom, io->lambda, io->K, io->k, io->cost_pred, io->delta_J, io->status};
    for (const void *p : hostp) if (!is_pinned(p)) return set_err(c, KPILQR_ERR_ARG, "kpilqr_iterate_streamed: host buffers must be pinned (kpilqr_host_alloc)");
    if (c->fused) { int rc = check_fused(c); if (rc) return rc; }
    int rc = pipe_setup(c);
    if (rc) return rc;
    // the chunk -> stream map must not change while earlier chunks are still in flight (same-stream order is what
    // protects a chunk's device buffers from the next iteration's uploads)
    if (c->pipe_dirty && c->pipe_chunks != nchunks) { rc = join_pipeline(c); if (rc) return rc; }
    c->pipe_chunks = nchunks;

    const int n = c->n, m = c->d.m, nr = c->d.nr, T = c->d.T, na = c->d.n_alpha;
    // uploads by SDMA, downloads by a copy kernel: the only pairing whose two directions overlap inside this pipeline
    // (KPILQR_PIPE_COPY: bit 0 uploads by kernel, bit 1 downloads by kernel; profiles/r02_pcie_inclusive.txt)
    const int pipe_copy = c->tune.pipe_copy >= 0 ? c->tune.pipe_copy : 2;
    const bool k_up = pipe_copy & 1, k_down = pipe_copy & 2;
    auto h2d = [&](void *dst, const void *src, size_t bytes, hipStream_t st) -> hipError_t {
        return k_up ? launch_copy_in(st, dst, src, bytes) : hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    kpilqr_fd_layout L{};
    const char *slab = (const char *)io->fd_slab;
    if (slab) {
        // ---- validate EVERYTHING before a single operation is enqueued or the context is changed --------------------
        if (io->traj_job_first[0] != 0 || io->traj_job_first[B] != io->njobs)
            return set_err(c, KPILQR_ERR_ARG, "traj_job_first must run from 0 to njobs");
        unsigned long long sig = 1469598103934665603ULL;           // FNV-1a over what decides where a chunk's payload lands
        auto mix = [&](unsigned long long v) { sig = (sig ^ v) * 1099511628211ULL; };
        mix((unsigned)io->njobs); mix((unsigned)io->nnom);
        for (int b = 0; b <= B; b++) {
            const int j = io->traj_job_first[b];
            if (j < 0 || j > io->njobs || (b > 0 && j < io->traj_job_first[b - 1]))
                return set_err(c, KPILQR_ERR_ARG, "traj_job_first not monotone in [0, njobs]");
            mix((unsigned)j);
            if (io->nnom > 0) {
                const int q = io->traj_nom_first[b];
                if (q < 0 || q > io->nnom || (b > 0 && q < io->traj_nom_first[b - 1]))
                    return set_err(c, KPILQR_ERR_ARG, "traj_nom_first not monotone in [0, nnom]");
                mix((unsigned)q);
            }
        }
        // The device slab is laid out from (njobs, nnom) and a chunk's payload lands at its trajectories' job / nominal
        // ranges.  Same-stream order protects a chunk's ranges from the NEXT iteration's uploads only while those ranges
        // are the same; when the layout or the per-trajectory offsets differ from the iteration still in flight, the
        // new uploads could overwrite regions another chunk stream is still differencing -- so the pipeline is joined
        // first (every chunk stream then starts behind everything enqueued so far, through pipe_in below).
        kpilqr_fd_layout probe;
        fd_layout(n, io->njobs, io->nnom, &probe);
        if (c->pipe_dirty && (sig != c->pipe_sig || probe.bytes > c->fd_dev_cap)) { rc = join_pipeline(c); if (rc) return rc; }
        if (probe.bytes > c->fd_dev_cap)                           // a growth frees the old slab: nothing may still read it
            for (int i = 0; i < Ctx::kPipeStreams; i++) KP_HIP(c, hipStreamSynchronize(c->pipe_stream[i]));
        rc = fd_bind(c, io->njobs, io->nnom, &L);
        if (rc) return rc;
        c->pipe_sig = sig;
        c->njobs = io->njobs; c->nnom = io->nnom; c->eps = io->eps;
    }
    kpilqr_fdkp_layout LK{};
    const char *kslab = (const char *)io->fd_kp_slab;
    if (kslab) {
        // a chunk's payload lands at its trajectories' entry range, which only the key-points decide: same-stream order
        // protects it from the next iteration's uploads (new key-points go through KP_ENTER, which joins the pipeline)
        kpilqr_fdkp_layout probe;
        fdkp_layout(n, io->entries, &probe);
        if (probe.bytes + probe.bytes / 8 + 4096 > c->fdk_dev_cap && c->pipe_ready) {
            rc = join_pipeline(c); if (rc) return rc;
            for (int i = 0; i < Ctx::kPipeStreams; i++) KP_HIP(c, hipStreamSynchronize(c->pipe_stream[i]));
        }
        rc = fdk_bind(c, io->entries, &LK);
        if (rc) return rc;
        c->fdk_entries = io->entries; c->fdk_first = 0; c->eps = io->eps;
    }
    const double *kcols = io->kp_columns;
    if (slab) c->fd_kind = 1;
    if (kslab) c->fd_kind = 2;
    if (kcols) { c->fd_kind = 3; c->fdk_entries = io->entries; c->fdk_first = 0; }
    if (slab || kslab || kcols) payload_changed(c);
    // allocations and tables the chunks need are made HERE, on the context: a view never allocates
    if (c->fused || c->fd_kind == 2 || c->fd_kind == 3) {
        rc = ensure_kpc(c); if (rc) return rc;
        rc = ensure_entry_tables(c); if (rc) return rc;
    }
    if (c->fused) {
        // per-DoF lists: a chunk computes the slopes of ITS entries, whose range only the host copy of the lists gives
        if (!c->kp_known_uniform && !c->kp_traj_first_host)
            return set_err(c, KPILQR_ERR_STATE, "kpilqr_iterate_streamed with per-DoF key-point lists: the lists must be known to the host (kpilqr_set_keypoints, or kpilqr_get_keypoints after generating them)");
        rc = ensure_kps(c); if (rc) return rc;
    }
    // No new payload, but the column store of the resident one is stale (key-points changed since a job-list upload): a chunk
    // view has no jobs (its njobs is 0), so the payload is re-differenced HERE, on the context, for the whole batch -- what
    // kpilqr_iterate would do.  (A key-point ordered payload is dropped by new key-points; the chunks handle a resident one.)
    if (c->fused && !slab && !kslab && !kcols && !c->kpc_valid && c->fd_kind == 1) {
        rc = difference_to_kpc(c); if (rc) return rc;
    }
    // Per-step Jacobians in this call end the constant mode -- recorded only HERE, behind every check that can still reject the
    // call (a rejected call must leave the context as it was: round-4 advisor; before, a call refused for an unpinned buffer had
    // already left the constant mode and the next sweep read an r_x buffer that never received the broadcast copy)
    if (io->r_u) c->ru_zero = false;
    if (io->r_x) { c->rx_const_on = false; c->rx_buf_valid = true; }
    // constant residual Jacobians and a kernel family that streams r_x: the broadcast copy is made here, on the context (the
    // chunks' wave organisation is the whole batch's: make_view gives a chunk its share of the SIMDs)
    if (c->rx_const_on && (!c->fused || backward_reads_rx_buffer(c) || forward_reads_rx_buffer(c))) { rc = ensure_rx_buffer(c); if (rc) return rc; }
    // order the chunk streams behind whatever the caller enqueued on the context's stream so far (key-points, weights ...)
    KP_HIP(c, hipEventRecord(c->pipe_in, c->stream));
    // from here on chunk streams hold work: every exit path, errors included, leaves the pipeline marked for joining, so a
    // later kpilqr_sync / kpilqr_destroy waits for the DMAs that read the caller's buffers
    c->pipe_dirty = true;

    bool vflags_valid = c->kpc_valid, vflags_touched = c->kpc_touched, vflags_slopes = c->kps_valid;
    for (int ch = 0; ch < nchunks; ch++) {
        const int b0 = (int)((long long)B * ch / nchunks), b1 = (int)((long long)B * (ch + 1) / nchunks), nb = b1 - b0;
        if (nb <= 0) continue;
        hipStream_t s = c->pipe_stream[ch % Ctx::kPipeStreams];
        KP_HIP(c, hipStreamWaitEvent(s, c->pipe_in, 0));
        kpilqr_ctx v;
        make_view(c, b0, nb, s, &v);
        const size_t o = b0, cnt = nb;
        // ---- H2D of the chunk ------------------------------------------------------------------------------------
        if (slab) {
            const int j0 = io->traj_job_first[b0], j1 = io->traj_job_first[b1];
            const size_t J = (size_t)(j1 - j0), jo = (size_t)j0;
            if (J) {
                KP_HIP(c, h2d(c->xplus + jo * n, slab + L.xplus + jo * n * 8, J * n * 8, s));
                KP_HIP(c, h2d(c->xminus + jo * n, slab + L.xminus + jo * n * 8, J * n * 8, s));
                KP_HIP(c, h2d(c->job_b + jo, slab + L.job_b + jo * 4, J * 4, s));
                KP_HIP(c, h2d(c->job_t + jo, slab + L.job_t + jo * 4, J * 4, s));
                KP_HIP(c, h2d(c->job_col + jo, slab + L.job_col + jo * 4, J * 4, s));
                KP_HIP(c, h2d(c->job_nom + jo, slab + L.job_nom + jo * 4, J * 4, s));
                KP_HIP(c, h2d(c->job_mode + jo, slab + L.job_mode + jo, J, s));
            }
            if (io->nnom > 0) {
                const int q0 = io->traj_nom_first[b0], q1 = io->traj_nom_first[b1];
                if (q1 > q0) KP_HIP(c, h2d(c->xnom + (size_t)q0 * n, slab + L.xnom + (size_t)q0 * n * 8, (size_t)(q1 - q0) * n * 8, s));
            }
            // the chunk's jobs: a contiguous range of the job arrays
            v.job_b = c->job_b + jo; v.job_t = c->job_t + jo; v.job_col = c->job_col + jo; v.job_nom = c->job_nom + jo;
            v.job_mode = c->job_mode + jo; v.xplus = c->xplus + jo * n; v.xminus = c->xminus + jo * n; v.njobs = (int)J;
        } else if (kslab) {
            const int e0 = c->kp_traj_first_host[b0], e1 = c->kp_traj_first_host[b1];
            const size_t E = (size_t)(e1 - e0), eo = (size_t)e0;
            if (E) KP_HIP(c, h2d(c->fdk_dev + eo * LK.entry_stride, kslab + eo * LK.entry_stride, E * LK.entry_stride, s));   // the chunk: ONE range
            v.fdk_first = e0; v.fdk_entries = (int)E;      // the chunk's entries
        } else if (kcols) {
            // the chunk's columns, straight into its range of the column store
            const int e0 = c->kp_traj_first_host[b0], e1 = c->kp_traj_first_host[b1];
            const size_t E = (size_t)(e1 - e0), eo = (size_t)e0 * 3 * n;
            if (E) KP_HIP(c, h2d(c->kpc + eo, kcols + eo, E * 3 * n * 8, s));
            v.njobs = 0; v.fdk_first = e0; v.fdk_entries = (int)E;
            v.kpc_valid = true;
        } else {
            // no new FD payload: what was differenced before is reused (kpc / the records' key-point columns)
            v.njobs = 0;
            if ((c->fd_kind == 2 || c->fd_kind == 3) && c->kp_traj_first_host) {
                v.fdk_first = c->kp_traj_first_host[b0]; v.fdk_entries = c->kp_traj_first_host[b1] - v.fdk_first;
            }
        }
        if (io->r) KP_HIP(c, h2d(v.r, io->r + o * (T + 1) * nr, cnt * (T + 1) * nr * 8, s));
        if (io->r_x) KP_HIP(c, h2d(v.r_x, io->r_x + o * (T + 1) * nr * n, cnt * (T + 1) * nr * n * 8, s));
        if (io->r_u) KP_HIP(c, h2d(v.r_u, io->r_u + o * (T + 1) * nr * m, cnt * (T + 1) * nr * m * 8, s));
        if (io->u_nom) KP_HIP(c, h2d(v.u_nom, io->u_nom + o * T * m, cnt * T * m * 8, s));
        if (io->lambda) KP_HIP(c, h2d(v.lambda, io->lambda + o, cnt * 8, s));
        // ---- kernels of the chunk --------------------------------------------------------------------------------
        if (!c->fused) {
            if (slab || kslab || kcols) { rc = records_from_payload(&v); if (rc) { c->err = v.err; return rc; } }
            KP_HIP(c, launch_interpolate(&v));
            if (!c->tiled_a6) KP_HIP(c, launch_cost_derivs(&v));
        }
        rc = run_backward(&v, pd_check_stride);
        if (rc) { c->err = v.err; return rc; }
        rc = run_forward(&v, nullptr);
        if (rc) { c->err = v.err; return rc; }
        vflags_valid = v.kpc_valid; vflags_touched = v.kpc_touched; vflags_slopes = v.kps_valid;
        c->last_bwd_form = v.last_bwd_form; c->last_fwd_form = v.last_fwd_form; c->last_bwd_raw = v.last_bwd_raw; c->last_fwd_form_ragged = v.last_fwd_form_ragged;
        c->last_bwd_ru0 = v.last_bwd_ru0; c->last_fwd_ru0 = v.last_fwd_ru0; c->last_bwd_rxc = v.last_bwd_rxc; c->last_fwd_rxc = v.last_fwd_rxc;
        c->last_bwd_slopes = v.last_bwd_slopes; c->last_fwd_slopes = v.last_fwd_slopes;
        // ---- D2H of the chunk ------------------------------------------------------------------------------------
        // K, k by a copy kernel: it overlaps with the SDMA uploads of the next chunks (two SDMA directions do not)
        if (k_down) {
            if (io->K) KP_HIP(c, launch_copy_out(s, io->K + o * T * n * m, v.K, cnt * T * n * m));
            if (io->k) KP_HIP(c, launch_copy_out(s, io->k + o * T * m, v.k, cnt * T * m));
        } else {
            if (io->K) KP_HIP(c, hipMemcpyAsync(io->K + o * T * n * m, v.K, cnt * T * n * m * 8, hipMemcpyDeviceToHost, s));
            if (io->k) KP_HIP(c, hipMemcpyAsync(io->k + o * T * m, v.k, cnt * T * m * 8, hipMemcpyDeviceToHost, s));
        }
        if (io->cost_pred) KP_HIP(c, hipMemcpyAsync(io->cost_pred + o * na, v.cost_pred, cnt * na * 8, hipMemcpyDeviceToHost, s));
        if (io->delta_J) KP_HIP(c, hipMemcpyAsync(io->delta_J + o, v.delta_J, cnt * 8, hipMemcpyDeviceToHost, s));
        if (io->status) KP_HIP(c, hipMemcpyAsync(io->status + o, v.status, cnt * 4, hipMemcpyDeviceToHost, s));
    }
    c->kpc_valid = vflags_valid; c->kpc_touched = vflags_touched; c->kps_valid = vflags_slopes;      // what every chunk did to its slice of kpc
    return KPILQR_OK;
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
    KP_ENTER(c);
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
    KP_ENTER(c);
    if (c->fused) { const int rcr = ensure_records(c); if (rcr) return rcr; }
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
    KP_ENTER(c);
    if (c->fused) { const int rcr = ensure_records(c); if (rcr) return rcr; }
    const size_t BT = (size_t)c->d.batch * c->d.T, n = c->n, m = c->d.m;
    const size_t szA = BT * n * n * 8, szB = BT * n * m * 8;
    int rc = ensure_stage(c, szA + szB);
    if (rc) return rc;
    double *dA = c->stage, *dB = (double *)((char *)c->stage + szA);
    KP_HIP(c, launch_unpack_AB(c, A ? dA : nullptr, B ? dB : nullptr));
    if (A) KP_HIP(c, hipMemcpyAsync(A, dA, szA, hipMemcpyDeviceToHost, c->stream));
    if (B) KP_HIP(c, hipMemcpyAsync(B, dB, szB, hipMemcpyDeviceToHost, c->stream));
    return sync_and_report(c);
}

int kpilqr_set_cost_derivs(kpilqr_ctx *c, const double *l_x, const double *l_xx, const double *l_u, const double *l_uu)
{
    if (!c) return KPILQR_ERR_ARG;
    KP_ENTER(c);
    if (c->fused) { const int rcr = ensure_records(c); if (rcr) return rcr; }
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
    KP_ENTER(c);
    if (c->fused) { const int rcr = ensure_records(c); if (rcr) return rcr; }
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
    return sync_and_report(c);
}

const char *kpilqr_backward_variant(kpilqr_ctx *c) { return c ? c->bwd_variant : ""; }
const char *kpilqr_forward_variant(kpilqr_ctx *c) { return c ? c->fwd_variant : ""; }

// What the last backward (which = 0) / forward (which = 1) launch of this context WAS: "<variant>" for the materialising
// families; for the fused sweeps "<variant>:<waves>:<columns>:<lists>[:ru0][:rxc][:slopes]" with
//   waves    w1 one wavefront per trajectory | w2 control / state split | pair | triple
//   columns  raw: the backward sweep differenced the key-point ordered payload itself | kpc: read from the column store
//   lists    uni: every DoF of a trajectory has the same key-point list (the straight-line crossing forms ran) | ragged
// The `lists` token is decided on the device (the host never needs it otherwise): this call reads the flag back, i.e. it
// waits for the context's stream.
const char *kpilqr_last_launch(kpilqr_ctx *c, int which)
{
    if (!c || which < 0 || which > 1) return "";
    std::string &out = c->launch_desc[which];
    const int form = which == 0 ? c->last_bwd_form : c->last_fwd_form;
    out = which == 0 ? c->bwd_variant : c->fwd_variant;
    if (form == 0) {
        if (c->fused) { out += ":none"; return out.c_str(); }
        // the two-tile forward sweep on materialised tiles: one wave per row tile, or state / cost wave groups (small batches)
        if (which == 1 && strcmp(c->fwd_variant, "mfma_f64_tiled") == 0 && forward_tiled_sc_selected(c)) out += ":state_cost_waves";
        return out.c_str();
    }
    int uni = 0;
    if (hipSetDevice(c->d.device) != hipSuccess || (c->pipe_dirty && join_pipeline(c) != KPILQR_OK) ||
        hipMemcpyAsync(&uni, c->kp_uniform, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { out += ":?"; return out.c_str(); }
    static const char *const wname[6] = {"", "w1", "w2", "pair", "triple", "pairh"};
    const int ran = (which == 1 && !uni && c->last_fwd_form_ragged) ? c->last_fwd_form_ragged : form;
    out += ":"; out += wname[ran < 6 ? ran : 0];
    // (the raw launch sequence differences inside the sweep for uniform sets only: per-DoF lists take k_fd_kp_difference and
    // the plain sweep, launched behind it -- unless KPILQR_FUSED_UNI=0 forces the general raw form, one wave per trajectory)
    if (which == 0) out += (c->last_bwd_raw && (uni || (c->tune.fused_uni == 0 && form == 1))) ? ":raw" : ":kpc";
    out += uni ? ":uni" : ":ragged";
    if (which == 0 ? c->last_bwd_ru0 : c->last_fwd_ru0) out += ":ru0";
    if (which == 0 ? c->last_bwd_rxc : c->last_fwd_rxc) out += ":rxc";
    if (!uni && (which == 0 ? c->last_bwd_slopes : c->last_fwd_slopes)) out += ":slopes";
    return out.c_str();
}

}  // extern "C"
