// common.h -- internal declarations shared by the translation units of libkpilqr.so.
// Not part of the public surface (that is include/kpilqr.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>

#include "../../include/kpilqr.h"

// Layout of the key-point ordered FD payload's records (kpilqr_fd_kp_layout): 1 = x+ and x- of an element side by side (round 4:
// the sweep that differences the payload fetches both with one 16-byte load); 0 = two blocks of 3n doubles (rounds 1-3; kept as
// a build switch for same-box A/B runs: tools/build_variant.sh NAME -DKP_RAW_PAIRS=0 -- bench.py packs whatever the library reports)
#ifndef KP_RAW_PAIRS
#define KP_RAW_PAIRS 1
#endif

namespace kpilqr {

// ---- device layout of one "step record" (all FP64), one per (trajectory b, time t) ----------
// [ A (n x n) | B (n x m) | l_xx (n x n) | l_x (n) | l_uu (m x m) | l_u (m) ].  A and B are COLUMN-major (as on the host):
// the finite differences produce COLUMNS, so k_fd_difference writes every job as one contiguous run of n doubles whatever
// the key-point pattern (with ragged per-DoF key-points a row-major record made every column n scattered 8-byte stores:
// n = 62, 26 % ragged key-points: 12.9 ms -> see DESIGN.md section 4.1), the interpolation walkers read contiguous
// runs, and the forward sweeps' A' operand is contiguous along lanes; the backward sweeps' tile loads become four
// consecutive rows (32 B) per lane quad.  l_xx and l_uu are symmetric, row-major.  Record stride is padded to 16 doubles
// (128 B).  a(row, col) / b(row, col): element offsets inside a record.
struct RecLayout {
    int n, m;
    int off_A, off_B, off_lxx, off_lx, off_luu, off_lu;
    int rec;      // used doubles
    int stride;   // padded stride in doubles
    __host__ __device__ RecLayout() {}
    __host__ __device__ RecLayout(int n_, int m_) : n(n_), m(m_) {
        off_A = 0;
        off_B = off_A + n * n;
        off_lxx = off_B + n * m;
        off_lx = off_lxx + n * n;
        off_luu = off_lx + n;
        off_lu = off_luu + m * m;
        rec = off_lu + m;
        stride = (rec + 15) & ~15;
    }
    __host__ __device__ int a(int row, int col) const { return off_A + col * n + row; }
    __host__ __device__ int b(int row, int col) const { return off_B + col * n + row; }
};

struct Ctx {
    kpilqr_dims d{};
    int n = 0;
    RecLayout L;
    hipStream_t stream = nullptr;
    int n_simd = 1024;             // SIMDs on the device (CUs x 4)
    bool own_stream = false;
    std::string err;

    // device buffers (cap[]: allocated bytes of the dimension-dependent ones, kpilqr_resize re-uses them)
    size_t cap[18] = {0};
    double *rec = nullptr;        // [batch][T][stride]
    int *kp_uniform = nullptr;    // device flag: every trajectory has one key-point list for all its DoFs (k_kp_uniform)
    double *K = nullptr;          // [batch][T][n*m]
    double *k = nullptr;          // [batch][T][m]
    double *r = nullptr;          // [batch][T+1][nr]
    double *r_x = nullptr;        // [batch][T+1][nr][n]
    double *r_u = nullptr;        // [batch][T+1][nr][m]
    double *w_run = nullptr, *w_term = nullptr;   // [nr]
    double *u_nom = nullptr;      // [batch][T][m]
    double *ctrl_lim = nullptr;   // [2m]
    double *lambda = nullptr;     // [batch]
    double *alphas = nullptr;     // [n_alpha]
    double *cost_pred = nullptr;  // [batch][n_alpha]
    double *delta_J = nullptr;    // [batch]
    double *traj_cost = nullptr;  // [batch]
    int *status = nullptr;        // [batch]
    int2 *segmap = nullptr;       // [batch][dof][T]: (start,end) key-points around t, or (-1,-1)
    int *kp_offsets = nullptr;    // [batch*dof+1]
    int *kp_times = nullptr;      // [kp_total]
    size_t kp_cap = 0;            // capacity of kp_times (ints)
    bool have_kp = false;
    bool kp_canonical = false;   // every DoF list strictly increasing, first 0, last T-1 (what the fused sweeps walk)
    bool fused = false;          // KPILQR_FLAG_FUSED and a supported shape
    bool tiled_a6 = false;       // KPILQR_FLAG_FUSED on a tiled shape: the cost derivatives (a6) are formed inside the sweeps
    bool ru_zero = true;         // r_u was never written since create / resize (the buffer starts zeroed): r_u = 0 exactly
    // constant residual Jacobians (kpilqr_upload_residual_jacobians_const): ONE r_x [nr][n] for every trajectory and step.
    // rx_const_on: the fused one-wave sweeps keep it in registers and never read the r_x buffer; rx_buf_valid: the r_x buffer
    // holds the broadcast copy (made on demand for every other kernel family, ensure_rx_buffer)
    double *rx_const = nullptr;
    size_t rx_const_cap = 0;
    bool rx_const_on = false, rx_buf_valid = true;

    // ---- fused contexts: the key-point column store (no step records) ---------------------------------------------------
    // A fused (one-tile) context does not allocate step records: its sweeps read the differenced key-point columns from
    //   kpc [entry][3][n],  entry = position in the per-DoF CSR (kp_offsets / kp_times),
    //                       kind 0: position column (A col d), 1: velocity column (A col d + dof), 2: control column (B col d, d < m)
    // i.e. 3n doubles per (trajectory, DoF, key-point) instead of a whole record per (trajectory, step).  The records are
    // allocated on demand when something asks for the materialised sequence (kpilqr_interpolate, get_AB, the error test ...).
    double *kpc = nullptr;
    size_t kpc_cap = 0;          // bytes
    bool kpc_valid = false;      // kpc holds ALL differenced columns of the resident FD payload for the current key-points
    bool kpc_touched = false;    // a raw backward sweep has (re)written kpc from the resident payload since it was uploaded
    // slope store beside kpc (k_kp_slopes): kps [entry][3][n][2] = (column value, (column of the list's next key-point - this column) /
    // (time gap)) pairs, slope 0 for a list's last entry.  Read by the general (per-DoF list) forms of the one-wave sweeps; allocated only when the lists may be
    // ragged (kp_known_uniform: the host has seen that every trajectory's DoFs share one list -- then the device flag says the same
    // and only the segment-loop forms run)
    double *kps = nullptr;
    size_t kps_cap = 0;
    bool kps_valid = false;      // kps holds the slopes of the columns kpc holds (kpc_valid)
    bool kp_known_uniform = false;
    int kp_view_entries = -1;    // a view of a trajectory range: the CSR entries of its trajectories (first: fdk_first); -1: the context
    int *kp_entry = nullptr;     // [batch*dof][T]: CSR entry of (list, t), or -1        (built with the segment map)
    int *kp_entry_list = nullptr;// [entries]: list (= b*dof + d) of a CSR entry
    size_t kp_entry_cap = 0, kp_entry_list_cap = 0;   // ints
    bool entry_tables_valid = false;
    bool have_rec = false;       // step records allocated (always on a non-fused context; on demand on a fused one)
    bool rec_synced = false;     // fused context: the records hold the key-point columns of the resident payload
    int kp_total_host = -1;      // number of CSR entries when the host knows it (kpilqr_set_keypoints), else -1
    int *kp_traj_first_host = nullptr;                // [batch+1] first CSR entry of every trajectory (host copy), or null
    // FD payload resident on the device: 0 none, 1 job lists (kpilqr_upload_fd / _slab), 2 key-point ordered (kpilqr_upload_fd_kp)
    int fd_kind = 0;
    // key-point ordered payload: one record per CSR entry, [(x+, x-) pairs of the 3n elements | int32 mode, pad] = fdk_stride() bytes
    char *fdk_dev = nullptr;
    size_t fdk_dev_cap = 0;
    size_t fdk_stride() const { return (size_t)(6 * n + 2) * 8; }
    int fdk_entries = 0;         // entries of the resident key-point ordered payload (a view: of its trajectories)
    int fdk_first = 0;           // first entry of a view's trajectories (0 for the context itself)

    // nominal states for on-device key-point placement (kpilqr_upload_states), allocated on first use
    double *X_states = nullptr;   // [batch][T][n]
    double *kp_thr = nullptr;     // [dof]
    unsigned long long *kp_mask = nullptr;   // [batch*dof][ceil(T/64)]
    int *kp_count = nullptr;      // [batch*dof]
    bool have_states = false;

    // RCCL communicator for the line-search reduction (kpilqr_comm_init), and its 8-double device buffer
    void *comm = nullptr;
    int comm_ranks = 1;
    double *ls8 = nullptr;

    // FD job buffers: ONE device slab (grown on demand) whose layout mirrors the host slab of kpilqr_fd_slab_layout,
    // so that an upload is one hipMemcpyAsync; the pointers below point into it and are recomputed by every upload
    char *fd_dev = nullptr;
    size_t fd_dev_cap = 0;                 // bytes
    int njobs = 0, nnom = 0;
    int *job_b = nullptr, *job_t = nullptr, *job_col = nullptr, *job_nom = nullptr;
    unsigned char *job_mode = nullptr;
    double *xplus = nullptr, *xminus = nullptr, *xnom = nullptr;
    double eps = 1e-6;
    // a view of a trajectory range (kpilqr_iterate_streamed) has shifted per-trajectory pointers; the FD jobs carry
    // ABSOLUTE trajectory indices, so fd_difference addresses the records through the unshifted base
    double *rec_fd_base = nullptr;
    int fd_batch_total = 0;
    // device-side argument checks raise bits here; kpilqr_sync reads it back through the pinned mirror
    int *err_flag = nullptr;
    int *err_flag_host = nullptr;          // pinned

    // trajectory-chunk pipeline of kpilqr_iterate_streamed: H2D(c+1) | kernels(c) | D2H(c-1) on separate streams
    // three streams: with the context's own stream that is the runtime's default of four hardware queues -- a fourth
    // chunk stream shares a queue with another one and the pipeline stalls (measured: B=256, resident Jacobians,
    // 22.7 ms with 3 chunks on 3 streams, 31.7 ms with 4 on 4; profiles/r02_pcie_inclusive.txt)
    static constexpr int kPipeStreams = 3;
    hipStream_t pipe_stream[kPipeStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t pipe_done[kPipeStreams] = {nullptr, nullptr, nullptr};
    hipEvent_t pipe_in = nullptr;
    bool pipe_ready = false, pipe_dirty = false;
    int pipe_chunks = 0;
    unsigned long long pipe_sig = 0;       // hash of (njobs, nnom, traj_job_first, traj_nom_first) of the streamed iteration in flight

    // staging for debug hooks / U_alpha
    double *stage = nullptr;
    size_t stage_cap = 0;   // bytes

    const char *bwd_variant = "";
    const char *fwd_variant = "";
    // what the last backward / forward launch of this context actually was (kpilqr_last_launch): wave organisation
    // (1 one wave per trajectory, 2 control / state split, 3 pair, 4 triple; 0: not a fused launch, or none yet), whether the
    // sweep differenced the raw payload itself, which residual instantiation ran
    int last_bwd_form = 0, last_fwd_form = 0, last_fwd_form_ragged = 0;      // (_ragged: the form that ran instead on per-DoF lists, if another)
    bool last_bwd_raw = false, last_bwd_ru0 = false, last_fwd_ru0 = false, last_bwd_rxc = false, last_fwd_rxc = false;
    bool last_bwd_slopes = false, last_fwd_slopes = false;
    std::string launch_desc[2];

    // Diagnostic switches, read from the environment ONCE by kpilqr_create; the launchers only
    // look here.  0 = let the library choose.
    struct Tuning {
        int fused_bwd_waves = 0;   // KPILQR_FUSED_WAVES: 1 one wave, 5 consumer / helper pair (include/kpilqr.h lists every switch)
        int fused_fwd_waves = 0;   // KPILQR_FUSED_FWD_WAVES: 1 | 2 | 3 | 4
        int fwd_ragged_pair = 0;   // KPILQR_FWD_RAGGED_PAIR: per-DoF lists at 256 < batch <= 512 on the state / cost+staging pair (A/B)
        int role_shift = 9;        // KPILQR_ROLE_SHIFT (wave-pair role placement probe)
        int tiled_nt_min = 0;      // KPILQR_TILED_NT_MIN: run the tiled kernels with more tiles than needed
        int tiled_a6 = -1;         // KPILQR_TILED_A6: -1 auto, 0 | 1
        int tiled_uw = -1;         // KPILQR_TILED_UW: 0 = no u-wave in the tiled backward sweep (NT <= 3, materialised tiles)
        int tiled_fsc = -1;        // KPILQR_TILED_FSC: -1 auto, 0 | 1: the state / cost wave groups of the two-tile forward sweep
        int fused_uni = -1;        // KPILQR_FUSED_UNI: 0 never take the uniform-key-point form of the one-wave backward sweep (diagnostic)
        int fused_raw = -1;        // KPILQR_FUSED_RAW: 0 never difference inside the backward sweep (diagnostic), else auto
        int pipe_copy = -1;        // KPILQR_PIPE_COPY: chunk pipeline copies by kernel: bit 0 uploads, bit 1 downloads (-1 auto)
    } tune;
};

Ctx::Tuning read_tuning_from_env();

#define KP_HIP(ctx, call)                                                        \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);      \
            return KPILQR_ERR_HIP;                                               \
        }                                                                        \
    } while (0)

// ---- launchers (defined in the .hip files) --------------------------------------------------
// elementwise.hip
hipError_t launch_fd_difference(Ctx *c);                 // job lists -> step records
hipError_t launch_fd_difference_kpc(Ctx *c);             // job lists -> key-point column store
hipError_t launch_fd_kp_difference(Ctx *c, bool only_if_ragged = false);   // key-point ordered payload -> key-point column store (only_if_ragged: leaves at once when the device flag kp_uniform is set)
hipError_t launch_kpc_to_records(Ctx *c);                // key-point column store -> step records
hipError_t launch_kp_slopes(Ctx *c, bool only_if_ragged = true);   // key-point column store -> slope store (per-DoF lists only)
hipError_t launch_build_entry_tables(Ctx *c);            // kp_entry, kp_entry_list from the CSR lists
hipError_t launch_copy_out(hipStream_t s, double *dst_host, const double *src_dev, size_t count);   // D2H by a kernel
hipError_t launch_copy_in(hipStream_t s, void *dst_dev, const void *src_host, size_t bytes);        // H2D by a kernel
hipError_t launch_build_segmap(Ctx *c);
// comm.cpp (RCCL opened lazily) and the pack kernel of elementwise.hip
const char *comm_unique_id(char *id128);
const char *comm_init(Ctx *c, int nranks, int rank, const char *id128);
void comm_destroy(Ctx *c);
const char *comm_allreduce8(Ctx *c, double *dev8);
hipError_t launch_pack_linesearch(Ctx *c, double *dev8);
// keypoints.hip
hipError_t launch_generate_keypoints(Ctx *c, int method, int min_N, int max_N, double dt, const double *thr_dev,
                                     const double *X_dev, unsigned long long *mask_dev, int *count_dev);
hipError_t launch_kp_error_test(Ctx *c, int n_iv, const int *iv_dev, int min_N, double threshold, unsigned char *good_dev);
hipError_t launch_interpolate(Ctx *c);
hipError_t launch_filter_dynamics(Ctx *c, int method, const double *coefs_dev, int ncoef);
hipError_t launch_dof_importance(Ctx *c, int sampling, double *sums_dev);
hipError_t launch_cost_derivs(Ctx *c);
hipError_t launch_broadcast_rx(Ctx *c);                  // rx_const -> r_x [batch][T+1][nr][n]
hipError_t launch_broadcast(hipStream_t s, const double *src_dev, int len, double *dst_dev, size_t reps);   // dst [reps][len] = src [len]
hipError_t launch_trajectory_cost(Ctx *c);
// pack/unpack between the reference layout (column-major, separate arrays) and step records
hipError_t launch_pack_AB(Ctx *c, const double *A, const double *B);      // device staging -> records
hipError_t launch_unpack_AB(Ctx *c, double *A, double *B);
hipError_t launch_pack_cost(Ctx *c, const double *lx, const double *lxx, const double *lu, const double *luu);
hipError_t launch_unpack_cost(Ctx *c, double *lx, double *lxx, double *lu, double *luu);

// riccati_generic.hip / forward_generic.hip: any (n, m); LDS-resident, op order of the reference.
hipError_t launch_backward_generic(Ctx *c, int pd_stride);
hipError_t launch_forward_generic(Ctx *c, double *U_alpha_dev);
size_t backward_generic_lds_bytes(int n, int m);

// riccati_mfma.hip / forward_mfma.hip: n+1 <= 16 and m <= 15 (one 16x16 f64 MFMA tile per block).
bool backward_mfma_supported(int n, int m);
hipError_t launch_backward_mfma(Ctx *c, int pd_stride);
bool forward_mfma_supported(int n, int m, int n_alpha);
hipError_t launch_forward_mfma(Ctx *c, double *U_alpha_dev);

// tiled_mfma.hip: n+2 <= 64 (NT x NT grids of 16x16 tiles in LDS), m in {1,7}
int tiled_tiles(int n, int nt_min);                       // tiles per side of the tiled kernels' state grid
bool backward_tiled_supported(int n, int m, int nt_min);
hipError_t launch_backward_tiled(Ctx *c, int pd_stride);
size_t backward_tiled_lds_bytes(int nt);
bool forward_tiled_supported(int n, int m, int n_alpha, int nt_min);
bool forward_tiled_sc_selected(const Ctx *c);             // the state / cost wave groups will run (two tiles, small batches)
hipError_t launch_forward_tiled(Ctx *c, double *U_alpha_dev);
// tiled_wide.hip: the same sweeps with the control block over up to two tiles (8 < m <= 32 backward, 16 < m <= 32 forward)
bool backward_wide_supported(int n, int m, int nt_min);
hipError_t launch_backward_wide(Ctx *c, int pd_stride);
bool forward_wide_supported(int n, int m, int n_alpha, int nt_min);
hipError_t launch_forward_wide(Ctx *c, double *U_alpha_dev);
// fused_mfma.hip: a4 + a6 evaluated inside the sweeps (n+2 <= 16)
bool fused_supported(int n, int m, int nr, int dof, int T, int stride, int n_alpha);
int backward_fused_form(const Ctx *c);
int forward_fused_form(const Ctx *c);
hipError_t launch_backward_fused(Ctx *c, int pd_stride, bool raw);
hipError_t launch_backward_fused_waves(Ctx *c, int pd_stride, bool raw, int form);      // forms 2..5 (fused_mfma.hip, part 2)
hipError_t launch_forward_fused(Ctx *c, double *U_alpha_dev);
hipError_t launch_backward_fused_stats(Ctx *c, int pd_stride, int *hist_dev);

}  // namespace kpilqr
