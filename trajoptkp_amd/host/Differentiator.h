// Differentiator.h -- host side of the finite differences.  The perturb / step / read-back loop of
// Differentiator::DynamicsDerivatives (src/Differentiator/Differentiator.cpp:8-428) and
// ::ResidualDerivatives (:464-663) stays on the host (it needs the simulator); what the reference does
// after it -- subtract, scale, scatter into A and B (:166-222,286-321,386-423,441-457) -- is the GPU's
// fd_difference stage, so this class only FILLS the job arrays of kpilqr_upload_fd.
#pragma once
#include <atomic>
#include <functional>
#include <memory>
#include <vector>
#include "ModelTranslator.h"
#include "ThreadPool.h"

// One batch of FD jobs in the layout of kpilqr_upload_fd (include/kpilqr.h).
struct FDJobs {
    std::vector<int> job_b, job_t, job_col, job_nom;
    std::vector<unsigned char> job_mode;
    std::vector<double> xplus, xminus, xnom;     // [njobs][n], [njobs][n], [nnom][n]
    int njobs() const { return (int)job_t.size(); }
    int nnom(int n) const { return n ? (int)(xnom.size() / n) : 0; }
    void clear() { job_b.clear(); job_t.clear(); job_col.clear(); job_nom.clear(); job_mode.clear(); xplus.clear(); xminus.clear(); xnom.clear(); }
};

// The same job arrays as raw, reusable buffers that the FD workers fill IN PLACE, each key-point writing its
// own slice -- no per-thread vectors, no merge, and the order (ascending key-point time, then the reference's
// column order) does not depend on thread scheduling.  With `alloc`/`release` bound to kpilqr_host_alloc /
// kpilqr_host_free the buffers are pinned and kpilqr_upload_fd is one DMA per array (north star: "residuals
// shipped in one pinned hipMemcpyAsync").
struct FDStaging {
    std::function<void *(size_t)> alloc;          // default: malloc
    std::function<void(void *)> release;          // default: free
    int *job_b = nullptr, *job_t = nullptr, *job_col = nullptr, *job_nom = nullptr;
    unsigned char *job_mode = nullptr;
    double *xplus = nullptr, *xminus = nullptr, *xnom = nullptr;
    int njobs = 0, nnom = 0, n = 0;
    size_t cap_jobs = 0, cap_nom = 0;
    ~FDStaging() { free_all(); }
    void reserve(size_t jobs, size_t noms, int n_);
    void free_all();

    // SLAB mode (the optimisers use it): all arrays live in ONE pinned allocation laid out by `layout` =
    // kpilqr_fd_slab_layout, so the upload is a single DMA (kpilqr_upload_fd_slab).  plan() sizes the slab for exactly the
    // totals of the coming fill and resets the cursors; Differentiator::DynamicsDerivativesPlanned appends one
    // trajectory's key-points at the cursors.
    std::function<void(int njobs, int nnom, size_t off[9])> layout;   // xplus, xminus, xnom, job_b, job_t, job_col, job_nom, job_mode, bytes
    char *slab = nullptr;
    size_t slab_cap = 0;
    int plan_jobs = 0, plan_noms = 0;
    void plan(int total_jobs, int total_noms, int n_);
    bool complete() const { return njobs == plan_jobs && nnom == plan_noms; }   // every planned job and nominal row was filled
};

class Differentiator {
public:
    Differentiator(std::shared_ptr<ModelTranslator> model_translator, std::shared_ptr<PhysicsSimulator> MuJoCo_helper);

    // Perturbed next states for the DoFs `cols` at saved state `data_index` (one key-point), appended to
    // `jobs` for trajectory `b`.  Control columns respect the control limits with the reference's
    // one-sided fallback (:94-143); central differences otherwise (central_diff = true in the optimiser,
    // src/Optimiser/Optimiser.cpp:319-321).
    void DynamicsDerivatives(FDJobs &jobs, int b, const std::vector<int> &cols, int data_index, int tid,
                             bool central_diff, double eps);
    // Same for every key-point of `keypoints`, spread over hardware_concurrency()-1 threads with an
    // atomic work counter (Optimiser::ComputeDynamicsDerivativesAtKeypoints, Optimiser.cpp:239-323).
    void DynamicsDerivativesAtKeypoints(FDJobs &jobs, int b, const std::vector<std::vector<int>> &keypoints, double eps);
    // r_x [nr][n], r_u [nr][m] at saved state `data_index` by central differences of Residuals()
    void ResidualDerivatives(double *r_x, double *r_u, int data_index, int tid, double eps);

    // All key-points of one trajectory into `st` (appended after what is already there when `append`), on the
    // persistent pool.  Jobs of one key-point are contiguous; key-points in ascending time.
    void DynamicsDerivativesBatch(FDStaging &st, int b, const std::vector<std::vector<int>> &keypoints, double eps,
                                  bool append = false);
    // Slab mode: totals of a key-point set (jobs, key-points) for FDStaging::plan, and the fill of one trajectory at the
    // staging's cursors (job order and contents exactly as DynamicsDerivativesBatch).
    void CountJobs(const std::vector<std::vector<int>> &keypoints, int &jobs, int &kps) const;
    void DynamicsDerivativesPlanned(FDStaging &st, int b, const std::vector<std::vector<int>> &keypoints, double eps);
    // Key-point ordered payload (kpilqr_fd_kp_layout: one record of `stride` bytes per CSR entry): the FD workers write every
    // perturbed next state straight into its slot of the trajectory's records; offs / times = the trajectory's per-DoF CSR,
    // entry0 = the position of its first entry in the batch's lists.  No job arrays, no nominal rows, nothing to walk.
    void DynamicsDerivativesKp(char *slab, size_t stride, int entry0, const std::vector<int> &offs, const std::vector<int> &times,
                               const std::vector<std::vector<int>> &keypoints, double eps);
    // Residual Jacobians of the saved states 0..T into r_x [T+1][nr][n], r_u [T+1][nr][m] on the pool; a task
    // that knows them in closed form (ModelTranslator::ResidualJacobians) skips the differencing altogether.
    void ResidualDerivativesAll(double *r_x, double *r_u, int T, double eps);

    std::atomic<long> count_integrations{0};
    ThreadPool &pool();
private:
    std::unique_ptr<ThreadPool> pool_;
    std::shared_ptr<ModelTranslator> model_translator;
    std::shared_ptr<PhysicsSimulator> MuJoCo_helper;
};
