// Differentiator.h -- host side of the finite differences.  The perturb / step / read-back loop of
// Differentiator::DynamicsDerivatives (src/Differentiator/Differentiator.cpp:8-428) and
// ::ResidualDerivatives (:464-663) stays on the host (it needs the simulator); what the reference does
// after it -- subtract, scale, scatter into A and B (:166-222,286-321,386-423,441-457) -- is the GPU's
// fd_difference stage, so this class only FILLS the job arrays of kpilqr_upload_fd.
#pragma once
#include <memory>
#include <vector>
#include "ModelTranslator.h"

// One batch of FD jobs in the layout of kpilqr_upload_fd (include/kpilqr.h).
struct FDJobs {
    std::vector<int> job_b, job_t, job_col, job_nom;
    std::vector<unsigned char> job_mode;
    std::vector<double> xplus, xminus, xnom;     // [njobs][n], [njobs][n], [nnom][n]
    int njobs() const { return (int)job_t.size(); }
    int nnom(int n) const { return n ? (int)(xnom.size() / n) : 0; }
    void clear() { job_b.clear(); job_t.clear(); job_col.clear(); job_nom.clear(); job_mode.clear(); xplus.clear(); xminus.clear(); xnom.clear(); }
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

    int count_integrations = 0;
private:
    std::shared_ptr<ModelTranslator> model_translator;
    std::shared_ptr<PhysicsSimulator> MuJoCo_helper;
};
