// KeyPointGenerator.h -- host-side key-point placement for the GPU iLQR path.  Public surface follows
// include/KeyPointGenerator.h:50-244 (GenerateKeyPoints, keypoints, last_percentages, Set/Return
// method, Resize); interpolation is NOT here: it runs on the GPU (kpilqr_interpolate).
#pragma once
#include <functional>
#include <vector>
#include "StdInclude.h"

class KeypointGenerator {
public:
    // Computes, for key-point `t` and DoF `dof_index`, the two A columns of that DoF (d/dq_i and
    // d/dqdot_i, n doubles each).  Needed by iterative_error only, which interleaves FD with
    // placement (src/KeyPointGenerator/KeyPointGenerator.cpp:550-640).
    using ColumnFD = std::function<void(int t, int dof_index, double *col_pos, double *col_vel)>;

    KeypointGenerator(int dof, int horizon);
    void SetKeypointMethod(const keypoint_method &method) { current_keypoint_method = method; }
    keypoint_method ReturnCurrentKeypointMethod() const { return current_keypoint_method; }
    void Resize(int new_dof, int new_num_ctrl, int new_horizon);

    // trajectory_states: horizon entries of [q; qdot] (2*dof x 1).  dt = model time-step.
    void GenerateKeyPoints(const std::vector<MatrixXd> &trajectory_states, double dt, const ColumnFD &fd = ColumnFD());
    void ResetCache() { keypoints_computed = false; }

    // keypoints[t] = DoF indices to finite-difference at step t (rows 0 and horizon-1 always full)
    std::vector<std::vector<int>> keypoints;
    std::vector<double> last_percentages;
    std::vector<int> last_num_keypoints;
    int horizon = 0;

    // The C ABI's form: per DoF, the sorted de-duplicated times (kpilqr_set_keypoints).
    void PerDofCSR(std::vector<int> &offsets, std::vector<int> &times) const;

private:
    void SetInterval();
    void Adaptive(const std::vector<MatrixXd> &X, double dt, bool accel);
    void VelocityChange(const std::vector<MatrixXd> &X);
    void IterativeError(const ColumnFD &fd);
    void UpdatePercentages();

    int dof = 0;
    keypoint_method current_keypoint_method;
    bool keypoints_computed = false;
};
