// StdInclude.h -- plain structs shared by the host classes; shapes follow include/StdInclude.h:82-88
// (struct residual) and include/KeyPointGenerator.h:34-48 (keypoint_method, index_tuple).
#pragma once
#include <string>
#include <vector>
#include "Matrix.h"

struct residual {
    std::string name;
    int resid_dimension = 1;
    double weight = 0.0;
    double weight_terminal = 0.0;
};

struct keypoint_method {
    std::string name = "set_interval";   // set_interval | adaptive_jerk | adaptive_accel | velocity_change | iterative_error
    int min_N = 1;
    int max_N = 1;
    std::vector<double> jerk_thresholds;
    std::vector<double> accell_thresholds;
    double iterative_error_threshold = 0.0;
    std::vector<double> velocity_change_thresholds;
    bool auto_adjust = false;
};

struct index_tuple {
    int start_index;
    int end_index;
};

// The part of stateVectorList (include/StdInclude.h:110-249) the optimiser reads.
struct stateVectorList {
    int dof = 0;
    int dof_quat = 0;     // == dof when the model has no quaternion DoFs (the only case the GPU path takes)
    int num_ctrl = 0;
};
