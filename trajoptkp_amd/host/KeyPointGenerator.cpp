// KeyPointGenerator.cpp -- see KeyPointGenerator.h.  Behaviour follows
// src/KeyPointGenerator/KeyPointGenerator.cpp of the reference (line numbers cited per method); the
// parity tests compare every method with the CPU oracle on seeded trajectories.
#include "KeyPointGenerator.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

KeypointGenerator::KeypointGenerator(int dof_, int horizon_) { Resize(dof_, 0, horizon_); }

void KeypointGenerator::Resize(int new_dof, int, int new_horizon)
{
    dof = new_dof;
    horizon = new_horizon;
    last_percentages.assign(dof, 0.0);
    last_num_keypoints.assign(dof, 0);
    keypoints_computed = false;
}

// GenerateKeyPoints dispatch: KeyPointGenerator.cpp:76-135 (unknown method is fatal there too)
void KeypointGenerator::GenerateKeyPoints(const std::vector<MatrixXd> &X, double dt, const ColumnFD &fd)
{
    if (keypoints_computed) return;
    keypoints.clear();
    const std::string &name = current_keypoint_method.name;
    if (name == "set_interval") SetInterval();
    else if (name == "adaptive_jerk") Adaptive(X, dt, false);
    else if (name == "adaptive_accel") Adaptive(X, dt, true);
    else if (name == "velocity_change") VelocityChange(X);
    else if (name == "iterative_error") {
        if (!fd) { std::fprintf(stderr, "ERROR: iterative_error needs a column FD callback\n"); std::exit(1); }
        IterativeError(fd);
    } else {
        std::fprintf(stderr, "ERROR: keyPointsMethod not recognised \n");
        std::exit(1);
    }
    UpdatePercentages();
}

// every min_N-th step for t < horizon-1, plus the last step  (:319-339)
void KeypointGenerator::SetInterval()
{
    std::vector<int> full(dof);
    for (int i = 0; i < dof; i++) full[i] = i;
    keypoints.assign(horizon, std::vector<int>());
    for (int t = 0; t + 1 < horizon; t++)
        if (t % current_keypoint_method.min_N == 0) keypoints[t] = full;
    keypoints[horizon - 1] = full;
}

// jerk profile |d2(qdot)/dt2| (:730-770) -- or, for "adaptive_accel", the signed velocity difference of consecutive steps
// (GenerateAccellerationProfile :772-795: no division by dt, no abs) -- thresholded per DoF with min_N / max_N spacing
// (:341-382; the placement reads jerk_thresholds for either profile, :360)
void KeypointGenerator::Adaptive(const std::vector<MatrixXd> &X, double dt, bool accel)
{
    const keypoint_method &km = current_keypoint_method;
    std::vector<int> full(dof);
    for (int i = 0; i < dof; i++) full[i] = i;
    keypoints.assign(horizon, std::vector<int>());
    keypoints[0] = full;
    std::vector<int> last(dof, 0);
    for (int t = 1; t + 1 < horizon; t++) {
        for (int j = 0; j < dof; j++) {
            double profile = 0.0;
            if (accel) {
                profile = X[t + 1](j + dof) - X[t](j + dof);
            } else if (t < horizon - 2) {
                const double a1 = (X[t + 1](j + dof) - X[t](j + dof)) / dt;
                const double a2 = (X[t + 2](j + dof) - X[t + 1](j + dof)) / dt;
                profile = std::fabs((a2 - a1) / dt);
            }
            if (t - last[j] >= km.min_N && profile > km.jerk_thresholds[j]) { keypoints[t].push_back(j); last[j] = t; }
            if (t - last[j] >= km.max_N) { keypoints[t].push_back(j); last[j] = t; }
        }
    }
    keypoints[horizon - 1] = full;
}

// accumulated |velocity| or a change of direction since the last key-point (:642-728)
void KeypointGenerator::VelocityChange(const std::vector<MatrixXd> &X)
{
    const keypoint_method &km = current_keypoint_method;
    keypoints.assign(horizon, std::vector<int>());
    for (int i = 0; i < dof; i++) keypoints[0].push_back(i);
    std::vector<int> since(dof, 0);
    std::vector<double> travelled(dof, 0.0), last_dir(dof, 0.0);
    for (int t = 1; t < horizon; t++) {
        for (int i = 0; i < dof; i++) {
            since[i]++;
            const double v = X[t](i + dof), dir = v - X[t - 1](i + dof);
            travelled[i] += std::fabs(v);
            bool key = false;
            if (since[i] >= km.min_N && std::fabs(travelled[i]) > km.velocity_change_thresholds[i]) key = true;
            if (!key) {
                if (since[i] >= km.min_N) { if (dir * last_dir[i] < 0) key = true; }
                else last_dir[i] = dir;
            }
            if (!key && since[i] >= km.max_N) key = true;
            if (key) { keypoints[t].push_back(i); travelled[i] = 0.0; since[i] = 0; }
        }
    }
    // the reference appends every DoF to the last row even if already present (:724-727)
    for (int i = 0; i < dof; i++) keypoints[horizon - 1].push_back(i);
}

// per-DoF bisection until the midpoint of each interval is well approximated by the mean of its ends
// (mean squared error over the DoF's velocity rows of both columns, :550-640), intervals <= min_N accepted
void KeypointGenerator::IterativeError(const ColumnFD &fd)
{
    const int n = 2 * dof;
    const keypoint_method &km = current_keypoint_method;
    keypoints.assign(horizon, std::vector<int>());
    std::vector<double> cache;                 // [t][2][n] columns of the DoF being processed
    std::vector<char> have;
    for (int d = 0; d < dof; d++) {
        cache.assign((size_t)horizon * 2 * n, 0.0);
        have.assign(horizon, 0);
        auto need = [&](int t) -> const double * {
            double *c = cache.data() + (size_t)t * 2 * n;
            if (!have[t]) { fd(t, d, c, c + n); have[t] = 1; }
            return c;
        };
        std::vector<index_tuple> todo(1, index_tuple{0, horizon - 1}), next;
        while (!todo.empty()) {
            next.clear();
            for (const index_tuple &iv : todo) {
                if (iv.end_index - iv.start_index <= km.min_N) continue;
                const int mid = (iv.start_index + iv.end_index) / 2;
                const double *cs = need(iv.start_index), *cm = need(mid), *ce = need(iv.end_index);
                double err = 0.0; int cnt = 0;
                for (int k = 0; k < 2; k++)
                    for (int j = dof; j < n; j++) {
                        const double diff = cm[k * n + j] - (cs[k * n + j] + ce[k * n + j]) / 2;
                        err += diff * diff; cnt++;
                    }
                if (!(err / cnt < km.iterative_error_threshold)) {
                    next.push_back(index_tuple{iv.start_index, mid});
                    next.push_back(index_tuple{mid, iv.end_index});
                }
            }
            todo.swap(next);
        }
        for (int t = 0; t < horizon; t++) if (have[t]) keypoints[t].push_back(d);
    }
}

void KeypointGenerator::UpdatePercentages()       // :810-838
{
    std::vector<int> count(dof, 0);
    for (int t = 0; t < horizon; t++)
        for (int i : keypoints[t]) if (i >= 0 && i < dof) count[i]++;
    for (int i = 0; i < dof; i++) {
        last_num_keypoints[i] = count[i];
        last_percentages[i] = ((double)count[i] / (double)horizon) * 100;
    }
}

void KeypointGenerator::PerDofCSR(std::vector<int> &offsets, std::vector<int> &times) const
{
    offsets.assign(1, 0);
    times.clear();
    for (int i = 0; i < dof; i++) {
        int prev = -1;
        for (int t = 0; t < horizon; t++)
            if (t != prev && std::find(keypoints[t].begin(), keypoints[t].end(), i) != keypoints[t].end()) { times.push_back(t); prev = t; }
        offsets.push_back((int)times.size());
    }
}
