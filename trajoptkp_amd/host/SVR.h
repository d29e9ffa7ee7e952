// SVR.h -- the state-vector-reduction pieces of iLQR_SVR that touch this path (src/Optimiser/iLQR_SVR.cpp:897-968):
// the importance of every DoF measured on the feedback gains of the last backward pass, and the list of DoFs below
// the threshold.  Two measures, as in the reference:
//   * "sampling and summing" (:952-968): sum over sampled steps and controls of |K(j, i)| + |K(j, i + dof)| -- this one
//     also runs on the device over the resident gains (kpilqr_dof_importance);
//   * the singular-vector method (:902-950): K[t] = U S V', sum over the three largest singular triplets of
//     |V(i, k) s_k| + |V(i + dof, k) s_k|.  A handful of small SVDs per optimisation: host work.
// The resize that follows a removal is kpilqr_resize (include/kpilqr.h).
#pragma once
#include <vector>
#include "Matrix.h"

// K[t]: num_ctrl x 2*dof.  Returns sums [dof], already divided by the horizon (:923-925, :963-965).
std::vector<double> DofImportance(const std::vector<MatrixXd> &K, int dof, int sampling_k_interval, bool eigen_vector_method);
// indices of the DoFs whose importance is below K_matrix_threshold (:946-950, :984-988)
std::vector<int> LeastImportantDofs(const std::vector<double> &K_dofs_sums, double K_matrix_threshold);
// Thin SVD of a (rows x cols, rows <= cols) by one-sided Jacobi on a': singular values (descending) and the matching
// right singular vectors V [cols x rows] (column-major).  What Eigen::JacobiSVD<ComputeFullV> yields for the leading
// `rows` triplets (vectors up to sign).
void ThinSVD(const MatrixXd &a, std::vector<double> &sing, MatrixXd &V);
