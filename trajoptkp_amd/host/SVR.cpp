#include "SVR.h"

#include <algorithm>
#include <cmath>
#include <numeric>

void ThinSVD(const MatrixXd &a, std::vector<double> &sing, MatrixXd &V)
{
    const int r = a.rows(), c = a.cols();
    // W = a' (c x r): rotate its columns until they are mutually orthogonal; then W = V S, the column norms are the
    // singular values and the normalised columns the right singular vectors of a
    MatrixXd W(c, r);
    for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) W(j, i) = a(i, j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < r - 1; p++)
            for (int q = p + 1; q < r; q++) {
                double app = 0, aqq = 0, apq = 0;
                for (int j = 0; j < c; j++) { app += W(j, p) * W(j, p); aqq += W(j, q) * W(j, q); apq += W(j, p) * W(j, q); }
                if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 1e-17 * std::sqrt(app * aqq)) continue;
                off = std::max(off, std::fabs(apq) / std::sqrt(app * aqq));
                const double zeta = (aqq - app) / (2.0 * apq);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int j = 0; j < c; j++) {
                    const double wp = W(j, p), wq = W(j, q);
                    W(j, p) = cs * wp - sn * wq;
                    W(j, q) = sn * wp + cs * wq;
                }
            }
        if (off < 1e-15) break;
    }
    std::vector<double> nrm(r);
    for (int i = 0; i < r; i++) { double s = 0; for (int j = 0; j < c; j++) s += W(j, i) * W(j, i); nrm[i] = std::sqrt(s); }
    std::vector<int> order(r);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return nrm[x] > nrm[y]; });
    sing.assign(r, 0.0);
    V = MatrixXd(c, r);
    for (int k = 0; k < r; k++) {
        const int i = order[k];
        sing[k] = nrm[i];
        for (int j = 0; j < c; j++) V(j, k) = nrm[i] > 0 ? W(j, i) / nrm[i] : 0.0;
    }
}

std::vector<double> DofImportance(const std::vector<MatrixXd> &K, int dof, int sampling_k_interval, bool eigen_vector_method)
{
    const int T = (int)K.size();
    std::vector<double> sums(dof, 0.0);
    if (T == 0) return sums;
    const int num_ctrl = K[0].rows();
    for (int t = 0; t < T; t += sampling_k_interval) {
        if (eigen_vector_method) {                                  // :902-921
            std::vector<double> sv;
            MatrixXd V;
            ThinSVD(K[t], sv, V);
            const int lead = std::min(3, num_ctrl);                 // the reference reads three triplets (needs num_ctrl >= 3)
            for (int j = 0; j < dof; j++)
                for (int mm = 0; mm < lead; mm++) {
                    sums[j] += std::fabs(V(j, mm) * sv[mm]);
                    sums[j] += std::fabs(V(j + dof, mm) * sv[mm]);
                }
        } else {                                                    // :952-961
            for (int i = 0; i < dof; i++)
                for (int j = 0; j < num_ctrl; j++) {
                    sums[i] += std::fabs(K[t](j, i));
                    sums[i] += std::fabs(K[t](j, i + dof));
                }
        }
    }
    for (int i = 0; i < dof; i++) sums[i] /= T;                       // :923-925, :963-965
    return sums;
}

std::vector<int> LeastImportantDofs(const std::vector<double> &K_dofs_sums, double K_matrix_threshold)
{
    std::vector<int> remove;
    for (int i = 0; i < (int)K_dofs_sums.size(); i++)
        if (K_dofs_sums[i] < K_matrix_threshold) remove.push_back(i);
    return remove;
}
