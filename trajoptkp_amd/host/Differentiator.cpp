#include "Differentiator.h"

#include <atomic>
#include <thread>

Differentiator::Differentiator(std::shared_ptr<ModelTranslator> mt, std::shared_ptr<PhysicsSimulator> sim)
    : model_translator(std::move(mt)), MuJoCo_helper(std::move(sim)) {}

static void append_state(std::vector<double> &dst, const MatrixXd &x)
{
    for (int i = 0; i < x.rows(); i++) dst.push_back(x(i));
}

void Differentiator::DynamicsDerivatives(FDJobs &jobs, int b, const std::vector<int> &cols, int data_index, int tid,
                                         bool central_diff, double eps)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int dof = sv.dof, num_ctrl = sv.num_ctrl, n = 2 * dof;
    SimData *d = MuJoCo_helper->fd_data[tid];
    SimData *src = MuJoCo_helper->saved_systems_state_list[data_index];
    auto reset = [&]() { MuJoCo_helper->CopySystemState(d, src); };

    // unperturbed next state (:66-71): the nominal row for one-sided differences
    reset();
    MuJoCo_helper->ForwardSimulator(d);
    const int nom_row = jobs.nnom(n);
    append_state(jobs.xnom, model_translator->ReturnStateVector(d, sv));
    reset();
    const MatrixXd u0 = model_translator->ReturnControlVector(d, sv);
    const MatrixXd x0 = model_translator->ReturnStateVector(d, sv);
    const MatrixXd lim = model_translator->ReturnControlLimits(sv);

    auto push_job = [&](int col, int mode, const MatrixXd &xp, const MatrixXd &xm) {
        jobs.job_b.push_back(b); jobs.job_t.push_back(data_index); jobs.job_col.push_back(col);
        jobs.job_mode.push_back((unsigned char)mode); jobs.job_nom.push_back(nom_row);
        append_state(jobs.xplus, xp); append_state(jobs.xminus, xm);
    };
    auto stepped = [&](int skip_stage) {
        count_integrations++;
        MuJoCo_helper->ForwardSimulatorWithSkip(d, skip_stage, 1);
        return model_translator->ReturnStateVector(d, sv);
    };
    const MatrixXd zero_state(n, 1);

    for (int i : cols) {
        if (i < num_ctrl) {                                   // ---- controls (:81-223)
            MatrixXd up = u0, um = u0;
            up(i) += eps; um(i) -= eps;
            const bool fwd = !(up(i) > lim(2 * i + 1));
            const bool bwd = (central_diff || !fwd) && !(um(i) < lim(2 * i));
            MatrixXd xp = zero_state, xm = zero_state;
            if (fwd) { model_translator->SetControlVector(up, d, sv); xp = stepped(2); reset(); }
            if (bwd) { model_translator->SetControlVector(um, d, sv); xm = stepped(2); reset(); }
            if (fwd && bwd) push_job(n + i, 0, xp, xm);
            else if (fwd) push_job(n + i, 1, xp, xm);
            else if (bwd) push_job(n + i, 2, xp, xm);
        }
        {                                                       // ---- velocities (:226-325)
            MatrixXd xq = x0; xq(dof + i) += eps;
            model_translator->SetStateVector(xq, d, sv);
            MatrixXd xp = stepped(1), xm = zero_state;
            reset();
            if (central_diff) {
                xq = x0; xq(dof + i) -= eps;
                model_translator->SetStateVector(xq, d, sv);
                xm = stepped(1);
                reset();
            }
            push_job(dof + i, central_diff ? 0 : 1, xp, xm);
        }
        {                                                       // ---- positions (:328-428), hinge/slide joints
            MatrixXd xq = x0; xq(i) += eps;
            model_translator->SetStateVector(xq, d, sv);
            MatrixXd xp = stepped(0), xm = zero_state;
            reset();
            if (central_diff) {
                xq = x0; xq(i) -= eps;
                model_translator->SetStateVector(xq, d, sv);
                xm = stepped(0);
                reset();
            }
            push_job(i, central_diff ? 0 : 1, xp, xm);
        }
    }
}

void Differentiator::DynamicsDerivativesAtKeypoints(FDJobs &jobs, int b, const std::vector<std::vector<int>> &keypoints, double eps)
{
    MuJoCo_helper->InitModelForFiniteDifferencing();
    std::vector<int> times;
    for (size_t t = 0; t < keypoints.size(); t++) if (!keypoints[t].empty()) times.push_back((int)t);
    int nthreads = (int)std::thread::hardware_concurrency() - 1;
    nthreads = std::max(1, std::min(nthreads, (int)MuJoCo_helper->fd_data.size()));
    std::vector<FDJobs> part(nthreads);
    std::atomic<int> next(0);
    auto worker = [&](int tid) {
        for (;;) {
            const int it = next.fetch_add(1);
            if (it >= (int)times.size()) break;
            DynamicsDerivatives(part[tid], b, keypoints[times[it]], times[it], tid, true, eps);
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < nthreads; i++) pool.emplace_back(worker, i);
    for (std::thread &th : pool) th.join();
    MuJoCo_helper->ResetModelAfterFiniteDifferencing();
    // merge, re-basing the nominal-row indices; jobs of one key-point stay contiguous (GPU slot = run of
    // equal (b,t))
    const int n = 2 * model_translator->current_state_vector.dof;
    for (FDJobs &p : part) {
        const int base = jobs.nnom(n);
        for (int j = 0; j < p.njobs(); j++) {
            jobs.job_b.push_back(p.job_b[j]); jobs.job_t.push_back(p.job_t[j]); jobs.job_col.push_back(p.job_col[j]);
            jobs.job_mode.push_back(p.job_mode[j]); jobs.job_nom.push_back(p.job_nom[j] + base);
        }
        jobs.xplus.insert(jobs.xplus.end(), p.xplus.begin(), p.xplus.end());
        jobs.xminus.insert(jobs.xminus.end(), p.xminus.begin(), p.xminus.end());
        jobs.xnom.insert(jobs.xnom.end(), p.xnom.begin(), p.xnom.end());
    }
}

void Differentiator::ResidualDerivatives(double *r_x, double *r_u, int data_index, int tid, double eps)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int dof = sv.dof, m = sv.num_ctrl, n = 2 * dof, nr = (int)model_translator->residual_list.size();
    SimData *d = MuJoCo_helper->fd_data[tid];
    SimData *src = MuJoCo_helper->saved_systems_state_list[data_index];
    MuJoCo_helper->CopySystemState(d, src);
    const MatrixXd x0 = model_translator->ReturnStateVector(d, sv), u0 = model_translator->ReturnControlVector(d, sv);
    MatrixXd rp(nr, 1), rm(nr, 1);
    for (int i = 0; i < m; i++) {                                // :r_u
        MatrixXd u = u0; u(i) += eps; model_translator->SetControlVector(u, d, sv); model_translator->Residuals(d, rp);
        u = u0; u(i) -= eps; model_translator->SetControlVector(u, d, sv); model_translator->Residuals(d, rm);
        for (int j = 0; j < nr; j++) r_u[j * m + i] = (rp(j) - rm(j)) / (2 * eps);
        model_translator->SetControlVector(u0, d, sv);
    }
    for (int i = 0; i < n; i++) {                                // :r_x (positions then velocities)
        MatrixXd x = x0; x(i) += eps; model_translator->SetStateVector(x, d, sv); model_translator->Residuals(d, rp);
        x = x0; x(i) -= eps; model_translator->SetStateVector(x, d, sv); model_translator->Residuals(d, rm);
        for (int j = 0; j < nr; j++) r_x[j * n + i] = (rp(j) - rm(j)) / (2 * eps);
        model_translator->SetStateVector(x0, d, sv);
    }
}
