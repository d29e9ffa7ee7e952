#include "Differentiator.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

Differentiator::Differentiator(std::shared_ptr<ModelTranslator> mt, std::shared_ptr<PhysicsSimulator> sim)
    : model_translator(std::move(mt)), MuJoCo_helper(std::move(sim)) {}

ThreadPool &Differentiator::pool()
{
    if (!pool_) {
        // hardware_concurrency()-1 workers as the reference (Optimiser.cpp:227,280), at most one per fd_data slot
        int nthreads = (int)std::thread::hardware_concurrency() - 1;
        nthreads = std::max(1, std::min(nthreads, (int)MuJoCo_helper->fd_data.size()));
        pool_.reset(new ThreadPool(nthreads));
    }
    return *pool_;
}

// ---- staging ---------------------------------------------------------------------------------------------
void FDStaging::free_all()
{
    auto rel = [&](void *p) { if (p) { if (release) release(p); else std::free(p); } };
    rel(job_b); rel(job_t); rel(job_col); rel(job_nom); rel(job_mode); rel(xplus); rel(xminus); rel(xnom);
    job_b = job_t = job_col = job_nom = nullptr; job_mode = nullptr; xplus = xminus = xnom = nullptr;
    cap_jobs = cap_nom = 0; njobs = nnom = 0;
}

void FDStaging::reserve(size_t jobs, size_t noms, int n_)
{
    auto get = [&](size_t bytes) { return alloc ? alloc(bytes) : std::malloc(bytes); };
    auto rel = [&](void *p) { if (p) { if (release) release(p); else std::free(p); } };
    auto grow = [&](auto *&ptr, size_t old_count, size_t new_count, size_t keep) {
        using T = typename std::remove_reference<decltype(*ptr)>::type;
        (void)old_count;
        T *np = (T *)get(sizeof(T) * std::max<size_t>(new_count, 1));
        if (ptr && keep) std::memcpy(np, ptr, sizeof(T) * keep);
        rel(ptr);
        ptr = np;
    };
    if (n_ != n) { free_all(); n = n_; }
    if (jobs > cap_jobs) {
        const size_t nc = jobs + jobs / 4 + 64;
        grow(job_b, cap_jobs, nc, (size_t)njobs); grow(job_t, cap_jobs, nc, (size_t)njobs); grow(job_col, cap_jobs, nc, (size_t)njobs);
        grow(job_nom, cap_jobs, nc, (size_t)njobs); grow(job_mode, cap_jobs, nc, (size_t)njobs);
        grow(xplus, cap_jobs * n, nc * n, (size_t)njobs * n); grow(xminus, cap_jobs * n, nc * n, (size_t)njobs * n);
        cap_jobs = nc;
    }
    if (noms > cap_nom) {
        const size_t nc = noms + noms / 4 + 16;
        grow(xnom, cap_nom * n, nc * n, (size_t)nnom * n);
        cap_nom = nc;
    }
}

// ---- one key-point --------------------------------------------------------------------------------------
// The perturb / step / read-back loop of Differentiator::DynamicsDerivatives
// (src/Differentiator/Differentiator.cpp:66-428) for the DoFs `cols` of saved state `data_index`, emitting
// through a sink: sink.nominal(x) -> row of the unperturbed next state; sink.job(col, mode, xp, xm).
// Every DoF emits [ctrl column if i < num_ctrl], velocity column, position column -- always, so the number
// of jobs of a key-point is known before it is differenced (a control column whose both one-sided steps
// violate the limits, i.e. a limit interval narrower than 2 eps, is emitted as a zero column).
namespace {
template <class Sink>
void fd_keypoint(ModelTranslator &mt, PhysicsSimulator &sim, std::atomic<long> &count, Sink &sink,
                 const std::vector<int> &cols, int data_index, int tid, bool central_diff, double eps)
{
    const stateVectorList &sv = mt.current_state_vector;
    const int dof = sv.dof, num_ctrl = sv.num_ctrl, n = 2 * dof;
    SimData *d = sim.fd_data[tid];
    SimData *src = sim.saved_systems_state_list[data_index];
    auto reset = [&]() { sim.CopySystemState(d, src); };
    long steps = 0;

    reset();                                                   // unperturbed next state (:66-71)
    sim.ForwardSimulator(d);
    sink.nominal(mt.ReturnStateVector(d, sv));
    reset();
    const MatrixXd u0 = mt.ReturnControlVector(d, sv);
    const MatrixXd x0 = mt.ReturnStateVector(d, sv);
    const MatrixXd lim = mt.ReturnControlLimits(sv);
    auto stepped = [&](int skip_stage) {
        steps++;
        sim.ForwardSimulatorWithSkip(d, skip_stage, 1);
        return mt.ReturnStateVector(d, sv);
    };
    const MatrixXd zero_state(n, 1);

    for (int i : cols) {
        if (i < num_ctrl) {                                   // ---- controls (:81-223)
            MatrixXd up = u0, um = u0;
            up(i) += eps; um(i) -= eps;
            const bool fwd = !(up(i) > lim(2 * i + 1));
            const bool bwd = (central_diff || !fwd) && !(um(i) < lim(2 * i));
            MatrixXd xp = zero_state, xm = zero_state;
            if (fwd) { mt.SetControlVector(up, d, sv); xp = stepped(2); reset(); }
            if (bwd) { mt.SetControlVector(um, d, sv); xm = stepped(2); reset(); }
            sink.job(n + i, (fwd && bwd) || (!fwd && !bwd) ? 0 : fwd ? 1 : 2, xp, xm);
        }
        {                                                       // ---- velocities (:226-325)
            MatrixXd xq = x0; xq(dof + i) += eps;
            mt.SetStateVector(xq, d, sv);
            MatrixXd xp = stepped(1), xm = zero_state;
            reset();
            if (central_diff) {
                xq = x0; xq(dof + i) -= eps;
                mt.SetStateVector(xq, d, sv);
                xm = stepped(1);
                reset();
            }
            sink.job(dof + i, central_diff ? 0 : 1, xp, xm);
        }
        {                                                       // ---- positions (:328-428), hinge/slide joints
            MatrixXd xq = x0; xq(i) += eps;
            mt.SetStateVector(xq, d, sv);
            MatrixXd xp = stepped(0), xm = zero_state;
            reset();
            if (central_diff) {
                xq = x0; xq(i) -= eps;
                mt.SetStateVector(xq, d, sv);
                xm = stepped(0);
                reset();
            }
            sink.job(i, central_diff ? 0 : 1, xp, xm);
        }
    }
    count.fetch_add(steps, std::memory_order_relaxed);
}

int jobs_of(const std::vector<int> &cols, int num_ctrl)
{
    int c = 0;
    for (int i : cols) c += 2 + (i < num_ctrl ? 1 : 0);
    return c;
}

struct VectorSink {
    FDJobs &jobs; int b, t, n, nom_row;
    void nominal(const MatrixXd &x) { nom_row = jobs.nnom(n); for (int i = 0; i < n; i++) jobs.xnom.push_back(x(i)); }
    void job(int col, int mode, const MatrixXd &xp, const MatrixXd &xm)
    {
        jobs.job_b.push_back(b); jobs.job_t.push_back(t); jobs.job_col.push_back(col);
        jobs.job_mode.push_back((unsigned char)mode); jobs.job_nom.push_back(nom_row);
        for (int i = 0; i < n; i++) { jobs.xplus.push_back(xp(i)); jobs.xminus.push_back(xm(i)); }
    }
};

struct SliceSink {
    FDStaging &st; int b, t, n, nom_row, at;       // at: next job index of this key-point's slice
    void nominal(const MatrixXd &x) { std::memcpy(st.xnom + (size_t)nom_row * n, x.data(), sizeof(double) * n); }
    void job(int col, int mode, const MatrixXd &xp, const MatrixXd &xm)
    {
        st.job_b[at] = b; st.job_t[at] = t; st.job_col[at] = col; st.job_mode[at] = (unsigned char)mode; st.job_nom[at] = nom_row;
        std::memcpy(st.xplus + (size_t)at * n, xp.data(), sizeof(double) * n);
        std::memcpy(st.xminus + (size_t)at * n, xm.data(), sizeof(double) * n);
        at++;
    }
};
}  // namespace

void Differentiator::DynamicsDerivatives(FDJobs &jobs, int b, const std::vector<int> &cols, int data_index, int tid,
                                         bool central_diff, double eps)
{
    VectorSink sink{jobs, b, data_index, 2 * model_translator->current_state_vector.dof, 0};
    fd_keypoint(*model_translator, *MuJoCo_helper, count_integrations, sink, cols, data_index, tid, central_diff, eps);
}

// The reference's shape: threads created and joined per call, per-thread job vectors merged afterwards
// (Optimiser::ComputeDynamicsDerivativesAtKeypoints, Optimiser.cpp:239-323).  Kept as the comparison point of
// the harness benchmark; the optimiser uses DynamicsDerivativesBatch.
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
    std::vector<std::thread> threads;
    for (int i = 0; i < nthreads; i++) threads.emplace_back(worker, i);
    for (std::thread &th : threads) th.join();
    MuJoCo_helper->ResetModelAfterFiniteDifferencing();
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

void Differentiator::DynamicsDerivativesBatch(FDStaging &st, int b, const std::vector<std::vector<int>> &keypoints,
                                              double eps, bool append)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int n = 2 * sv.dof;
    if (!append || st.n != n) { st.njobs = 0; st.nnom = 0; }
    std::vector<int> times, first;           // key-point times and the first job index of each
    int total = st.njobs;
    for (size_t t = 0; t < keypoints.size(); t++)
        if (!keypoints[t].empty()) { times.push_back((int)t); first.push_back(total); total += jobs_of(keypoints[t], sv.num_ctrl); }
    const int nom0 = st.nnom;
    st.reserve((size_t)total, (size_t)nom0 + times.size(), n);
    MuJoCo_helper->InitModelForFiniteDifferencing();
    pool().parallel_for((int)times.size(), [&](int it, int tid) {
        SliceSink sink{st, b, times[it], n, nom0 + it, first[it]};
        fd_keypoint(*model_translator, *MuJoCo_helper, count_integrations, sink, keypoints[times[it]], times[it], tid, true, eps);
    });
    MuJoCo_helper->ResetModelAfterFiniteDifferencing();
    st.njobs = total;
    st.nnom = nom0 + (int)times.size();
}

void Differentiator::ResidualDerivatives(double *r_x, double *r_u, int data_index, int tid, double eps)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int dof = sv.dof, m = sv.num_ctrl, n = 2 * dof, nr = (int)model_translator->residual_list.size();
    SimData *d = MuJoCo_helper->fd_data[tid];
    SimData *src = MuJoCo_helper->saved_systems_state_list[data_index];
    MuJoCo_helper->CopySystemState(d, src);
    if (model_translator->ResidualJacobians(d, r_x, r_u)) return;      // closed form: no differencing
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

// Optimiser::ComputeResidualDerivatives (src/Optimiser/Optimiser.cpp:217-236,325-338) on the persistent pool
void Differentiator::ResidualDerivativesAll(double *r_x, double *r_u, int T, double eps)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int m = sv.num_ctrl, n = 2 * sv.dof, nr = (int)model_translator->residual_list.size();
    pool().parallel_for(T + 1, [&](int t, int tid) {
        ResidualDerivatives(r_x + (size_t)t * nr * n, r_u + (size_t)t * nr * m, std::min(t, T), tid, eps);
    });
}
