#include "Differentiator.h"
#include "SimData.h"

#include <algorithm>
#include <cstdio>
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
void FDStaging::plan(int total_jobs, int total_noms, int n_)
{
    size_t off[9];
    layout(total_jobs, total_noms, off);
    if (off[8] > slab_cap || !slab) {
        if (slab) { if (release) release(slab); else std::free(slab); }
        slab_cap = off[8] + off[8] / 4 + 4096;
        slab = (char *)(alloc ? alloc(slab_cap) : std::malloc(slab_cap));
    }
    n = n_;
    xplus = (double *)(slab + off[0]); xminus = (double *)(slab + off[1]); xnom = (double *)(slab + off[2]);
    job_b = (int *)(slab + off[3]); job_t = (int *)(slab + off[4]); job_col = (int *)(slab + off[5]);
    job_nom = (int *)(slab + off[6]); job_mode = (unsigned char *)(slab + off[7]);
    plan_jobs = total_jobs; plan_noms = total_noms;
    njobs = nnom = 0;
}

void FDStaging::free_all()
{
    auto rel = [&](void *p) { if (p) { if (release) release(p); else std::free(p); } };
    if (slab) {                     // slab mode: the arrays point into the slab
        rel(slab); slab = nullptr; slab_cap = 0;
        job_b = job_t = job_col = job_nom = nullptr; job_mode = nullptr; xplus = xminus = xnom = nullptr;
        cap_jobs = cap_nom = 0; njobs = nnom = 0;
        return;
    }
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
    // Models with free / ball joints (dof != dof_quat): the position rows of every column are tangent-space
    // differences, mj_differentiatePos on the full next states (:170-174,288-292,388-393).  The device differences plain
    // vectors, so the rows are handed over already differenced against the step's other state:
    //   central  x+ = (plus (-) minus), x- = 0;   forward  x+ = (plus (-) nominal);   backward  x- = -(nominal (-) minus)
    // with the position entries of the nominal row set to 0.  Hinge / slide models keep plain coordinates (identical there).
    const bool tangent = sv.dof != sv.dof_quat;
    SimData *nomS = nullptr, *plusS = nullptr, *minusS = nullptr;
    std::vector<double> vd;
    if (tangent) {
        if ((int)sim.fd_scratch.size() < 3 * (tid + 1)) { std::fprintf(stderr, "Differentiator: simulator has no fd_scratch states for tangent-space differences\n"); std::exit(1); }
        nomS = sim.fd_scratch[3 * tid]; plusS = sim.fd_scratch[3 * tid + 1]; minusS = sim.fd_scratch[3 * tid + 2];
        vd.resize(sim.nv());
    }
    auto tangent_rows = [&](MatrixXd &x, const SimData *from, const SimData *to, double sign) {
        sim.DifferentiatePos(vd.data(), 1.0, from, to);
        for (int j = 0; j < dof; j++) x(j) = sign * vd[mt.StateIndexToQposIndex(j, sv)];
    };

    reset();                                                   // unperturbed next state (:66-71)
    sim.ForwardSimulator(d);
    {
        MatrixXd xn = mt.ReturnStateVector(d, sv);
        if (tangent) { sim.CopySystemState(nomS, d); for (int j = 0; j < dof; j++) xn(j) = 0.0; }
        sink.nominal(xn);
    }
    reset();
    const MatrixXd u0 = mt.ReturnControlVector(d, sv);
    const MatrixXd x0 = mt.ReturnStateVector(d, sv);
    MatrixXd v0(dof, 1);
    for (int j = 0; j < dof; j++) v0(j) = x0(dof + j);
    const MatrixXd lim = mt.ReturnControlLimits(sv);
    auto stepped = [&](int skip_stage, SimData *keep) {
        steps++;
        sim.ForwardSimulatorWithSkip(d, skip_stage, 1);
        if (tangent) sim.CopySystemState(keep, d);
        return mt.ReturnStateVector(d, sv);
    };
    const MatrixXd zero_state(n, 1);
    // position rows of a finished column (mode as emitted: 0 central, 1 forward, 2 backward)
    auto finish = [&](int mode, bool have_p, bool have_m, MatrixXd &xp, MatrixXd &xm) {
        if (!tangent) return;
        if (mode == 0 && have_p && have_m) { tangent_rows(xp, minusS, plusS, 1.0); for (int j = 0; j < dof; j++) xm(j) = 0.0; }
        else if (mode == 1 && have_p) tangent_rows(xp, nomS, plusS, 1.0);
        else if (mode == 2 && have_m) tangent_rows(xm, minusS, nomS, -1.0);
    };

    for (int i : cols) {
        if (i < num_ctrl) {                                   // ---- controls (:81-223)
            MatrixXd up = u0, um = u0;
            up(i) += eps; um(i) -= eps;
            const bool fwd = !(up(i) > lim(2 * i + 1));
            const bool bwd = (central_diff || !fwd) && !(um(i) < lim(2 * i));
            MatrixXd xp = zero_state, xm = zero_state;
            if (fwd) { mt.SetControlVector(up, d, sv); xp = stepped(2, plusS); reset(); }
            if (bwd) { mt.SetControlVector(um, d, sv); xm = stepped(2, minusS); reset(); }
            const int mode = (fwd && bwd) || (!fwd && !bwd) ? 0 : fwd ? 1 : 2;
            finish(mode, fwd, bwd, xp, xm);
            sink.job(n + i, mode, xp, xm);
        }
        {                                                       // ---- velocities (:226-325)
            MatrixXd vq = v0; vq(i) += eps;
            mt.SetVelocityVector(vq, d, sv);
            MatrixXd xp = stepped(1, plusS), xm = zero_state;
            reset();
            if (central_diff) {
                vq = v0; vq(i) -= eps;
                mt.SetVelocityVector(vq, d, sv);
                xm = stepped(1, minusS);
                reset();
            }
            finish(central_diff ? 0 : 1, true, central_diff, xp, xm);
            sink.job(dof + i, central_diff ? 0 : 1, xp, xm);
        }
        {                                                       // ---- positions (:328-428): mj_integratePos on the tangent index
            const int qi = mt.StateIndexToQposIndex(i, sv);
            sim.IntegratePos(d, qi, eps);
            MatrixXd xp = stepped(0, plusS), xm = zero_state;
            reset();
            if (central_diff) {
                sim.IntegratePos(d, qi, -eps);
                xm = stepped(0, minusS);
                reset();
            }
            finish(central_diff ? 0 : 1, true, central_diff, xp, xm);
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
// Key-point ordered payload (include/kpilqr.h, kpilqr_fd_kp_layout): one record per CSR entry (trajectory, DoF, key-point),
// [x+ (3n) | x- (3n) | int32 mode, pad]; the perturbed next states of DoF i at this key-point go straight to the three
// slots of entry entry_of[i] -- position kind 0, velocity kind 1, control kind 2 -- a one-sided job with the unperturbed
// next state in the slot of the side that was not stepped and its bit in the mode word.  The three kinds of an entry are
// written by the one worker that owns the key-point, so the mode word needs no atomics (it is zeroed with the plan).
struct KpSink {
    char *slab; size_t stride; int n, dof; const int *entry_of;      // entry_of[i]: CSR entry of DoF i at this key-point time
    MatrixXd xnom;
    void nominal(const MatrixXd &x) { xnom = x; }
    void job(int col, int mode, const MatrixXd &xp, const MatrixXd &xm)
    {
        const int kind = col < dof ? 0 : col < n ? 1 : 2;
        const int i = kind == 0 ? col : kind == 1 ? col - dof : col - n;
        // (x+, x-) side by side, element by element (kpilqr_fd_kp_layout: xplus 0, xminus 8, elem_stride 16)
        double *rec = (double *)(slab + (size_t)entry_of[i] * stride);
        const double *P = (mode == 2 ? xnom : xp).data(), *Mn = (mode == 1 ? xnom : xm).data();
        double *dst = rec + (size_t)kind * n * 2;
        for (int r = 0; r < n; r++) { dst[2 * r] = P[r]; dst[2 * r + 1] = Mn[r]; }
        if (mode != 0) *(int *)(rec + (size_t)6 * n) |= 1 << kind;
    }
};
}  // namespace

// Key-point ordered fill of one trajectory: offs / times = its per-DoF CSR (KeypointGenerator::PerDofCSR), entry0 = the CSR
// position of its first entry in the batch's lists.  Every entry's record must belong to [slab, slab + entries * stride).
void Differentiator::DynamicsDerivativesKp(char *slab, size_t stride, int entry0, const std::vector<int> &offs,
                                           const std::vector<int> &times, const std::vector<std::vector<int>> &keypoints, double eps)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int dof = sv.dof, n = 2 * dof, T = (int)keypoints.size();
    std::vector<int> entry_of((size_t)T * dof, -1);            // [t][i]
    for (int i = 0; i < dof; i++)
        for (int e = offs[i]; e < offs[i + 1]; e++) entry_of[(size_t)times[e] * dof + i] = entry0 + e;
    for (int e = offs[0]; e < offs[dof]; e++) {                // zero the mode words (and the pad) of this trajectory's records
        double *rec = (double *)(slab + (size_t)(entry0 + e) * stride);
        rec[6 * n] = 0.0; rec[6 * n + 1] = 0.0;
    }
    std::vector<int> kts;
    for (int t = 0; t < T; t++) if (!keypoints[t].empty()) kts.push_back(t);
    MuJoCo_helper->InitModelForFiniteDifferencing();
    pool().parallel_for((int)kts.size(), [&](int it, int tid) {
        const int t = kts[it];
        KpSink sink{slab, stride, n, dof, entry_of.data() + (size_t)t * dof, MatrixXd()};
        fd_keypoint(*model_translator, *MuJoCo_helper, count_integrations, sink, keypoints[t], t, tid, true, eps);
    });
    MuJoCo_helper->ResetModelAfterFiniteDifferencing();
}

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

// Differentiator::ResidualDerivatives (src/Differentiator/Differentiator.cpp:464-663), central differences: controls with
// the reference's limit-aware one-sided fallback (:496-556), velocities through the state vector (:575-623), positions
// through mj_integratePos on the tangent index (:626-656); the state is restored from the saved one after every column.
void Differentiator::CountJobs(const std::vector<std::vector<int>> &keypoints, int &jobs, int &kps) const
{
    const int num_ctrl = model_translator->current_state_vector.num_ctrl;
    jobs = kps = 0;
    for (const std::vector<int> &cols : keypoints)
        if (!cols.empty()) { kps++; jobs += jobs_of(cols, num_ctrl); }
}

void Differentiator::DynamicsDerivativesPlanned(FDStaging &st, int b, const std::vector<std::vector<int>> &keypoints, double eps)
{
    const stateVectorList &sv = model_translator->current_state_vector;
    const int n = 2 * sv.dof;
    std::vector<int> times, first;           // key-point times and the first job index of each
    int total = st.njobs;
    for (size_t t = 0; t < keypoints.size(); t++)
        if (!keypoints[t].empty()) { times.push_back((int)t); first.push_back(total); total += jobs_of(keypoints[t], sv.num_ctrl); }
    const int nom0 = st.nnom;
    if (total > st.plan_jobs || nom0 + (int)times.size() > st.plan_noms || st.n != n) {
        std::fprintf(stderr, "Differentiator: FD staging plan exceeded\n");
        std::exit(1);
    }
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
    auto reset = [&]() { MuJoCo_helper->CopySystemState(d, src); };
    reset();
    if (model_translator->ResidualJacobians(d, r_x, r_u)) return;      // closed form: no differencing
    const bool central_diff = true;                                     // Optimiser.cpp:325-338
    MatrixXd r0(nr, 1), rp(nr, 1), rm(nr, 1);
    model_translator->Residuals(d, r0);                                 // :487
    const MatrixXd x0 = model_translator->ReturnStateVector(d, sv), u0 = model_translator->ReturnControlVector(d, sv);
    const MatrixXd lim = model_translator->ReturnControlLimits(sv);
    MatrixXd v0(dof, 1);
    for (int j = 0; j < dof; j++) v0(j) = x0(dof + j);
    for (int i = 0; i < m; i++) {                                       // ---- r_u (:496-556)
        MatrixXd u = u0; u(i) += eps;
        const bool fwd = !(u(i) > lim(2 * i + 1));
        if (fwd) { model_translator->SetControlVector(u, d, sv); model_translator->Residuals(d, rp); reset(); }
        u = u0; u(i) -= eps;
        const bool bwd = (central_diff || !fwd) && !(u(i) < lim(2 * i));
        if (bwd) { model_translator->SetControlVector(u, d, sv); model_translator->Residuals(d, rm); reset(); }
        for (int j = 0; j < nr; j++)
            r_u[j * m + i] = (fwd && bwd) ? (rp(j) - rm(j)) / (2 * eps) : fwd ? (rp(j) - r0(j)) / (eps) : bwd ? (r0(j) - rm(j)) / (eps) : 0.0;
    }
    for (int i = 0; i < dof; i++) {                                     // ---- velocities -> r_x column i + dof (:575-623)
        MatrixXd v = v0; v(i) += eps; model_translator->SetVelocityVector(v, d, sv); model_translator->Residuals(d, rp); reset();
        v = v0; v(i) -= eps; model_translator->SetVelocityVector(v, d, sv); model_translator->Residuals(d, rm); reset();
        for (int j = 0; j < nr; j++) r_x[j * n + dof + i] = (rp(j) - rm(j)) / (2 * eps);
    }
    for (int i = 0; i < dof; i++) {                                     // ---- positions -> r_x column i (:626-656)
        const int qi = model_translator->StateIndexToQposIndex(i, sv);
        MuJoCo_helper->IntegratePos(d, qi, eps); model_translator->Residuals(d, rp); reset();
        MuJoCo_helper->IntegratePos(d, qi, -eps); model_translator->Residuals(d, rm); reset();
        for (int j = 0; j < nr; j++) r_x[j * n + i] = (rp(j) - rm(j)) / (2 * eps);
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
