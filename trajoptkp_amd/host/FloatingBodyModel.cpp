#include "FloatingBodyModel.h"

#include <cmath>

namespace {
// quaternions are [w, x, y, z]; body-frame angular velocity, as MuJoCo's free joint
void qmul(const double *a, const double *b, double *o)
{
    const double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    const double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    const double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    const double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
void qexp(const double *v, double *o)                 // rotation vector -> quaternion
{
    const double th = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double s = th > 1e-12 ? std::sin(0.5 * th) / th : 0.5 - th * th / 48.0;
    o[0] = std::cos(0.5 * th); o[1] = s * v[0]; o[2] = s * v[1]; o[3] = s * v[2];
}
void qlog(const double *q, double *v)                 // quaternion -> rotation vector (shortest)
{
    double w = q[0], x = q[1], y = q[2], z = q[3];
    if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
    const double sn = std::sqrt(x * x + y * y + z * z);
    const double f = sn > 1e-12 ? 2.0 * std::atan2(sn, w) / sn : 2.0 / w;
    v[0] = f * x; v[1] = f * y; v[2] = f * z;
}
void qconj(const double *q, double *o) { o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3]; }
void qnormalise(double *q)
{
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nn;
}
void rotate_inv(const double *q, const double *v, double *o)     // o = R(q)^T v
{
    double qc[4], t[4], vq[4] = {0, v[0], v[1], v[2]}, r[4];
    qconj(q, qc); qmul(qc, vq, t); qmul(t, q, r);
    o[0] = r[1]; o[1] = r[2]; o[2] = r[3];
}
SimData *fresh()
{
    SimData *d = new SimData();
    d->nq = 7; d->nv = 6; d->nu = 3;
    d->qpos[3] = 1.0;
    return d;
}
}  // namespace

FloatingBodySimulator::FloatingBodySimulator(double timestep, int fd_threads) : dt(timestep)
{
    main_data = fresh();
    master_reset_data = fresh();
    for (int i = 0; i < fd_threads; i++) {
        fd_data.push_back(fresh());
        for (int k = 0; k < 3; k++) fd_scratch.push_back(fresh());
    }
}

FloatingBodySimulator::~FloatingBodySimulator()
{
    delete main_data; delete master_reset_data;
    for (SimData *d : fd_data) delete d;
    for (SimData *d : fd_scratch) delete d;
    for (SimData *d : saved_systems_state_list) delete d;
}

bool FloatingBodySimulator::AppendSystemStateToEnd(SimData *d)
{
    saved_systems_state_list.push_back(new SimData(*d));
    return true;
}

// semi-implicit Euler: velocities from the forces, then positions from the new velocities (the quaternion by the
// exponential of the body-frame angular velocity: mj_integratePos)
bool FloatingBodySimulator::ForwardSimulator(SimData *d) const
{
    double *p = d->qpos, *q = d->qpos + 3, *v = d->qvel, *w = d->qvel + 3;
    const double *F = d->ctrl;
    double Fb[3];
    rotate_inv(q, F, Fb);                                         // force in the body frame
    const double tau[3] = {offset[1] * Fb[2] - offset[2] * Fb[1], offset[2] * Fb[0] - offset[0] * Fb[2], offset[0] * Fb[1] - offset[1] * Fb[0]};
    const double Iw[3] = {inertia[0] * w[0], inertia[1] * w[1], inertia[2] * w[2]};
    const double gyro[3] = {w[1] * Iw[2] - w[2] * Iw[1], w[2] * Iw[0] - w[0] * Iw[2], w[0] * Iw[1] - w[1] * Iw[0]};
    for (int i = 0; i < 3; i++) {
        v[i] += dt * (F[i] / mass - lin_damp * v[i]);
        w[i] += dt * ((tau[i] - gyro[i]) / inertia[i] - ang_damp * w[i]);
    }
    for (int i = 0; i < 3; i++) p[i] += dt * v[i];
    const double rv[3] = {dt * w[0], dt * w[1], dt * w[2]};
    double dq[4], qn[4];
    qexp(rv, dq); qmul(q, dq, qn); qnormalise(qn);
    for (int i = 0; i < 4; i++) q[i] = qn[i];
    d->time += dt;
    return true;
}

// mj_differentiatePos for a free joint: translation by subtraction, rotation as the body-frame rotation vector that
// takes q1 to q2 (mju_subQuat)
void FloatingBodySimulator::DifferentiatePos(double *qvel, double dt_, const SimData *d1, const SimData *d2) const
{
    for (int i = 0; i < 3; i++) qvel[i] = (d2->qpos[i] - d1->qpos[i]) / dt_;
    double qc[4], rel[4], rv[3];
    qconj(d1->qpos + 3, qc); qmul(qc, d2->qpos + 3, rel); qlog(rel, rv);
    for (int i = 0; i < 3; i++) qvel[3 + i] = rv[i] / dt_;
}

// mj_integratePos with dpos = e_{vel_index}
void FloatingBodySimulator::IntegratePos(SimData *d, int vel_index, double eps) const
{
    if (vel_index < 3) { d->qpos[vel_index] += eps; return; }
    double rv[3] = {0, 0, 0}, dq[4], qn[4];
    rv[vel_index - 3] = eps;
    qexp(rv, dq); qmul(d->qpos + 3, dq, qn); qnormalise(qn);
    for (int i = 0; i < 4; i++) d->qpos[3 + i] = qn[i];
}

FloatingBodyTranslator::FloatingBodyTranslator(std::shared_ptr<PhysicsSimulator> sim)
{
    MuJoCo_helper = sim;
    current_state_vector.dof = full_state_vector.dof = 6;
    current_state_vector.dof_quat = full_state_vector.dof_quat = 7;
    current_state_vector.num_ctrl = full_state_vector.num_ctrl = 3;
    const char *names[12] = {"x", "y", "z", "rx", "ry", "rz", "vx", "vy", "vz", "wx", "wy", "wz"};
    for (int i = 0; i < 12; i++) {
        residual r; r.name = names[i];
        r.weight = i < 6 ? 1.0 : 0.05; r.weight_terminal = i < 6 ? 200.0 : 5.0;
        residual_list.push_back(r);
    }
    keypoint_method = "set_interval"; min_N = 4; max_N = 50;
    jerk_thresholds.assign(6, 50.0); velocity_change_thresholds.assign(6, 2.0); iterative_error_threshold = 1e-4;
}

void FloatingBodyTranslator::Residuals(SimData *d, MatrixXd &r)
{
    double gc[4], rel[4], rv[3];
    qconj(goal_quat, gc); qmul(gc, d->qpos + 3, rel); qlog(rel, rv);
    for (int i = 0; i < 3; i++) { r(i) = d->qpos[i] - goal_pos[i]; r(3 + i) = rv[i]; r(6 + i) = d->qvel[i]; r(9 + i) = d->qvel[3 + i]; }
}

MatrixXd FloatingBodyTranslator::ReturnStateVector(SimData *d, const stateVectorList &)
{
    MatrixXd x(12, 1);
    double rv[3];
    qlog(d->qpos + 3, rv);
    for (int i = 0; i < 3; i++) { x(i) = d->qpos[i]; x(3 + i) = rv[i]; x(6 + i) = d->qvel[i]; x(9 + i) = d->qvel[3 + i]; }
    return x;
}

bool FloatingBodyTranslator::SetStateVector(const MatrixXd &x, SimData *d, const stateVectorList &)
{
    if (x.rows() != 12) return false;
    const double rv[3] = {x(3), x(4), x(5)};
    double q[4];
    qexp(rv, q);
    for (int i = 0; i < 3; i++) { d->qpos[i] = x(i); d->qvel[i] = x(6 + i); d->qvel[3 + i] = x(9 + i); }
    for (int i = 0; i < 4; i++) d->qpos[3 + i] = q[i];
    return true;
}

MatrixXd FloatingBodyTranslator::ReturnControlVector(SimData *d, const stateVectorList &)
{
    MatrixXd u(3, 1);
    for (int i = 0; i < 3; i++) u(i) = d->ctrl[i];
    return u;
}

bool FloatingBodyTranslator::SetControlVector(const MatrixXd &u, SimData *d, const stateVectorList &)
{
    if (u.rows() != 3) return false;
    for (int i = 0; i < 3; i++) d->ctrl[i] = u(i);
    return true;
}

MatrixXd FloatingBodyTranslator::ReturnControlLimits(const stateVectorList &)
{
    MatrixXd lim(6, 1);
    for (int i = 0; i < 3; i++) { lim(2 * i) = -force_limit; lim(2 * i + 1) = force_limit; }
    return lim;
}
