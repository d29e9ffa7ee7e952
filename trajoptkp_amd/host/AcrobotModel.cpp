#include "AcrobotModel.h"

#include <cmath>

AcrobotSimulator::AcrobotSimulator(double timestep, int fd_threads) : dt(timestep)
{
    auto fresh = [] { SimData *d = new SimData(); d->nq = 2; d->nv = 2; d->nu = 1; return d; };
    main_data = fresh();
    master_reset_data = fresh();
    for (int i = 0; i < fd_threads; i++) fd_data.push_back(fresh());
}

AcrobotSimulator::~AcrobotSimulator()
{
    delete main_data; delete master_reset_data;
    for (SimData *d : fd_data) delete d;
    for (SimData *d : saved_systems_state_list) delete d;
}

bool AcrobotSimulator::AppendSystemStateToEnd(SimData *d)
{
    saved_systems_state_list.push_back(new SimData(*d));
    return true;
}

// M(q) qdd + C(q,qd) + G(q) + D qd = [u, 0]; semi-implicit Euler (velocity first, then position).
// Angles measured from the upright position as in the task (goal [0,0], hanging = [pi,0]).
bool AcrobotSimulator::ForwardSimulator(SimData *d) const
{
    const double m1 = 1.0, m2 = 1.0, l1 = 1.0, lc1 = 0.5, lc2 = 0.5, I1 = 1.0 / 12.0, I2 = 1.0 / 12.0, g = 9.81, damp = 0.05;
    const double q1 = d->qpos[0], q2 = d->qpos[1], v1 = d->qvel[0], v2 = d->qvel[1];
    const double c2 = std::cos(q2), s2 = std::sin(q2);
    const double M11 = I1 + I2 + m1 * lc1 * lc1 + m2 * (l1 * l1 + lc2 * lc2 + 2 * l1 * lc2 * c2);
    const double M12 = I2 + m2 * (lc2 * lc2 + l1 * lc2 * c2);
    const double M22 = I2 + m2 * lc2 * lc2;
    const double h = m2 * l1 * lc2 * s2;
    const double C1 = -h * v2 * (2 * v1 + v2), C2 = h * v1 * v1;
    const double G1 = -(m1 * lc1 + m2 * l1) * g * std::sin(q1) - m2 * lc2 * g * std::sin(q1 + q2);
    const double G2 = -m2 * lc2 * g * std::sin(q1 + q2);
    const double f1 = d->ctrl[0] - C1 - G1 - damp * v1, f2 = -C2 - G2 - damp * v2;
    const double det = M11 * M22 - M12 * M12;
    const double a1 = (M22 * f1 - M12 * f2) / det, a2 = (M11 * f2 - M12 * f1) / det;
    d->qvel[0] = v1 + dt * a1; d->qvel[1] = v2 + dt * a2;
    d->qpos[0] = q1 + dt * d->qvel[0]; d->qpos[1] = q2 + dt * d->qvel[1];
    d->time += dt;
    return true;
}

AcrobotTranslator::AcrobotTranslator(std::shared_ptr<PhysicsSimulator> sim)
{
    MuJoCo_helper = sim;
    current_state_vector.dof = full_state_vector.dof = 2;
    current_state_vector.dof_quat = full_state_vector.dof_quat = 2;
    current_state_vector.num_ctrl = full_state_vector.num_ctrl = 1;
    const char *names[5] = {"joint_0", "joint_1", "joint_0_vel", "joint_1_vel", "joint_0_torque"};
    const double w[5] = {0, 0, 0.001, 0.001, 100}, wt[5] = {100, 100, 1, 1, 100};      // acrobot.yaml:22-42
    for (int i = 0; i < 5; i++) { residual r; r.name = names[i]; r.weight = w[i]; r.weight_terminal = wt[i]; residual_list.push_back(r); }
    keypoint_method = "set_interval"; min_N = 5; max_N = 100;
    jerk_thresholds = {150, 150}; velocity_change_thresholds = {6.0, 6.0}; iterative_error_threshold = 1e-4;
}

void AcrobotTranslator::Residuals(SimData *d, MatrixXd &r)     // Acrobot.cpp:26-55 (targets are zero)
{
    r(0) = d->qpos[0]; r(1) = d->qpos[1]; r(2) = d->qvel[0]; r(3) = d->qvel[1]; r(4) = d->ctrl[0];
}

bool AcrobotTranslator::ResidualJacobians(SimData *, double *r_x, double *r_u)
{
    if (!analytic_residual_jacobians) return false;
    for (int j = 0; j < 5; j++) {
        for (int i = 0; i < 4; i++) r_x[j * 4 + i] = (j == i) ? 1.0 : 0.0;
        r_u[j] = (j == 4) ? 1.0 : 0.0;
    }
    return true;
}

bool AcrobotTranslator::ConstantResidualJacobians(double *r_x, double *r_u)
{
    if (!constant_residual_jacobians) return false;
    for (int j = 0; j < 5; j++) {
        for (int i = 0; i < 4; i++) r_x[j * 4 + i] = (j == i) ? 1.0 : 0.0;
        r_u[j] = (j == 4) ? 1.0 : 0.0;
    }
    return true;
}

MatrixXd AcrobotTranslator::ReturnStateVector(SimData *d, const stateVectorList &)
{
    MatrixXd x(4, 1);
    x(0) = d->qpos[0]; x(1) = d->qpos[1]; x(2) = d->qvel[0]; x(3) = d->qvel[1];
    return x;
}

bool AcrobotTranslator::SetStateVector(const MatrixXd &x, SimData *d, const stateVectorList &)
{
    if (x.rows() != 4) return false;                               // size mismatch -> false, as ModelTranslator.cpp:989-994
    d->qpos[0] = x(0); d->qpos[1] = x(1); d->qvel[0] = x(2); d->qvel[1] = x(3);
    return true;
}

MatrixXd AcrobotTranslator::ReturnControlVector(SimData *d, const stateVectorList &) { MatrixXd u(1, 1); u(0) = d->ctrl[0]; return u; }

bool AcrobotTranslator::SetControlVector(const MatrixXd &u, SimData *d, const stateVectorList &)
{
    if (u.rows() != 1) return false;
    d->ctrl[0] = u(0);
    return true;
}

MatrixXd AcrobotTranslator::ReturnControlLimits(const stateVectorList &)
{
    MatrixXd lim(2, 1);
    lim(0) = -torque_limit; lim(1) = torque_limit;
    return lim;
}
