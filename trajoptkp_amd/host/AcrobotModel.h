// AcrobotModel.h -- a dependency-free PhysicsSimulator + ModelTranslator pair for the plumbing
// configuration (BASELINE configs[0]: acrobot, T=100): a planar double pendulum with the dimensions of
// src/tests/test_xml/Acrobot/acrobot.xml:14-36 (two unit-mass 1 m links, joint damping 0.05, torque on the
// shoulder) and the task of TaskConfigs/toys/acrobot.yaml:13-42 / src/ModelTranslator/Acrobot.cpp:26-55.
// It stands in for MuJoCo, which is not available in this image; it is NOT a MuJoCo re-implementation.
#pragma once
#include "ModelTranslator.h"
#include "SimData.h"

class AcrobotSimulator : public PhysicsSimulator {
public:
    explicit AcrobotSimulator(double timestep = 0.01, int fd_threads = 4);
    ~AcrobotSimulator() override;
    bool ForwardSimulator(SimData *d) const override;
    bool ForwardSimulatorWithSkip(SimData *d, int, int) const override { return ForwardSimulator(d); }
    bool AppendSystemStateToEnd(SimData *d) override;
    bool CopySystemState(SimData *dst, const SimData *src) const override { *dst = *src; return true; }
    double ReturnModelTimeStep() const override { return dt; }
    int nv() const override { return 2; }
private:
    double dt;
};

class AcrobotTranslator : public ModelTranslator {
public:
    explicit AcrobotTranslator(std::shared_ptr<PhysicsSimulator> sim);
    void Residuals(SimData *d, MatrixXd &residuals) override;
    MatrixXd ReturnStateVector(SimData *d, const stateVectorList &) override;
    bool SetStateVector(const MatrixXd &x, SimData *d, const stateVectorList &) override;
    MatrixXd ReturnControlVector(SimData *d, const stateVectorList &) override;
    bool SetControlVector(const MatrixXd &u, SimData *d, const stateVectorList &) override;
    MatrixXd ReturnControlLimits(const stateVectorList &) override;
    // r = [q, qdot, u] is linear: r_x = [I4; 0], r_u = e5 (used only when analytic_residual_jacobians is set;
    // off by default so that the plumbing run differences Residuals() exactly as the reference does)
    bool ResidualJacobians(SimData *d, double *r_x, double *r_u) override;
    // ... and the same at every state: with `constant_residual_jacobians` the shim uploads them once
    bool ConstantResidualJacobians(double *r_x, double *r_u) override;
    bool constant_residual_jacobians = false;
    bool analytic_residual_jacobians = false;
    double torque_limit = 100.0;
};
