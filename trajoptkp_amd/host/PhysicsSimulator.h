// PhysicsSimulator.h -- what the optimiser needs from the physics side, with the method and member
// names of the reference's MuJoCoHelper (include/MuJoCoHelper.h:70-173).  The reference class is
// concrete and owns mjModel/mjData; here it is an interface with an opaque per-state handle so that
// MuJoCo stays on the host behind it (`using SimData = mjData;`).
#pragma once
#include <vector>

struct SimData;   // simulator state (mjData in the reference; SimData.h for the stand-in models of this tree)

class PhysicsSimulator {
public:
    virtual ~PhysicsSimulator() {}
    // mj_step / mj_stepSkip (src/PhysicsSimulators/MuJoCoHelper.cpp; fork-only mj_stepSkip,
    // call sites src/Differentiator/Differentiator.cpp:118,155,253,276,357,377)
    virtual bool ForwardSimulator(SimData *d) const = 0;
    virtual bool ForwardSimulatorWithSkip(SimData *d, int skip_stage, int skip_sensor) const = 0;
    // saved state list (include/MuJoCoHelper.h:117-121)
    virtual bool AppendSystemStateToEnd(SimData *d) = 0;
    virtual bool CheckIfDataIndexExists(int list_index) const { return list_index >= 0 && list_index < (int)saved_systems_state_list.size(); }
    virtual bool CopySystemState(SimData *d_dest, const SimData *d_src) const = 0;
    // FD solver settings: iterations = 5, tolerance = 0 (MuJoCoHelper.cpp:925-937)
    virtual void InitModelForFiniteDifferencing() {}
    virtual void ResetModelAfterFiniteDifferencing() const {}
    virtual double ReturnModelTimeStep() const = 0;

    // ---- tangent-space position arithmetic (models with free / ball joints: nq != nv) --------------------------------
    // mjModel::nv: the size of the velocity (tangent) vector
    virtual int nv() const = 0;
    // mj_differentiatePos(m, qvel, dt, qpos1, qpos2): qvel [nv] = (qpos(d2) (-) qpos(d1)) / dt, quaternion differences as
    // rotation vectors (src/Optimiser/iLQR.cpp:857-861, src/Differentiator/Differentiator.cpp:170-174,288-292,388-393).
    // Default: plain coordinate difference (hinge / slide joints only, nq == nv).
    virtual void DifferentiatePos(double *qvel, double dt, const SimData *d1, const SimData *d2) const;
    // mj_integratePos(m, qpos, dpos, eps) with dpos = e_{vel_index} (Differentiator.cpp:349-357):
    // qpos(d) <- qpos(d) (+) eps * e_{vel_index}.  Default: qpos[vel_index] += eps.
    virtual void IntegratePos(SimData *d, int vel_index, double eps) const;

    std::vector<SimData *> saved_systems_state_list;
    SimData *main_data = nullptr;
    SimData *master_reset_data = nullptr;
    std::vector<SimData *> fd_data;        // one per FD worker thread
    // three more states per FD worker (nominal / plus / minus next state) for models whose position differences are
    // taken in the tangent space -- the reference keeps them as mj_getState arrays (Differentiator.cpp:53-60)
    std::vector<SimData *> fd_scratch;
};
