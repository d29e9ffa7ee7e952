// PhysicsSimulator.h -- what the optimiser needs from the physics side, with the method and member
// names of the reference's MuJoCoHelper (include/MuJoCoHelper.h:70-173).  The reference class is
// concrete and owns mjModel/mjData; here it is an interface with an opaque per-state handle so that
// MuJoCo stays on the host behind it (`using SimData = mjData;`).
#pragma once
#include <vector>

struct SimData;   // opaque simulator state (mjData in the reference)

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

    std::vector<SimData *> saved_systems_state_list;
    SimData *main_data = nullptr;
    SimData *master_reset_data = nullptr;
    std::vector<SimData *> fd_data;        // one per FD worker thread
};
