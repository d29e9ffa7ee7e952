// Defaults of the tangent-space hooks for models made of hinge / slide joints only (nq == nv): plain coordinate
// arithmetic, which is what mj_differentiatePos / mj_integratePos reduce to for such joints.
#include "PhysicsSimulator.h"
#include "SimData.h"

void PhysicsSimulator::DifferentiatePos(double *qvel, double dt, const SimData *d1, const SimData *d2) const
{
    for (int i = 0; i < nv(); i++) qvel[i] = (d2->qpos[i] - d1->qpos[i]) / dt;
}

void PhysicsSimulator::IntegratePos(SimData *d, int vel_index, double eps) const
{
    d->qpos[vel_index] += eps;
}
