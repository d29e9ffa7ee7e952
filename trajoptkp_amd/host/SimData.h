// SimData.h -- the simulator state handle of this tree.  In the reference it is MuJoCo's mjData (INTEGRATION.md:
// `using SimData = mjData;`); MuJoCo is not in this image, so the stand-in models (AcrobotModel, FloatingBodyModel)
// share this small POD: generalised positions (nq, quaternions included), velocities (nv), controls (nu).
#pragma once

#define KP_SIM_MAX 16

struct SimData {
    double time = 0.0;
    int nq = 0, nv = 0, nu = 0;
    double qpos[KP_SIM_MAX] = {0};
    double qvel[KP_SIM_MAX] = {0};
    double ctrl[KP_SIM_MAX] = {0};
};
