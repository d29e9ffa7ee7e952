// comm.cpp -- the one collective of the path: an all-reduce of 8 doubles per iteration,
// [sum_b J_pred(alpha_1..6), sum_b delta_J, #trajectories with a valid backward pass], over the intra-node RCCL
// communicator (xGMI), for hosts that are not Python (bench.py does the same through torch.distributed's
// "nccl" backend, which is RCCL on ROCm).  RCCL is opened lazily with dlopen, so libkpilqr.so has no
// link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <cstring>

#include "common.h"

namespace kpilqr {

// the slice of the NCCL/RCCL C API that is used (rccl.h: ncclUniqueId is 128 opaque bytes; ncclDouble = 8; ncclSum = 0)
typedef struct { char internal[128]; } kp_ncclUniqueId;
typedef void *kp_ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    bool ready = false;                        // every symbol below resolved
    int (*GetUniqueId)(kp_ncclUniqueId *) = nullptr;
    int (*CommInitRank)(kp_ncclComm_t *, int, kp_ncclUniqueId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, kp_ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(kp_ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static Rccl g_rccl;

static const char *rccl_open()
{
    if (g_rccl.ready) return nullptr;
    if (g_rccl.lib) { dlclose(g_rccl.lib); g_rccl = Rccl(); }      // an earlier attempt found the library but not its symbols
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) { g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) return "RCCL (librccl.so) not found";
    g_rccl.GetUniqueId = (int (*)(kp_ncclUniqueId *))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(kp_ncclComm_t *, int, kp_ncclUniqueId, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, kp_ncclComm_t, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(kp_ncclComm_t))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
        dlclose(g_rccl.lib);
        g_rccl = Rccl();
        return "RCCL symbols missing";
    }
    g_rccl.ready = true;
    return nullptr;
}

const char *comm_unique_id(char *id128)
{
    if (const char *e = rccl_open()) return e;
    kp_ncclUniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "ncclGetUniqueId failed";
    std::memcpy(id128, id.internal, 128);
    return nullptr;
}

const char *comm_init(Ctx *c, int nranks, int rank, const char *id128)
{
    if (const char *e = rccl_open()) return e;
    kp_ncclUniqueId id;
    std::memcpy(id.internal, id128, 128);
    kp_ncclComm_t comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);
    if (rc != 0) return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "ncclCommInitRank failed";
    c->comm = comm; c->comm_ranks = nranks;
    return nullptr;
}

void comm_destroy(Ctx *c)
{
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy((kp_ncclComm_t)c->comm);
    c->comm = nullptr;
}

const char *comm_allreduce8(Ctx *c, double *dev8)
{
    if (!c->comm) return nullptr;                 // single rank without a communicator: the local sums are the answer
    const int rc = g_rccl.AllReduce(dev8, dev8, 8, /*ncclDouble*/ 8, /*ncclSum*/ 0, (kp_ncclComm_t)c->comm, c->stream);
    if (rc != 0) return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "ncclAllReduce failed";
    return nullptr;
}

}  // namespace kpilqr
