// cvo_comm.cpp -- see cvo_comm.h.
#include "cvo_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

namespace {

// types come from rccl.h; the entry points are bound with dlsym (no -lrccl)
typedef ncclUniqueId uid_t_;
typedef ncclComm_t ncomm_t_;
typedef ncclResult_t nres_t_;
static_assert(sizeof(uid_t_) == 128, "cvo_hip.h promises a 128-byte unique id");

struct Api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

Api &api()
{
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {   // prefer an instance already in the process (torch's)
            a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (a.handle) break;
        }
        for (int i = 0; !a.handle && i < 3; ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) {
            a.err = std::string("dlopen(librccl) failed: ") + dlerror();
            return;
        }
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) {
            a.err = "librccl is missing nccl* symbols";
            a.handle = nullptr;
        }
    });
    return a;
}

}   // namespace

struct cvo_comm {
    ncomm_t_ comm = nullptr;
    int rank = 0, world = 1;
    std::string err;
};

int cvo_comm_unique_id(void *id_bytes_128)
{
    Api &a = api();
    if (!a.handle) return -1;
    uid_t_ id;
    if (a.GetUniqueId(&id) != ncclSuccess) return -1;
    std::memcpy(id_bytes_128, id.internal, 128);
    return 0;
}

cvo_comm *cvo_comm_create(const void *id_bytes_128, int rank, int world)
{
    Api &a = api();
    if (!a.handle) return nullptr;
    cvo_comm *c = new cvo_comm();
    c->rank = rank;
    c->world = world;
    uid_t_ id;
    std::memcpy(id.internal, id_bytes_128, 128);
    if (a.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) {
        delete c;
        return nullptr;
    }
    return c;
}

void cvo_comm_destroy(cvo_comm *c)
{
    if (!c) return;
    Api &a = api();
    if (a.handle && c->comm) a.CommDestroy(c->comm);
    delete c;
}

int cvo_comm_allreduce(cvo_comm *c, double *dev_buf, int count, hipStream_t stream)
{
    Api &a = api();
    if (!c || !a.handle) return -1;
    const nres_t_ r = a.AllReduce(dev_buf, dev_buf, (size_t)count, ncclFloat64, ncclSum, c->comm,
                                   stream);
    if (r != ncclSuccess) {
        c->err = a.GetErrorString ? a.GetErrorString(r) : "ncclAllReduce failed";
        return -1;
    }
    return 0;
}

const char *cvo_comm_last_error(const cvo_comm *c) { return c ? c->err.c_str() : "no communicator"; }
