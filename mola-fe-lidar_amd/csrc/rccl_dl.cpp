// rccl_dl.cpp -- RCCL, loaded at run time (no link-time dependency: a single-GPU process never loads it).
// The query-sharded path needs ONE collective: the in-place all-reduce (sum) of the 24-double accumulator
// block, on the workspace's stream, straight on the device buffer.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "hip_backend.hpp"

namespace mola_icp_amd {

namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_api;
std::mutex g_mtx;
std::string g_path_hint;

int load_locked()
{
    if (g_api.lib) return MOLA_ICP_OK;
    void* h = nullptr;
    // the copy this process already uses (torch bundles one) first: two RCCLs on two HIP runtimes do not mix
    const char* names[] = {g_path_hint.empty() ? nullptr : g_path_hint.c_str(), "librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MOLA_ICP_E_COMM, std::string("cannot load RCCL: ") + dlerror());
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy)
        return fail(MOLA_ICP_E_COMM, "the RCCL library lacks a required symbol");
    g_api = a;
    return MOLA_ICP_OK;
}

int rccl_fail(const char* what, int rc)
{
    return fail(MOLA_ICP_E_COMM, std::string(what) + ": " +
                                     (g_api.GetErrorString ? g_api.GetErrorString(rc) : "RCCL error") + " (" +
                                     std::to_string(rc) + ")");
}
}  // namespace

int rccl_set_library(const char* path)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    if (g_api.lib) return MOLA_ICP_OK;  // already loaded
    g_path_hint = path ? path : "";
    return MOLA_ICP_OK;
}

int rccl_unique_id(RcclUniqueId* id)
{
    std::lock_guard<std::mutex> lk(g_mtx);
    int rc = load_locked();
    if (rc) return rc;
    const int r = g_api.GetUniqueId(id);
    return r ? rccl_fail("ncclGetUniqueId", r) : MOLA_ICP_OK;
}

int rccl_comm_init(void** comm, int nranks, const RcclUniqueId& id, int rank)
{
    {
        std::lock_guard<std::mutex> lk(g_mtx);
        int rc = load_locked();
        if (rc) return rc;
    }
    const int r = g_api.CommInitRank(comm, nranks, id, rank);
    return r ? rccl_fail("ncclCommInitRank", r) : MOLA_ICP_OK;
}

int rccl_allreduce_sum_f64(void* comm, double* dev_buf, size_t n, hipStream_t stream)
{
    const int r = g_api.AllReduce(dev_buf, dev_buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, stream);
    return r ? rccl_fail("ncclAllReduce", r) : MOLA_ICP_OK;
}

int rccl_comm_count(void* comm, int* nranks)
{
    if (!comm || !g_api.CommCount) return fail(MOLA_ICP_E_COMM, "no RCCL communicator (or the library lacks ncclCommCount)");
    const int r = g_api.CommCount(comm, nranks);
    return r ? rccl_fail("ncclCommCount", r) : MOLA_ICP_OK;
}

int rccl_comm_destroy(void* comm)
{
    if (!comm || !g_api.CommDestroy) return MOLA_ICP_OK;
    const int r = g_api.CommDestroy(comm);
    return r ? rccl_fail("ncclCommDestroy", r) : MOLA_ICP_OK;
}

}  // namespace mola_icp_amd
