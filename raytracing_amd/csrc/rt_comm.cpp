/* rt_comm.cpp -- see rt_comm.h. */
#include "rt_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

namespace rtc {
namespace {

struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Api api;
ncclComm_t comm = nullptr;
int my_rank = 0, n_ranks = 1;

template <class F>
bool sym(F& f, const char* name, std::string& err) {
    f = reinterpret_cast<F>(dlsym(api.lib, name));
    if (!f) err = std::string("RCCL symbol missing: ") + name;
    return f != nullptr;
}

bool load(std::string& err) {
    if (api.lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
    bool ok = sym(api.GetUniqueId, "ncclGetUniqueId", err) && sym(api.CommInitRank, "ncclCommInitRank", err) &&
              sym(api.CommDestroy, "ncclCommDestroy", err) && sym(api.GroupStart, "ncclGroupStart", err) &&
              sym(api.GroupEnd, "ncclGroupEnd", err) && sym(api.Send, "ncclSend", err) && sym(api.Recv, "ncclRecv", err) &&
              sym(api.GetErrorString, "ncclGetErrorString", err);
    if (!ok) { dlclose(api.lib); api = Api(); }
    return ok;
}

bool check(ncclResult_t r, const char* what, std::string& err) {
    if (r == ncclSuccess) return true;
    err = std::string(what) + ": " + api.GetErrorString(r);
    return false;
}

}  // namespace

bool unique_id(char out[128], std::string& err) {
    static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
    if (!load(err)) return false;
    ncclUniqueId id;
    if (!check(api.GetUniqueId(&id), "ncclGetUniqueId", err)) return false;
    std::memcpy(out, &id, 128);
    return true;
}

bool init(const char id_bytes[128], int rank_, int world_, std::string& err) {
    if (comm) { err = "communicator already initialised"; return false; }
    if (world_ < 1 || rank_ < 0 || rank_ >= world_) { err = "rank outside [0, world)"; return false; }
    if (!load(err)) return false;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, 128);
    if (!check(api.CommInitRank(&comm, world_, id, rank_), "ncclCommInitRank", err)) { comm = nullptr; return false; }
    my_rank = rank_; n_ranks = world_;
    return true;
}

void destroy() {
    if (comm) (void)api.CommDestroy(comm);
    comm = nullptr; my_rank = 0; n_ranks = 1;
}

bool active() { return comm != nullptr; }
int rank() { return my_rank; }
int world() { return n_ranks; }

bool gather(const float* d_mine, size_t count, float* d_gathered, int root, hipStream_t st, std::string& err) {
    if (!comm) { err = "no communicator"; return false; }
    if (root < 0 || root >= n_ranks) { err = "no such root"; return false; }
    if (my_rank == root) {
        hipError_t e = hipMemcpyAsync(d_gathered + (size_t)root * count, d_mine, count * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { err = std::string("gather: own part: ") + hipGetErrorString(e); return false; }
    }
    if (n_ranks == 1) return true;
    /* one group: the root posts a receive per peer (each arrives over its own xGMI link), every peer one send */
    if (!check(api.GroupStart(), "ncclGroupStart", err)) return false;
    bool ok = true;
    if (my_rank == root) {
        for (int r = 0; r < n_ranks && ok; ++r)
            if (r != root) ok = check(api.Recv(d_gathered + (size_t)r * count, count, ncclFloat, r, comm, st), "ncclRecv", err);
    } else ok = check(api.Send(d_mine, count, ncclFloat, root, comm, st), "ncclSend", err);
    std::string end_err;
    if (!check(api.GroupEnd(), "ncclGroupEnd", end_err) && ok) { err = end_err; ok = false; }
    return ok;
}

}  // namespace rtc
