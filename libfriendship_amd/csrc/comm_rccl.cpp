// comm_rccl.cpp -- the exchange step over RCCL (xGMI): pairwise ncclSend/ncclRecv on the call's stream.
//
// xGMI is point-to-point (7 links per GPU), and the exchange this engine needs is pairwise by construction
// (recursive halving: partner = rank ^ 2^j), so every step uses one direct link per GPU in each direction; no ring
// or tree collective is involved, which also keeps the summation order ours (the graph's own Sum2 association).
//
// RCCL is loaded on first use (dlopen by SONAME): a single-GPU host never touches it, and inside a process that has
// already loaded a copy (PyTorch bundles one under the same SONAME) that copy serves both.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "graph.hpp"   // fr::Error

namespace fr {
namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) { r.error = std::string("cannot load librccl.so.1: ") + dlerror(); return; }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("librccl: missing symbol ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!r.error.empty()) throw Error(FR_ERR_COMM, r.error);
    return r;
}

void check(ncclResult_t rc, const char *what) {
    if (rc != ncclSuccess) throw Error(FR_ERR_COMM, std::string(what) + ": " + rccl().GetErrorString(rc));
}

static_assert(sizeof(ncclUniqueId) == FR_COMM_ID_BYTES, "fr_comm_unique_id hands out an ncclUniqueId");

class RcclTransport : public Transport {
    ncclComm_t comm_ = nullptr;

public:
    RcclTransport(const uint8_t id[128], uint32_t rank, uint32_t world) {
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        check(rccl().CommInitRank(&comm_, (int)world, uid, (int)rank), "ncclCommInitRank");
    }
    ~RcclTransport() override {
        if (comm_) (void)rccl().CommDestroy(comm_);
    }
    void sendrecv(uint32_t peer, const float *d_send, size_t n_send, float *d_recv, size_t n_recv, hipStream_t st) override {
        if (!n_send && !n_recv) return;
        Rccl &r = rccl();
        check(r.GroupStart(), "ncclGroupStart");
        if (n_send) check(r.Send(d_send, n_send, ncclFloat, (int)peer, comm_, st), "ncclSend");
        if (n_recv) check(r.Recv(d_recv, n_recv, ncclFloat, (int)peer, comm_, st), "ncclRecv");
        check(r.GroupEnd(), "ncclGroupEnd");
    }
    const char *name() const override { return "rccl"; }
};

}  // namespace

void rccl_unique_id(uint8_t id[128]) {
    ncclUniqueId uid;
    check(rccl().GetUniqueId(&uid), "ncclGetUniqueId");
    std::memcpy(id, &uid, sizeof uid);
}

std::unique_ptr<Transport> make_rccl_transport(const uint8_t id[128], uint32_t rank, uint32_t world) {
    return std::unique_ptr<Transport>(new RcclTransport(id, rank, world));
}

}  // namespace fr
