// Tensor-parallel communicator over RCCL (xGMI).  The reference resolves the nccl* symbols at run
// time with libloading (crates/core/src/distributed/nccl.rs:479-560) behind `DeviceCommunicator`
// (distributed/communicator.rs:29-98); RCCL exports the same ABI, so this does the same with dlopen
// and exposes only what the decode path needs: all-reduce(sum) of bf16 activations after row-parallel
// linears and all-reduce(max) of 64-bit argmax keys after the vocab-parallel lm_head.  Collectives are
// enqueued on the caller's stream and are hipGraph-capturable.
#include "vmx_kernels.h"

#include <dlfcn.h>
#include <mutex>

namespace {

typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_UINT64 = 5, NCCL_BF16 = 9 };

struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_uid_t *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) return;
        api.GetUniqueId = (int (*)(nccl_uid_t *))dlsym(api.handle, "ncclGetUniqueId");
        api.CommInitRank = (int (*)(nccl_comm_t *, int, nccl_uid_t, int))dlsym(api.handle, "ncclCommInitRank");
        api.CommDestroy = (int (*)(nccl_comm_t))dlsym(api.handle, "ncclCommDestroy");
        api.AllReduce = (int (*)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(api.handle, "ncclAllReduce");
        api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
    });
    return api;
}

int check_rccl(int rc, const char *what) {
    if (rc == 0) return VMX_OK;
    RcclApi &a = rccl();
    vmx_set_error("%s: RCCL error %d (%s)", what, rc, a.GetErrorString ? a.GetErrorString(rc) : "?");
    return VMX_ELAUNCH;
}

}  // namespace

struct vmx_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int vmx_comm_unique_id(void *out128) {
    VMX_REQUIRE(out128, "comm_unique_id: null pointer");
    RcclApi &a = rccl();
    VMX_REQUIRE(a.ok, "comm: librccl could not be loaded");
    nccl_uid_t id;
    if (int rc = check_rccl(a.GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    memcpy(out128, &id, sizeof(id));
    return VMX_OK;
}

extern "C" int vmx_comm_create(const void *id128, int rank, int world, vmx_comm **out) {
    VMX_REQUIRE(id128 && out, "comm_create: null pointer");
    VMX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_create: rank %d / world %d invalid", rank, world);
    RcclApi &a = rccl();
    VMX_REQUIRE(a.ok, "comm: librccl could not be loaded");
    vmx_comm *c = new vmx_comm();
    c->rank = rank;
    c->world = world;
    nccl_uid_t id;
    memcpy(&id, id128, sizeof(id));
    if (int rc = check_rccl(a.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) {
        delete c;
        return rc;
    }
    *out = c;
    return VMX_OK;
}

extern "C" void vmx_comm_destroy(vmx_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
}

extern "C" int vmx_comm_rank(const vmx_comm *c) { return c ? c->rank : 0; }
extern "C" int vmx_comm_world(const vmx_comm *c) { return c ? c->world : 1; }

extern "C" int vmx_comm_all_reduce_sum_bf16(vmx_comm *c, uint16_t *buf, size_t count, void *stream) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    return check_rccl(rccl().AllReduce(buf, buf, count, NCCL_BF16, NCCL_SUM, c->comm, vmx_stream(stream)), "ncclAllReduce(bf16,sum)");
}

extern "C" int vmx_comm_all_reduce_max_u64(vmx_comm *c, unsigned long long *buf, size_t count, void *stream) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    return check_rccl(rccl().AllReduce(buf, buf, count, NCCL_UINT64, NCCL_MAX, c->comm, vmx_stream(stream)), "ncclAllReduce(u64,max)");
}
