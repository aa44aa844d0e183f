// Tensor-parallel communicator over RCCL (xGMI).  The reference resolves the nccl* symbols at run
// time with libloading (crates/core/src/distributed/nccl.rs:479-560) behind `DeviceCommunicator`
// (distributed/communicator.rs:29-98); RCCL exports the same ABI, so this does the same with dlopen
// and exposes only what the decode path needs: all-reduce(sum) of bf16 activations after row-parallel
// linears and all-reduce(max) of 64-bit argmax keys after the vocab-parallel lm_head.  Collectives are
// enqueued on the caller's stream and are hipGraph-capturable.
#include "vmx_kernels.h"

#include <dlfcn.h>
#include <stdlib.h>
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

namespace {
// ---------------------------------------------------------------- one-shot all-reduce over peer-mapped memory (xGMI)
// The decode path all-reduces one hidden-size row of bf16 (8 KiB on Qwen3-8B) 72 times per token; a ring collective pays
// 2 (N - 1) hops of launch + link latency for it.  With every rank's staging buffer mapped into every other rank
// (hipIpc handles), the message fits one step (design reference: crates/core/kernels/custom_allreduce.cu:264-484, the
// one-shot kernel of the reference's custom all-reduce): each workgroup copies its 2048-element slice of the input into its
// own rank's staging slot, raises a flag in every peer's flag array, waits for the peers' flags for the same slice, then
// reads the slice from every rank's staging buffer and adds them in rank order (the same order on every rank, so all
// ranks hold the same bits).  Slots and flags are double-buffered by the parity of a per-workgroup call counter kept in
// device memory (so a replayed hipGraph advances it): a rank can run at most one call ahead of its slowest peer, and
// never overwrites a slot a peer may still be reading.  A flag wait gives up after about two seconds and counts the
// timeout in a device word (vmx_comm_p2p_timeouts) instead of hanging the GPU.
constexpr int P2P_MAX_RANKS = 8, P2P_MAX_WG = 64, P2P_WG_ELEMS = 2048;

struct P2PArgs {
    uint16_t *stage[P2P_MAX_RANKS];   // rank r's staging buffer as mapped in this process: [2 parities][slot_elems]
    uint32_t *flags[P2P_MAX_RANKS];   // rank r's flag array: [2 parities][P2P_MAX_RANKS sources][P2P_MAX_WG]
    uint32_t *calls;                  // this rank: [P2P_MAX_WG] call counters
    uint32_t *timeouts;               // this rank: [1]
    int rank, world;
    int64_t slot_elems;
};

// OP 0: bf16 sum of `count` elements (count % 8 == 0); OP 1: u64 max of `count` elements (the vocab-parallel argmax keys)
template <int OP>
__global__ __launch_bounds__(256) void p2p_all_reduce_kernel(P2PArgs a, void *__restrict__ buf_v, int count) {
    __shared__ uint32_t epoch_s;
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (tid == 0) {   // this workgroup's call number: the same on every rank, because every rank makes the same calls
        const uint32_t e = a.calls[wg] + 1;
        a.calls[wg] = e;
        epoch_s = e;
    }
    __syncthreads();
    const uint32_t epoch = epoch_s;
    const int par = (int)(epoch & 1u);
    // 16-byte chunk of this thread: 8 bf16 or 2 u64; idx = its offset in 2-byte units (the staging buffer's element type)
    const int64_t chunk = (int64_t)wg * 256 + tid;
    const int64_t idx = chunk * 8;
    const int64_t nchunks = OP == 0 ? count / 8 : (count + 1) / 2;
    const bool live = chunk < nchunks;
    const bool odd_tail = OP == 1 && live && 2 * chunk + 1 >= count;   // odd u64 count: the last chunk holds one element
    uint16_t *buf = static_cast<uint16_t *>(buf_v);
    uint4 mine = make_uint4(0, 0, 0, 0);
    if (live) {
        if (odd_tail) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(buf + idx);
            mine = make_uint4(lo.x, lo.y, 0, 0);
        } else {
            mine = *reinterpret_cast<const uint4 *>(buf + idx);
        }
        *reinterpret_cast<uint4 *>(a.stage[a.rank] + par * a.slot_elems + idx) = mine;
    }
    __threadfence_system();   // the slice is in memory before any flag says so
    __syncthreads();
    if (tid < a.world && tid != a.rank)
        __hip_atomic_store(a.flags[tid] + ((size_t)par * P2P_MAX_RANKS + a.rank) * P2P_MAX_WG + wg, epoch, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < a.world && tid != a.rank) {
        const uint32_t *f = a.flags[a.rank] + ((size_t)par * P2P_MAX_RANKS + tid) * P2P_MAX_WG + wg;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        bool ok = false;
        for (;;) {
            const uint32_t v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int32_t)(v - epoch) >= 0) { ok = true; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) break;   // ~2 s: give up, never hang
            __builtin_amdgcn_s_sleep(4);
        }
        if (!ok) atomicAdd(a.timeouts, 1u);
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (system scope by default in HIP device code) peers' slices are read afresh
    if (!live) return;
    if (OP == 0) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < a.world; r++) {   // rank order: identical bits on every rank
            const uint4 v = r == a.rank ? mine : *reinterpret_cast<const uint4 *>(a.stage[r] + par * a.slot_elems + idx);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[2 * j] += __uint_as_float(w[j] << 16);
                acc[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u);
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (uint32_t)vmx_f32_to_bf16(acc[2 * j]) | ((uint32_t)vmx_f32_to_bf16(acc[2 * j + 1]) << 16);
        *reinterpret_cast<uint4 *>(buf + idx) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        unsigned long long m0 = ((unsigned long long)mine.y << 32) | mine.x, m1 = ((unsigned long long)mine.w << 32) | mine.z;
        for (int r = 0; r < a.world; r++) {
            if (r == a.rank) continue;
            const uint4 v = *reinterpret_cast<const uint4 *>(a.stage[r] + par * a.slot_elems + idx);
            const unsigned long long v0 = ((unsigned long long)v.y << 32) | v.x, v1 = ((unsigned long long)v.w << 32) | v.z;
            m0 = v0 > m0 ? v0 : m0;
            m1 = v1 > m1 ? v1 : m1;
        }
        if (odd_tail) *reinterpret_cast<uint2 *>(buf + idx) = make_uint2((uint32_t)m0, (uint32_t)(m0 >> 32));
        else *reinterpret_cast<uint4 *>(buf + idx) = make_uint4((uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32));
    }
}

struct P2PState {
    bool exported = false, ready = false;
    size_t max_bytes = 0;
    void *my_stage = nullptr, *my_flags = nullptr, *my_calls = nullptr;
    void *peer_stage[P2P_MAX_RANKS] = {}, *peer_flags[P2P_MAX_RANKS] = {};
    P2PArgs args = {};
};

}  // namespace

struct vmx_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
    P2PState p2p;
};

namespace {
int p2p_alloc(void **p, size_t bytes) {
    // fine-grained device memory: coherent for peers over the fabric; plain device memory if the flag is refused
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        VMX_CHECK_HIP(hipMalloc(p, bytes));
    }
    VMX_CHECK_HIP(hipMemset(*p, 0, bytes));
    return VMX_OK;
}
}  // namespace

// A communicator without RCCL: ranks that only use the peer-mapped path (and exchange the handles themselves).
extern "C" int vmx_comm_create_local(int rank, int world, vmx_comm **out) {
    VMX_REQUIRE(out, "comm_create_local: null pointer");
    VMX_REQUIRE(world >= 1 && world <= P2P_MAX_RANKS && rank >= 0 && rank < world, "comm_create_local: rank %d / world %d invalid (at most %d ranks)",
                rank, world, P2P_MAX_RANKS);
    vmx_comm *c = new vmx_comm();
    c->rank = rank;
    c->world = world;
    *out = c;
    return VMX_OK;
}

// Step 1 of the set-up, on every rank: allocate this rank's staging buffer (two slots of max_bytes) and flag array and
// return their IPC handles (2 x 64 bytes) for the caller to all-gather.
extern "C" int vmx_comm_p2p_export(vmx_comm *c, size_t max_bytes, void *handles_out128) {
    VMX_REQUIRE(c && handles_out128, "comm_p2p_export: null pointer");
    VMX_REQUIRE(c->world <= P2P_MAX_RANKS, "comm_p2p_export: at most %d ranks", P2P_MAX_RANKS);
    VMX_REQUIRE(!c->p2p.exported, "comm_p2p_export: already exported");
    VMX_REQUIRE(max_bytes >= 16 && max_bytes <= (size_t)P2P_MAX_WG * P2P_WG_ELEMS * 2, "comm_p2p_export: max_bytes %zu (16 .. %d)", max_bytes,
                P2P_MAX_WG * P2P_WG_ELEMS * 2);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "two handles must fit 128 bytes");
    P2PState &p = c->p2p;
    p.max_bytes = (max_bytes + 4095) & ~(size_t)4095;
    if (int rc = p2p_alloc(&p.my_stage, 2 * p.max_bytes)) return rc;
    if (int rc = p2p_alloc(&p.my_flags, (size_t)2 * P2P_MAX_RANKS * P2P_MAX_WG * 4)) return rc;
    if (int rc = p2p_alloc(&p.my_calls, (size_t)(P2P_MAX_WG + 1) * 4)) return rc;
    hipIpcMemHandle_t h[2];
    VMX_CHECK_HIP(hipIpcGetMemHandle(&h[0], p.my_stage));
    VMX_CHECK_HIP(hipIpcGetMemHandle(&h[1], p.my_flags));
    memcpy(handles_out128, h, 128);
    p.exported = true;
    return VMX_OK;
}

// Step 2, on every rank: all_handles = the world x 128 bytes of every rank's export, in rank order.
extern "C" int vmx_comm_p2p_import(vmx_comm *c, const void *all_handles) {
    VMX_REQUIRE(c && all_handles, "comm_p2p_import: null pointer");
    P2PState &p = c->p2p;
    VMX_REQUIRE(p.exported && !p.ready, "comm_p2p_import: call vmx_comm_p2p_export first (once)");
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) {
            p.peer_stage[r] = p.my_stage;
            p.peer_flags[r] = p.my_flags;
            continue;
        }
        hipIpcMemHandle_t h[2];
        memcpy(h, (const char *)all_handles + (size_t)r * 128, 128);
        VMX_CHECK_HIP(hipIpcOpenMemHandle(&p.peer_stage[r], h[0], hipIpcMemLazyEnablePeerAccess));
        VMX_CHECK_HIP(hipIpcOpenMemHandle(&p.peer_flags[r], h[1], hipIpcMemLazyEnablePeerAccess));
    }
    for (int r = 0; r < c->world; r++) {
        p.args.stage[r] = (uint16_t *)p.peer_stage[r];
        p.args.flags[r] = (uint32_t *)p.peer_flags[r];
    }
    p.args.calls = (uint32_t *)p.my_calls;
    p.args.timeouts = (uint32_t *)p.my_calls + P2P_MAX_WG;
    p.args.rank = c->rank;
    p.args.world = c->world;
    p.args.slot_elems = (int64_t)(p.max_bytes / 2);
    p.ready = true;
    return VMX_OK;
}

// Flag waits that gave up since the set-up (0 in a healthy job); synchronises the device.
extern "C" int vmx_comm_p2p_timeouts(vmx_comm *c, int *out) {
    VMX_REQUIRE(c && out && c->p2p.exported, "comm_p2p_timeouts: peer path not set up");
    uint32_t v = 0;
    VMX_CHECK_HIP(hipMemcpy(&v, (uint32_t *)c->p2p.my_calls + P2P_MAX_WG, 4, hipMemcpyDeviceToHost));
    *out = (int)v;
    return VMX_OK;
}

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
    P2PState &p = c->p2p;
    for (int r = 0; r < c->world && p.ready; r++)
        if (r != c->rank) {
            if (p.peer_stage[r]) (void)hipIpcCloseMemHandle(p.peer_stage[r]);
            if (p.peer_flags[r]) (void)hipIpcCloseMemHandle(p.peer_flags[r]);
        }
    for (void *q : {p.my_stage, p.my_flags, p.my_calls})
        if (q) (void)hipFree(q);
    delete c;
}

extern "C" int vmx_comm_rank(const vmx_comm *c) { return c ? c->rank : 0; }
extern "C" int vmx_comm_world(const vmx_comm *c) { return c ? c->world : 1; }

extern "C" int vmx_comm_all_reduce_sum_bf16(vmx_comm *c, uint16_t *buf, size_t count, void *stream) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    if (c->world == 1 && !c->comm) return VMX_OK;
    // decode-sized messages go over the peer-mapped path when it is set up (VMX_NO_P2P_ALLREDUCE=1 keeps them on RCCL)
    static const bool p2p_off = getenv("VMX_NO_P2P_ALLREDUCE") && atoi(getenv("VMX_NO_P2P_ALLREDUCE"));
    if (c->p2p.ready && (!p2p_off || !c->comm) && count > 0 && count % 8 == 0 && count * 2 <= c->p2p.max_bytes && ((uintptr_t)buf & 15) == 0) {
        const int grid = (int)((count + P2P_WG_ELEMS - 1) / P2P_WG_ELEMS);
        hipLaunchKernelGGL(p2p_all_reduce_kernel<0>, dim3(grid), dim3(256), 0, vmx_stream(stream), c->p2p.args, (void *)buf, (int)count);
        VMX_CHECK_LAUNCH("p2p_all_reduce");
        return VMX_OK;
    }
    VMX_REQUIRE(c->comm, "comm_all_reduce: %zu elements do not fit the peer-mapped path and this communicator has no RCCL side", count);
    return check_rccl(rccl().AllReduce(buf, buf, count, NCCL_BF16, NCCL_SUM, c->comm, vmx_stream(stream)), "ncclAllReduce(bf16,sum)");
}

extern "C" int vmx_comm_all_reduce_max_u64(vmx_comm *c, unsigned long long *buf, size_t count, void *stream) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    if (c->world == 1 && !c->comm) return VMX_OK;
    static const bool p2p_off = getenv("VMX_NO_P2P_ALLREDUCE") && atoi(getenv("VMX_NO_P2P_ALLREDUCE"));
    if (c->p2p.ready && (!p2p_off || !c->comm) && count > 0 && count * 8 <= c->p2p.max_bytes && count <= (size_t)P2P_MAX_WG * 512 &&
        ((uintptr_t)buf & 7) == 0) {
        const int grid = (int)(((count + 1) / 2 + 255) / 256);
        hipLaunchKernelGGL(p2p_all_reduce_kernel<1>, dim3(grid), dim3(256), 0, vmx_stream(stream), c->p2p.args, (void *)buf, (int)count);
        VMX_CHECK_LAUNCH("p2p_all_reduce(max)");
        return VMX_OK;
    }
    VMX_REQUIRE(c->comm, "comm_all_reduce: %zu keys do not fit the peer-mapped path and this communicator has no RCCL side", count);
    return check_rccl(rccl().AllReduce(buf, buf, count, NCCL_UINT64, NCCL_MAX, c->comm, vmx_stream(stream)), "ncclAllReduce(u64,max)");
}
