// Device communicator of the tensor- / expert-parallel group.  The reference resolves the nccl* symbols at run time with
// libloading (crates/core/src/distributed/nccl.rs:479-560) behind `DeviceCommunicator`
// (distributed/communicator.rs:29-94: all_reduce, all_gather, reduce_scatter, broadcast, send, recv, barrier, all_to_all,
// all_to_all_v); RCCL exports the same ABI, so this does the same with dlopen.  Every collective is enqueued on the caller's
// stream and is hipGraph-capturable.  Two transports:
//   * RCCL (ring / tree over xGMI) for any size;
//   * peer-mapped memory for the small, latency-bound messages of a decode step (8 KiB row-parallel sums 72 times per token,
//     argmax keys, expert dispatch of a few rows): every rank's staging buffer is mapped into every other rank through hipIpc
//     handles and a collective is ONE kernel (publish into the own slot, raise flags, wait for the peers' flags, read the
//     peers' slots).
// The choice between the two depends only on values that are equal on all ranks (set-up state agreed by the caller, message
// size), never on a pointer or on per-rank counts, so the ranks of a group always take the same transport.
#include "vmx_kernels.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <mutex>

namespace {

typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
enum { NCCL_SUM = 0, NCCL_PROD = 1, NCCL_MAX = 2, NCCL_MIN = 3, NCCL_AVG = 4 };
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_UINT32 = 3, NCCL_INT64 = 4, NCCL_UINT64 = 5, NCCL_F16 = 6, NCCL_F32 = 7, NCCL_F64 = 8, NCCL_BF16 = 9 };

struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_uid_t *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
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
        auto sym = [&](const char *n) { return dlsym(api.handle, n); };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.ReduceScatter = (decltype(api.ReduceScatter))sym("ncclReduceScatter");
        api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.AllGather && api.ReduceScatter &&
                 api.Broadcast && api.Send && api.Recv && api.GroupStart && api.GroupEnd;
    });
    return api;
}

int check_rccl(int rc, const char *what) {
    if (rc == 0) return VMX_OK;
    RcclApi &a = rccl();
    vmx_set_error("%s: RCCL error %d (%s)", what, rc, a.GetErrorString ? a.GetErrorString(rc) : "?");
    return VMX_ELAUNCH;
}

// ---------------------------------------------------------------- collectives over peer-mapped memory (xGMI)
// The decode path all-reduces one hidden-size row of bf16 (8 KiB on Qwen3-8B) 72 times per token; a ring collective pays
// 2 (N - 1) hops of launch + link latency for it.  With every rank's staging buffer mapped into every other rank the message
// fits one step (design reference: the one-shot kernel of crates/core/kernels/custom_allreduce.cu:264-484).
//
// All-reduce: each workgroup copies its 2048-element slice of the input into its own rank's staging slot, raises a flag in
// every peer's flag array, waits for the peers' flags for the same slice, then reads the slice from every rank's slot and adds
// them in rank order (the same order on every rank, so all ranks hold the same bits).  Exchange collectives (all-gather,
// reduce-scatter, broadcast, all-to-all(v), barrier): a fixed grid of XCH_WG workgroups publishes the rank's payload, every
// workgroup raises its flag in every peer and waits for ALL workgroups of ALL peers (a full barrier), then each rank reads what
// the collective gives it.  Slots and flags are double-buffered by the parity of a per-workgroup call counter kept in device
// memory (so a replayed hipGraph advances it): a rank can run at most one call ahead of its slowest peer and never overwrites
// a slot a peer may still be reading.
//
// Failure is loud and sticky.  A flag wait gives up after the time-out (2 s unless changed): the rank then stores an error
// code in its device state word and in a host-mapped mirror, poisons its peers (a word in their flag arrays that ends their
// waits at once), writes NaN instead of a sum and from then on does nothing but write NaN: it neither publishes nor reads, so
// no rank ever consumes a stale or half-written slot.  The host side reads the mirror without synchronising: the next vmx_*
// call on the communicator (or on an executor it is attached to) returns VMX_ELAUNCH.
constexpr int P2P_MAX_RANKS = 8, P2P_MAX_WG = 64, P2P_WG_ELEMS = 2048, XCH_WG = 32, XCH_HDR = 128;
// this rank's state words (device memory, u32): call counters of the all-reduce workgroups, time-out count, error code, call
// counters of the exchange workgroups
constexpr int ST_CALLS = 0, ST_TIMEOUTS = P2P_MAX_WG, ST_ERR = P2P_MAX_WG + 1, ST_XCALLS = P2P_MAX_WG + 2, ST_WORDS = ST_XCALLS + XCH_WG;
// flag words every rank exports (u32): [2 parities][source rank][workgroup] for the all-reduce, the same for the exchange
// kernels, one poison word
constexpr int F_AR = 0, F_X = 2 * P2P_MAX_RANKS * P2P_MAX_WG, F_POISON = F_X + 2 * P2P_MAX_RANKS * XCH_WG, F_WORDS = F_POISON + 1;
enum { P2P_ERR_TIMEOUT = 1, P2P_ERR_SIZE = 2, P2P_ERR_POISONED = 3 };

struct P2PArgs {
    uint8_t *stage[P2P_MAX_RANKS];    // rank r's staging allocation as mapped in this process: [2][ar_slot] then [2][x_slot] bytes
    uint32_t *flags[P2P_MAX_RANKS];   // rank r's flag words
    uint32_t *state;                  // this rank's state words
    uint32_t *err_host;               // host-mapped mirror of state[ST_ERR]
    unsigned long long timeout_ticks; // of the 100 MHz constant clock
    int rank, world;
    int64_t ar_slot, x_slot;          // bytes
};

__device__ __forceinline__ void p2p_fail(const P2PArgs &a, uint32_t code) {
    atomicCAS(a.state + ST_ERR, 0u, code);
    __hip_atomic_store(a.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int r = 0; r < a.world; r++)
        if (r != a.rank) __hip_atomic_store(a.flags[r] + F_POISON, 1u + (uint32_t)a.rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// At kernel entry (one thread per workgroup): the rank's sticky error, or "poisoned" when a peer has failed since.
__device__ __forceinline__ uint32_t p2p_entry_state(const P2PArgs &a) {
    uint32_t bad = __hip_atomic_load(a.state + ST_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bad == 0u && __hip_atomic_load(a.flags[a.rank] + F_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
        bad = P2P_ERR_POISONED;
        p2p_fail(a, bad);
    }
    return bad;
}

// Wait until *f has reached `epoch`; false when the time-out passed or a peer poisoned this rank.
__device__ __forceinline__ bool p2p_wait(const P2PArgs &a, const uint32_t *f, uint32_t epoch, uint32_t *why) {
    const uint32_t *poison = a.flags[a.rank] + F_POISON;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - epoch) >= 0) return true;
        if (__hip_atomic_load(poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { *why = P2P_ERR_POISONED; return false; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) { *why = P2P_ERR_TIMEOUT; return false; }
        __builtin_amdgcn_s_sleep(4);
    }
}

// OP 0: bf16 sum of `count` elements (count % 8 == 0); OP 1: u64 max of `count` elements (the vocab-parallel argmax keys);
// OP 2: fp16 sum (as OP 0: f32 accumulation in rank order, one rounding)
template <int OP>
__global__ __launch_bounds__(256) void p2p_all_reduce_kernel(P2PArgs a, void *__restrict__ buf_v, int count) {
    __shared__ uint32_t epoch_s, bad_s;
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (tid == 0) {   // this workgroup's call number: the same on every rank, because every rank makes the same calls
        const uint32_t e = a.state[ST_CALLS + wg] + 1;
        a.state[ST_CALLS + wg] = e;
        epoch_s = e;
        bad_s = p2p_entry_state(a);
    }
    __syncthreads();
    const uint32_t epoch = epoch_s;
    const int par = (int)(epoch & 1u);
    // 16-byte chunk of this thread: 8 bf16 or 2 u64; idx = its offset in 2-byte units
    const int64_t chunk = (int64_t)wg * 256 + tid;
    const int64_t idx = chunk * 8;
    constexpr bool SUM = OP == 0 || OP == 2;
    const int64_t nchunks = SUM ? count / 8 : (count + 1) / 2;
    const bool live = chunk < nchunks;
    const bool odd_tail = OP == 1 && live && 2 * chunk + 1 >= count;   // odd u64 count: the last chunk holds one element
    uint16_t *buf = static_cast<uint16_t *>(buf_v);
    const uint32_t nan2 = OP == 2 ? 0x7E007E00u : 0x7FC07FC0u;   // two quiet NaNs of the element type
    const uint4 nan4 = make_uint4(nan2, nan2, nan2, nan2);
    if (bad_s != 0u) {   // a failed rank takes no further part; a sum becomes NaN, the keys keep the local maximum
        if (SUM && live) *reinterpret_cast<uint4 *>(buf + idx) = nan4;
        return;
    }
    uint4 mine = make_uint4(0, 0, 0, 0);
    if (live) {
        if (odd_tail) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(buf + idx);
            mine = make_uint4(lo.x, lo.y, 0, 0);
        } else {
            mine = *reinterpret_cast<const uint4 *>(buf + idx);
        }
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(a.stage[a.rank] + par * a.ar_slot) + idx) = mine;
    }
    __threadfence_system();   // the slice is in memory before any flag says so
    __syncthreads();
    if (tid < a.world && tid != a.rank) {
        __hip_atomic_store(a.flags[tid] + F_AR + ((size_t)par * P2P_MAX_RANKS + a.rank) * P2P_MAX_WG + wg, epoch, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        uint32_t why = 0;
        if (!p2p_wait(a, a.flags[a.rank] + F_AR + ((size_t)par * P2P_MAX_RANKS + tid) * P2P_MAX_WG + wg, epoch, &why)) {
            if (why == P2P_ERR_TIMEOUT) atomicAdd(a.state + ST_TIMEOUTS, 1u);
            p2p_fail(a, why);
            bad_s = why;
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (system scope by default in HIP device code) peers' slices are read afresh
    if (!live) return;
    if (bad_s != 0u) {
        if (SUM) *reinterpret_cast<uint4 *>(buf + idx) = nan4;
        return;
    }
    if (SUM) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < a.world; r++) {   // rank order: identical bits on every rank
            const uint4 v = r == a.rank ? mine : *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.stage[r] + par * a.ar_slot) + idx);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[2 * j] += vmx_ld16<OP == 2>((uint16_t)(w[j] & 0xFFFFu));
                acc[2 * j + 1] += vmx_ld16<OP == 2>((uint16_t)(w[j] >> 16));
            }
        }
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (uint32_t)vmx_st16<OP == 2>(acc[2 * j]) | ((uint32_t)vmx_st16<OP == 2>(acc[2 * j + 1]) << 16);
        *reinterpret_cast<uint4 *>(buf + idx) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
        unsigned long long m0 = ((unsigned long long)mine.y << 32) | mine.x, m1 = ((unsigned long long)mine.w << 32) | mine.z;
        for (int r = 0; r < a.world; r++) {
            if (r == a.rank) continue;
            const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.stage[r] + par * a.ar_slot) + idx);
            const unsigned long long v0 = ((unsigned long long)v.y << 32) | v.x, v1 = ((unsigned long long)v.w << 32) | v.z;
            m0 = v0 > m0 ? v0 : m0;
            m1 = v1 > m1 ? v1 : m1;
        }
        if (odd_tail) *reinterpret_cast<uint2 *>(buf + idx) = make_uint2((uint32_t)m0, (uint32_t)(m0 >> 32));
        else *reinterpret_cast<uint4 *>(buf + idx) = make_uint4((uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32));
    }
}

// ---- exchange collectives
enum { X_ALL_GATHER = 0, X_REDUCE_SCATTER = 1, X_BROADCAST = 2, X_ALL_TO_ALL_V = 3, X_BARRIER = 4, X_ALL_TO_ALL_V_DEV = 5 };

struct XchOp {
    int mode, root, dtype;     // dtype of a reduce-scatter: 0 = bf16, 1 = f32 (sum, f32 accumulation in rank order)
    const uint8_t *in;
    uint8_t *out;
    int64_t bytes;             // all-gather / broadcast / reduce-scatter: bytes of one rank's chunk
    int64_t send_off[P2P_MAX_RANKS + 1], recv_off[P2P_MAX_RANKS + 1];   // all_to_all_v: byte offsets of the per-rank segments in `in` / `out`
    // all_to_all_v with device-resident counts: rows of row_bytes, send_rows[world] read by the kernel, recv_rows[world] written
    const int32_t *send_rows;
    int32_t *recv_rows;
    int64_t row_bytes, out_capacity_rows;
};

// Grid-stride copy of n bytes by the whole launch; 16-byte pieces when everything is aligned, else 4-byte, else bytes.
__device__ __forceinline__ void xch_copy(uint8_t *dst, const uint8_t *src, int64_t n, int64_t gtid, int64_t gsize) {
    const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)n;
    if ((al & 15) == 0) {
        for (int64_t i = gtid; i < (n >> 4); i += gsize) reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    } else if ((al & 3) == 0) {
        for (int64_t i = gtid; i < (n >> 2); i += gsize) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
    } else {
        for (int64_t i = gtid; i < n; i += gsize) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void p2p_exchange_kernel(P2PArgs a, XchOp op) {
    __shared__ uint32_t epoch_s, bad_s;
    __shared__ int64_t soff_s[P2P_MAX_RANKS + 1];
    const int tid = threadIdx.x, wg = blockIdx.x;
    const int64_t gtid = (int64_t)wg * 256 + tid, gsize = (int64_t)XCH_WG * 256;
    if (tid == 0) {
        const uint32_t e = a.state[ST_XCALLS + wg] + 1;
        a.state[ST_XCALLS + wg] = e;
        epoch_s = e;
        bad_s = p2p_entry_state(a);
        if (op.mode == X_ALL_TO_ALL_V) {
            for (int r = 0; r <= a.world; r++) soff_s[r] = op.send_off[r];
        } else if (op.mode == X_ALL_TO_ALL_V_DEV) {
            int64_t o = 0;
            for (int r = 0; r < a.world; r++) {
                soff_s[r] = o;
                o += (int64_t)max(op.send_rows[r], 0) * op.row_bytes;
            }
            soff_s[a.world] = o;
        }
    }
    __syncthreads();
    if (bad_s != 0u) return;   // a failed rank takes no further part
    const uint32_t epoch = epoch_s;
    const int par = (int)(epoch & 1u);
    const int64_t xbase = 2 * a.ar_slot + par * a.x_slot;
    uint8_t *mine = a.stage[a.rank] + xbase;
    const bool a2a = op.mode == X_ALL_TO_ALL_V || op.mode == X_ALL_TO_ALL_V_DEV;
    // ---- publish
    if (op.mode == X_ALL_GATHER) xch_copy(mine, op.in, op.bytes, gtid, gsize);
    else if (op.mode == X_REDUCE_SCATTER) xch_copy(mine, op.in, op.bytes * a.world, gtid, gsize);
    else if (op.mode == X_BROADCAST) { if (a.rank == op.root) xch_copy(mine, op.in, op.bytes, gtid, gsize); }
    else if (a2a) {
        if (wg == 0 && tid <= a.world) reinterpret_cast<int64_t *>(mine)[tid] = soff_s[tid];
        const int64_t total = min(soff_s[a.world], a.x_slot - XCH_HDR);   // the host checked the size; device counts are clamped and reported below
        xch_copy(mine + XCH_HDR, op.in, total, gtid, gsize);
        if (op.mode == X_ALL_TO_ALL_V_DEV && gtid == 0 && soff_s[a.world] > a.x_slot - XCH_HDR) {
            p2p_fail(a, P2P_ERR_SIZE);
        }
    }
    __threadfence_system();
    __syncthreads();
    // ---- every workgroup tells every peer, then waits for every workgroup of every peer
    if (tid < a.world && tid != a.rank)
        __hip_atomic_store(a.flags[tid] + F_X + ((size_t)par * P2P_MAX_RANKS + a.rank) * XCH_WG + wg, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    {
        const int peer = tid / XCH_WG, w = tid % XCH_WG;   // 256 threads = 8 peers x 32 workgroups
        if (peer < a.world && peer != a.rank) {
            uint32_t why = 0;
            if (!p2p_wait(a, a.flags[a.rank] + F_X + ((size_t)par * P2P_MAX_RANKS + peer) * XCH_WG + w, epoch, &why)) {
                if (why == P2P_ERR_TIMEOUT) atomicAdd(a.state + ST_TIMEOUTS, 1u);
                p2p_fail(a, why);
                bad_s = why;
            }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (bad_s != 0u) return;
    // ---- consume
    if (op.mode == X_ALL_GATHER) {
        for (int r = 0; r < a.world; r++) xch_copy(op.out + (int64_t)r * op.bytes, r == a.rank ? op.in : a.stage[r] + xbase, op.bytes, gtid, gsize);
    } else if (op.mode == X_BROADCAST) {
        if (a.rank != op.root) xch_copy(op.out, a.stage[op.root] + xbase, op.bytes, gtid, gsize);
        else if (op.out != op.in) xch_copy(op.out, op.in, op.bytes, gtid, gsize);
    } else if (op.mode == X_REDUCE_SCATTER) {
        const int64_t off = (int64_t)a.rank * op.bytes;
        for (int64_t i = gtid; i < (op.bytes >> 4); i += gsize) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < a.world; r++) {
                const uint8_t *src = r == a.rank ? op.in : a.stage[r] + xbase;
                const uint4 v = reinterpret_cast<const uint4 *>(src + off)[i];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (op.dtype == 0) {
                        acc[2 * j] += __uint_as_float(w[j] << 16);
                        acc[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u);
                    } else {
                        acc[j] += __uint_as_float(w[j]);
                    }
                }
            }
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                o[j] = op.dtype == 0 ? ((uint32_t)vmx_f32_to_bf16(acc[2 * j]) | ((uint32_t)vmx_f32_to_bf16(acc[2 * j + 1]) << 16)) : __float_as_uint(acc[j]);
            reinterpret_cast<uint4 *>(op.out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else if (a2a) {
        int64_t roff = 0;   // running byte offset of the device-count form
        for (int r = 0; r < a.world; r++) {
            const int64_t *hdr = reinterpret_cast<const int64_t *>(a.stage[r] + xbase);
            const int64_t o0 = r == a.rank ? soff_s[a.rank] : hdr[a.rank], o1 = r == a.rank ? soff_s[a.rank + 1] : hdr[a.rank + 1];
            int64_t n = o1 - o0;
            const uint8_t *src = r == a.rank ? op.in + o0 : a.stage[r] + xbase + XCH_HDR + o0;
            if (op.mode == X_ALL_TO_ALL_V) {
                if (n != op.recv_off[r + 1] - op.recv_off[r]) {   // the peer sends another amount than this rank expects
                    if (gtid == 0) p2p_fail(a, P2P_ERR_SIZE);
                    continue;
                }
                xch_copy(op.out + op.recv_off[r], src, n, gtid, gsize);
            } else {
                const int64_t room = op.out_capacity_rows * op.row_bytes - roff;
                if (n > room) {
                    if (gtid == 0) p2p_fail(a, P2P_ERR_SIZE);
                    n = room > 0 ? room / op.row_bytes * op.row_bytes : 0;
                }
                xch_copy(op.out + roff, src, n, gtid, gsize);
                if (gtid == 0) op.recv_rows[r] = (int32_t)(n / op.row_bytes);
                roff += n;
            }
        }
    }
}

struct P2PState {
    bool exported = false, ready = false;
    size_t ar_slot = 0, x_slot = 0;
    void *my_stage = nullptr, *my_flags = nullptr, *my_state = nullptr;
    uint32_t *err_host = nullptr;   // host-mapped (hipHostMalloc)
    void *peer_stage[P2P_MAX_RANKS] = {}, *peer_flags[P2P_MAX_RANKS] = {};
    P2PArgs args = {};
};

}  // namespace

struct vmx_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
    void *barrier_word = nullptr;   // 4 bytes of device memory for the RCCL barrier
    P2PState p2p;
};

namespace {
// Fine-grained device memory: coherent for peers over the fabric.  No coarse-grained fallback: a kernel polling and reading
// another GPU's plain device memory may keep seeing its own L2's copy.
int p2p_alloc(void **p, size_t bytes) {
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        vmx_set_error("comm_p2p_export: fine-grained device memory is not available on this device; keep this group on RCCL");
        return VMX_EUNSUPPORTED;
    }
    VMX_CHECK_HIP(hipMemset(*p, 0, bytes));
    return VMX_OK;
}

void p2p_release(vmx_comm *c) {
    P2PState &p = c->p2p;
    for (int r = 0; r < c->world && p.ready; r++)
        if (r != c->rank) {
            if (p.peer_stage[r]) (void)hipIpcCloseMemHandle(p.peer_stage[r]);
            if (p.peer_flags[r]) (void)hipIpcCloseMemHandle(p.peer_flags[r]);
        }
    for (void *q : {p.my_stage, p.my_flags, p.my_state})
        if (q) (void)hipFree(q);
    if (p.err_host) (void)hipHostFree(p.err_host);
    p = P2PState();
}

// The sticky error of the peer-mapped transport, read from the host-mapped mirror (no synchronisation, no device call).
int comm_check(const vmx_comm *c, const char *what) {
    if (!c || !c->p2p.err_host) return VMX_OK;
    const uint32_t code = *(volatile const uint32_t *)c->p2p.err_host;
    if (code == 0) return VMX_OK;
    vmx_set_error("%s: the peer-mapped transport of rank %d failed earlier (%s); results since then are invalid", what, c->rank,
                  code == P2P_ERR_TIMEOUT ? "a peer did not arrive within the time-out"
                  : code == P2P_ERR_SIZE  ? "all_to_all_v sizes of two ranks disagree or exceed the staging buffer"
                                          : "a peer rank reported a failure");
    return VMX_ELAUNCH;
}

int launch_exchange(vmx_comm *c, const XchOp &op, void *stream, const char *what) {
    hipLaunchKernelGGL(p2p_exchange_kernel, dim3(XCH_WG), dim3(256), 0, vmx_stream(stream), c->p2p.args, op);
    VMX_CHECK_LAUNCH(what);
    return VMX_OK;
}

bool x_fits(const vmx_comm *c, size_t bytes) { return c->p2p.ready && c->p2p.x_slot >= bytes + XCH_HDR; }
}  // namespace

// A communicator without RCCL: ranks that only use the peer-mapped transport (and exchange the handles themselves).
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

// Step 1 of the set-up, on every rank: allocate this rank's staging (two all-reduce slots of max_bytes and two exchange slots
// of exchange_bytes) and flag words and return their IPC handles (2 x 64 bytes) for the caller to all-gather.
extern "C" int vmx_comm_p2p_export(vmx_comm *c, size_t max_bytes, size_t exchange_bytes, void *handles_out128) {
    VMX_REQUIRE(c && handles_out128, "comm_p2p_export: null pointer");
    VMX_REQUIRE(c->world <= P2P_MAX_RANKS, "comm_p2p_export: at most %d ranks", P2P_MAX_RANKS);
    VMX_REQUIRE(!c->p2p.exported, "comm_p2p_export: already exported");
    VMX_REQUIRE(max_bytes >= 16 && max_bytes <= (size_t)P2P_MAX_WG * P2P_WG_ELEMS * 2, "comm_p2p_export: max_bytes %zu (16 .. %d)", max_bytes,
                P2P_MAX_WG * P2P_WG_ELEMS * 2);
    VMX_REQUIRE(exchange_bytes <= ((size_t)1 << 30), "comm_p2p_export: exchange_bytes %zu (at most 1 GiB)", exchange_bytes);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "two handles must fit 128 bytes");
    P2PState &p = c->p2p;
    p.ar_slot = (max_bytes + 4095) & ~(size_t)4095;
    p.x_slot = exchange_bytes ? ((exchange_bytes + XCH_HDR + 4095) & ~(size_t)4095) : 0;
    int rc = p2p_alloc(&p.my_stage, 2 * p.ar_slot + 2 * p.x_slot);
    if (!rc) rc = p2p_alloc(&p.my_flags, (size_t)((F_WORDS + 1023) & ~1023) * 4);
    if (!rc) rc = p2p_alloc(&p.my_state, (size_t)ST_WORDS * 4);
    if (!rc && hipHostMalloc((void **)&p.err_host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        p.err_host = nullptr;
        vmx_set_error("comm_p2p_export: host-mapped memory for the error word is not available");
        rc = VMX_EUNSUPPORTED;
    }
    hipIpcMemHandle_t h[2];
    if (!rc && (hipIpcGetMemHandle(&h[0], p.my_stage) != hipSuccess || hipIpcGetMemHandle(&h[1], p.my_flags) != hipSuccess)) {
        (void)hipGetLastError();
        vmx_set_error("comm_p2p_export: hipIpcGetMemHandle failed");
        rc = VMX_EUNSUPPORTED;
    }
    if (rc) {
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", vmx_last_error());
        p2p_release(c);
        vmx_set_error("%s", msg);
        return rc;
    }
    *p.err_host = 0;
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
        p.args.stage[r] = (uint8_t *)p.peer_stage[r];
        p.args.flags[r] = (uint32_t *)p.peer_flags[r];
    }
    p.args.state = (uint32_t *)p.my_state;
    void *err_dev = nullptr;
    VMX_CHECK_HIP(hipHostGetDevicePointer(&err_dev, p.err_host, 0));
    p.args.err_host = (uint32_t *)err_dev;
    p.args.timeout_ticks = 200000000ull;   // 2 s of the 100 MHz clock
    p.args.rank = c->rank;
    p.args.world = c->world;
    p.args.ar_slot = (int64_t)p.ar_slot;
    p.args.x_slot = (int64_t)p.x_slot;
    p.ready = true;
    return VMX_OK;
}

// Drop the peer-mapped transport (every rank of the group must do the same): collectives go back to RCCL.
extern "C" int vmx_comm_p2p_disable(vmx_comm *c) {
    VMX_REQUIRE(c, "comm_p2p_disable: null pointer");
    VMX_CHECK_HIP(hipDeviceSynchronize());
    p2p_release(c);
    return VMX_OK;
}

extern "C" int vmx_comm_p2p_set_timeout_ms(vmx_comm *c, int ms) {
    VMX_REQUIRE(c && c->p2p.ready, "comm_p2p_set_timeout_ms: peer-mapped transport not set up");
    VMX_REQUIRE(ms >= 1 && ms <= 600000, "comm_p2p_set_timeout_ms: %d ms (1 .. 600000)", ms);
    c->p2p.args.timeout_ticks = (unsigned long long)ms * 100000ull;
    return VMX_OK;
}

// Flag waits that gave up since the set-up (0 in a healthy job); synchronises the device.
extern "C" int vmx_comm_p2p_timeouts(vmx_comm *c, int *out) {
    VMX_REQUIRE(c && out, "comm_p2p_timeouts: null pointer");
    *out = 0;
    if (!c->p2p.exported) return VMX_OK;   // RCCL only: nothing to count
    uint32_t v = 0;
    VMX_CHECK_HIP(hipMemcpy(&v, (uint32_t *)c->p2p.my_state + ST_TIMEOUTS, 4, hipMemcpyDeviceToHost));
    *out = (int)v;
    return VMX_OK;
}

// Sticky error code of the peer-mapped transport (0 = healthy, 1 = time-out, 2 = size disagreement, 3 = a peer failed);
// a plain host read, usable between graph replays.
extern "C" int vmx_comm_error(const vmx_comm *c, int *code_out) {
    VMX_REQUIRE(c && code_out, "comm_error: null pointer");
    *code_out = c->p2p.err_host ? (int)*(volatile const uint32_t *)c->p2p.err_host : 0;
    return VMX_OK;
}

extern "C" int vmx_comm_p2p_ready(const vmx_comm *c) { return c && c->p2p.ready ? 1 : 0; }

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
    if (hipMalloc(&c->barrier_word, 256) != hipSuccess || hipMemset(c->barrier_word, 0, 256) != hipSuccess) {
        (void)hipGetLastError();
        (void)a.CommDestroy(c->comm);
        delete c;
        vmx_set_error("comm_create: device allocation failed");
        return VMX_ENOMEM;
    }
    *out = c;
    return VMX_OK;
}

extern "C" void vmx_comm_destroy(vmx_comm *c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->barrier_word) (void)hipFree(c->barrier_word);
    p2p_release(c);
    delete c;
}

extern "C" int vmx_comm_rank(const vmx_comm *c) { return c ? c->rank : 0; }
extern "C" int vmx_comm_world(const vmx_comm *c) { return c ? c->world : 1; }

static int all_reduce_sum16(vmx_comm *c, uint16_t *buf, size_t count, void *stream, bool f16) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    if (int rc = comm_check(c, "comm_all_reduce")) return rc;
    if (c->world == 1 && !c->comm) return VMX_OK;
    // decode-sized messages go over the peer-mapped transport when it is set up
    if (c->p2p.ready && count > 0 && count % 8 == 0 && count * 2 <= c->p2p.ar_slot) {
        VMX_REQUIRE(((uintptr_t)buf & 15) == 0, "comm_all_reduce: the buffer must be 16-byte aligned");
        const int grid = (int)((count + P2P_WG_ELEMS - 1) / P2P_WG_ELEMS);
        if (f16) hipLaunchKernelGGL(p2p_all_reduce_kernel<2>, dim3(grid), dim3(256), 0, vmx_stream(stream), c->p2p.args, (void *)buf, (int)count);
        else hipLaunchKernelGGL(p2p_all_reduce_kernel<0>, dim3(grid), dim3(256), 0, vmx_stream(stream), c->p2p.args, (void *)buf, (int)count);
        VMX_CHECK_LAUNCH("p2p_all_reduce");
        return VMX_OK;
    }
    VMX_REQUIRE(c->comm, "comm_all_reduce: %zu elements do not fit the peer-mapped transport and this communicator has no RCCL side", count);
    return check_rccl(rccl().AllReduce(buf, buf, count, f16 ? VMX_COMM_F16 : NCCL_BF16, NCCL_SUM, c->comm, vmx_stream(stream)),
                      f16 ? "ncclAllReduce(f16,sum)" : "ncclAllReduce(bf16,sum)");
}

extern "C" int vmx_comm_all_reduce_sum_bf16(vmx_comm *c, uint16_t *buf, size_t count, void *stream) {
    return all_reduce_sum16(c, buf, count, stream, false);
}

// The same sum over fp16 elements: the row-parallel outputs of a model that runs in fp16 (EXL3 checkpoints)
extern "C" int vmx_comm_all_reduce_sum_f16(vmx_comm *c, uint16_t *buf, size_t count, void *stream) {
    return all_reduce_sum16(c, buf, count, stream, true);
}

extern "C" int vmx_comm_all_reduce_max_u64(vmx_comm *c, unsigned long long *buf, size_t count, void *stream) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    if (int rc = comm_check(c, "comm_all_reduce")) return rc;
    if (c->world == 1 && !c->comm) return VMX_OK;
    if (c->p2p.ready && count > 0 && count * 8 <= c->p2p.ar_slot && count <= (size_t)P2P_MAX_WG * 512) {
        VMX_REQUIRE(((uintptr_t)buf & 7) == 0, "comm_all_reduce: the key buffer must be 8-byte aligned");
        const int grid = (int)(((count + 1) / 2 + 255) / 256);
        hipLaunchKernelGGL(p2p_all_reduce_kernel<1>, dim3(grid), dim3(256), 0, vmx_stream(stream), c->p2p.args, (void *)buf, (int)count);
        VMX_CHECK_LAUNCH("p2p_all_reduce(max)");
        return VMX_OK;
    }
    VMX_REQUIRE(c->comm, "comm_all_reduce: %zu keys do not fit the peer-mapped transport and this communicator has no RCCL side", count);
    return check_rccl(rccl().AllReduce(buf, buf, count, NCCL_UINT64, NCCL_MAX, c->comm, vmx_stream(stream)), "ncclAllReduce(u64,max)");
}

// all_reduce of any RCCL data type and operation (ReduceOp: Sum, Product, Min, Max, Average), in place; RCCL only.
extern "C" int vmx_comm_all_reduce(vmx_comm *c, void *buf, size_t count, int dtype, int op, void *stream) {
    VMX_REQUIRE(c && buf, "comm_all_reduce: null pointer");
    VMX_REQUIRE(dtype >= VMX_COMM_I8 && dtype <= VMX_COMM_BF16 && op >= VMX_COMM_SUM && op <= VMX_COMM_AVG, "comm_all_reduce: dtype %d / op %d invalid", dtype, op);
    if (int rc = comm_check(c, "comm_all_reduce")) return rc;
    if (c->world == 1 && !c->comm) return VMX_OK;
    if (dtype == VMX_COMM_BF16 && op == VMX_COMM_SUM) return vmx_comm_all_reduce_sum_bf16(c, (uint16_t *)buf, count, stream);
    if (dtype == VMX_COMM_F16 && op == VMX_COMM_SUM) return vmx_comm_all_reduce_sum_f16(c, (uint16_t *)buf, count, stream);
    if (dtype == VMX_COMM_U64 && op == VMX_COMM_MAX) return vmx_comm_all_reduce_max_u64(c, (unsigned long long *)buf, count, stream);
    VMX_REQUIRE(c->comm, "comm_all_reduce: dtype %d / op %d needs the RCCL side of the communicator", dtype, op);
    return check_rccl(rccl().AllReduce(buf, buf, count, dtype, op, c->comm, vmx_stream(stream)), "ncclAllReduce");
}

extern "C" int vmx_comm_all_gather(vmx_comm *c, const void *in, void *out, size_t bytes_per_rank, void *stream) {
    VMX_REQUIRE(c && in && out, "comm_all_gather: null pointer");
    if (int rc = comm_check(c, "comm_all_gather")) return rc;
    if (c->world == 1 && !c->comm) {
        if (in != out) VMX_CHECK_HIP(hipMemcpyAsync(out, in, bytes_per_rank, hipMemcpyDeviceToDevice, vmx_stream(stream)));
        return VMX_OK;
    }
    if (x_fits(c, bytes_per_rank) && (!c->comm || bytes_per_rank <= (size_t)256 * 1024)) {
        XchOp op = {};
        op.mode = X_ALL_GATHER;
        op.in = (const uint8_t *)in;
        op.out = (uint8_t *)out;
        op.bytes = (int64_t)bytes_per_rank;
        return launch_exchange(c, op, stream, "p2p_all_gather");
    }
    VMX_REQUIRE(c->comm, "comm_all_gather: %zu bytes per rank do not fit the peer-mapped transport and this communicator has no RCCL side", bytes_per_rank);
    return check_rccl(rccl().AllGather(in, out, bytes_per_rank, NCCL_UINT8, c->comm, vmx_stream(stream)), "ncclAllGather");
}

// out[count_per_rank] = sum over ranks of in[rank * count_per_rank ...]; dtype VMX_COMM_BF16 or VMX_COMM_F32
extern "C" int vmx_comm_reduce_scatter_sum(vmx_comm *c, const void *in, void *out, size_t count_per_rank, int dtype, void *stream) {
    VMX_REQUIRE(c && in && out, "comm_reduce_scatter: null pointer");
    VMX_REQUIRE(dtype == VMX_COMM_BF16 || dtype == VMX_COMM_F32, "comm_reduce_scatter: dtype %d (bf16 or f32)", dtype);
    if (int rc = comm_check(c, "comm_reduce_scatter")) return rc;
    const size_t es = dtype == VMX_COMM_BF16 ? 2 : 4, bytes = count_per_rank * es;
    if (c->world == 1 && !c->comm) {
        if (in != out) VMX_CHECK_HIP(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, vmx_stream(stream)));
        return VMX_OK;
    }
    if (x_fits(c, bytes * c->world) && bytes % 16 == 0 && (!c->comm || bytes * c->world <= (size_t)256 * 1024)) {
        VMX_REQUIRE((((uintptr_t)in | (uintptr_t)out) & 15) == 0, "comm_reduce_scatter: buffers must be 16-byte aligned");
        XchOp op = {};
        op.mode = X_REDUCE_SCATTER;
        op.dtype = dtype == VMX_COMM_BF16 ? 0 : 1;
        op.in = (const uint8_t *)in;
        op.out = (uint8_t *)out;
        op.bytes = (int64_t)bytes;
        return launch_exchange(c, op, stream, "p2p_reduce_scatter");
    }
    VMX_REQUIRE(c->comm, "comm_reduce_scatter: %zu bytes do not fit the peer-mapped transport (16-byte multiples) and this communicator has no RCCL side", bytes);
    return check_rccl(rccl().ReduceScatter(in, out, count_per_rank, dtype, NCCL_SUM, c->comm, vmx_stream(stream)), "ncclReduceScatter");
}

extern "C" int vmx_comm_broadcast(vmx_comm *c, void *buf, size_t bytes, int root, void *stream) {
    VMX_REQUIRE(c && buf, "comm_broadcast: null pointer");
    VMX_REQUIRE(root >= 0 && root < c->world, "comm_broadcast: root %d of %d ranks", root, c->world);
    if (int rc = comm_check(c, "comm_broadcast")) return rc;
    if (c->world == 1 && !c->comm) return VMX_OK;
    if (x_fits(c, bytes) && (!c->comm || bytes <= (size_t)256 * 1024)) {
        XchOp op = {};
        op.mode = X_BROADCAST;
        op.root = root;
        op.in = (const uint8_t *)buf;
        op.out = (uint8_t *)buf;
        op.bytes = (int64_t)bytes;
        return launch_exchange(c, op, stream, "p2p_broadcast");
    }
    VMX_REQUIRE(c->comm, "comm_broadcast: %zu bytes do not fit the peer-mapped transport and this communicator has no RCCL side", bytes);
    return check_rccl(rccl().Broadcast(buf, buf, bytes, NCCL_UINT8, root, c->comm, vmx_stream(stream)), "ncclBroadcast");
}

extern "C" int vmx_comm_send(vmx_comm *c, const void *buf, size_t bytes, int dst, void *stream) {
    VMX_REQUIRE(c && buf, "comm_send: null pointer");
    VMX_REQUIRE(dst >= 0 && dst < c->world && dst != c->rank, "comm_send: destination %d (this is rank %d of %d)", dst, c->rank, c->world);
    VMX_REQUIRE(c->comm, "comm_send: point-to-point transfers need the RCCL side of the communicator");
    return check_rccl(rccl().Send(buf, bytes, NCCL_UINT8, dst, c->comm, vmx_stream(stream)), "ncclSend");
}

extern "C" int vmx_comm_recv(vmx_comm *c, void *buf, size_t bytes, int src, void *stream) {
    VMX_REQUIRE(c && buf, "comm_recv: null pointer");
    VMX_REQUIRE(src >= 0 && src < c->world && src != c->rank, "comm_recv: source %d (this is rank %d of %d)", src, c->rank, c->world);
    VMX_REQUIRE(c->comm, "comm_recv: point-to-point transfers need the RCCL side of the communicator");
    return check_rccl(rccl().Recv(buf, bytes, NCCL_UINT8, src, c->comm, vmx_stream(stream)), "ncclRecv");
}

// Orders the stream behind every rank's arrival (the reference's barrier is an all-reduce of one element, nccl.rs).
extern "C" int vmx_comm_barrier(vmx_comm *c, void *stream) {
    VMX_REQUIRE(c, "comm_barrier: null pointer");
    if (int rc = comm_check(c, "comm_barrier")) return rc;
    if (c->world == 1 && !c->comm) return VMX_OK;
    if (c->p2p.ready) {
        XchOp op = {};
        op.mode = X_BARRIER;
        return launch_exchange(c, op, stream, "p2p_barrier");
    }
    VMX_REQUIRE(c->comm, "comm_barrier: no transport");
    return check_rccl(rccl().AllReduce(c->barrier_word, c->barrier_word, 1, NCCL_INT32, NCCL_SUM, c->comm, vmx_stream(stream)), "ncclAllReduce(barrier)");
}

// all_to_all_v: segment r of `in` (send_counts[r] elements) goes to rank r; segment r of `out` (recv_counts[r] elements) comes
// from rank r.  With an RCCL side: grouped send / recv as distributed/nccl.rs:939-1007.  Without one (local communicator):
// the peer-mapped exchange; a rank whose recv_counts disagree with what its peer sends fails the group (error code 2).
extern "C" int vmx_comm_all_to_all_v(vmx_comm *c, const void *in, const size_t *send_counts, void *out, const size_t *recv_counts,
                                     size_t elem_bytes, void *stream) {
    VMX_REQUIRE(c && send_counts && recv_counts && elem_bytes > 0, "comm_all_to_all_v: null pointer");
    if (int rc = comm_check(c, "comm_all_to_all_v")) return rc;
    size_t soff[P2P_MAX_RANKS + 1] = {0}, roff[P2P_MAX_RANKS + 1] = {0};
    VMX_REQUIRE(c->world <= P2P_MAX_RANKS || c->comm, "comm_all_to_all_v: at most %d ranks without RCCL", P2P_MAX_RANKS);
    if (c->world == 1) {
        VMX_REQUIRE(send_counts[0] == recv_counts[0], "comm_all_to_all_v: single rank but send_splits != recv_splits");
        if (in != out && send_counts[0]) {
            VMX_REQUIRE(in && out, "comm_all_to_all_v: null buffer");
            VMX_CHECK_HIP(hipMemcpyAsync(out, in, send_counts[0] * elem_bytes, hipMemcpyDeviceToDevice, vmx_stream(stream)));
        }
        return VMX_OK;
    }
    if (c->comm) {
        RcclApi &a = rccl();
        size_t so = 0, ro = 0;
        if (int rc = check_rccl(a.GroupStart(), "ncclGroupStart")) return rc;
        int rc = 0;
        for (int peer = 0; peer < c->world && !rc; peer++) {
            const size_t sb = send_counts[peer] * elem_bytes, rb = recv_counts[peer] * elem_bytes;
            if (peer != c->rank) {
                if (sb) rc = a.Send((const char *)in + so, sb, NCCL_UINT8, peer, c->comm, vmx_stream(stream));
                if (!rc && rb) rc = a.Recv((char *)out + ro, rb, NCCL_UINT8, peer, c->comm, vmx_stream(stream));
            } else if (sb) {
                if (sb != rb) {
                    (void)a.GroupEnd();
                    vmx_set_error("comm_all_to_all_v: this rank's own send and receive counts differ");
                    return VMX_EINVAL;
                }
                if (hipMemcpyAsync((char *)out + ro, (const char *)in + so, sb, hipMemcpyDeviceToDevice, vmx_stream(stream)) != hipSuccess) rc = -1;
            }
            so += sb;
            ro += rb;
        }
        const int rc_end = a.GroupEnd();
        if (rc) return check_rccl(rc, "ncclSend/ncclRecv (all_to_all_v)");
        return check_rccl(rc_end, "ncclGroupEnd");
    }
    for (int r = 0; r < c->world; r++) {
        soff[r + 1] = soff[r] + send_counts[r] * elem_bytes;
        roff[r + 1] = roff[r] + recv_counts[r] * elem_bytes;
    }
    VMX_REQUIRE(c->p2p.ready, "comm_all_to_all_v: no transport (neither RCCL nor the peer-mapped one is set up)");
    VMX_REQUIRE(x_fits(c, soff[c->world]), "comm_all_to_all_v: %zu bytes to send exceed the peer-mapped staging buffer (%zu)", soff[c->world],
                c->p2p.x_slot ? c->p2p.x_slot - XCH_HDR : 0);
    VMX_REQUIRE((in || !soff[c->world]) && (out || !roff[c->world]), "comm_all_to_all_v: null buffer");
    XchOp op = {};
    op.mode = X_ALL_TO_ALL_V;
    op.in = (const uint8_t *)in;
    op.out = (uint8_t *)out;
    for (int r = 0; r <= c->world; r++) {
        op.send_off[r] = (int64_t)soff[r];
        op.recv_off[r] = (int64_t)roff[r];
    }
    return launch_exchange(c, op, stream, "p2p_all_to_all_v");
}

// all_to_all: equal chunks, chunk r of `in` to rank r (communicator.rs:67-76; the count exchange of the expert dispatch).
extern "C" int vmx_comm_all_to_all(vmx_comm *c, const void *in, void *out, size_t bytes_per_rank, void *stream) {
    VMX_REQUIRE(c, "comm_all_to_all: null pointer");
    size_t counts[P2P_MAX_RANKS];
    VMX_REQUIRE(c->world <= P2P_MAX_RANKS, "comm_all_to_all: at most %d ranks", P2P_MAX_RANKS);
    for (int r = 0; r < c->world; r++) counts[r] = bytes_per_rank;
    return vmx_comm_all_to_all_v(c, in, counts, out, counts, 1, stream);
}

// all_to_all_v whose counts live in device memory (no host read of the routing): rows of row_bytes; `in` holds the rows for
// rank 0, then rank 1, ... (send_rows[r] of them, int32 on the device); `out` receives the rows from rank 0, then rank 1, ...
// packed, and recv_rows[r] their counts.  The count exchange of the reference's dispatch (moe/token_dispatch.rs:306-327) is
// part of the same kernel.  Peer-mapped transport only.
extern "C" int vmx_comm_all_to_all_v_dev(vmx_comm *c, const void *in, const int32_t *send_rows, void *out, int32_t *recv_rows,
                                         size_t row_bytes, size_t out_capacity_rows, void *stream) {
    VMX_REQUIRE(c && in && send_rows && out && recv_rows && row_bytes > 0, "comm_all_to_all_v_dev: null pointer");
    if (int rc = comm_check(c, "comm_all_to_all_v_dev")) return rc;
    VMX_REQUIRE(c->p2p.ready && c->p2p.x_slot > XCH_HDR, "comm_all_to_all_v_dev: needs the peer-mapped transport with an exchange buffer");
    XchOp op = {};
    op.mode = X_ALL_TO_ALL_V_DEV;
    op.in = (const uint8_t *)in;
    op.out = (uint8_t *)out;
    op.send_rows = send_rows;
    op.recv_rows = recv_rows;
    op.row_bytes = (int64_t)row_bytes;
    op.out_capacity_rows = (int64_t)out_capacity_rows;
    return launch_exchange(c, op, stream, "p2p_all_to_all_v_dev");
}

// For the executors: VMX_ELAUNCH when the attached communicator's peer-mapped transport has failed (a plain host read).
int vmx_comm_check_health(const vmx_comm *c, const char *what) { return comm_check(c, what); }
