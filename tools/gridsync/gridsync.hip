// Measurement aid: cost of a grid-wide barrier inside one resident kernel on gfx950 (8 XCDs, one L2 each).
//   build:  hipcc -O3 --offload-arch=gfx950 -o tools/gridsync/gridsync tools/gridsync/gridsync.hip
//   run:    tools/gridsync/gridsync [workgroups] [threads] [iterations]
// Variants: 0 one counter (atomic add + spin on it), 1 one flag per workgroup (store own, poll all), 2 flags with a payload
// hand-off (each workgroup writes 64 bytes before the barrier and reads another workgroup's after it: checks visibility
// across XCDs), 3 = two-level (per-XCD counter, then a counter of XCDs).
// Every spin gives up after 2 ms of wall clock and raises an error word, so a mistake cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now_ticks() {   // 100 MHz
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

struct Args {
    unsigned *counter;     // variant 0 / 3
    unsigned *flags;       // [nwg] variant 1 / 2
    unsigned *xcd_ctr;     // [8 * 16] variant 3 (one cache line each)
    unsigned *err;
    float *payload;        // [nwg][16]
    unsigned long long *ticks;   // [2]
    int iters;
};

template <int V>
__global__ __launch_bounds__(1024) void sync_kernel(Args a) {
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    unsigned long long t0 = 0;
    if (wg == 0 && tid == 0) t0 = now_ticks();
    __shared__ int sdead;
    if (tid == 0) sdead = 0;
    __syncthreads();
    bool dead = false;
    float acc = 0.f;
    for (int it = 1; it <= a.iters && !dead; it++) {
        if (V == 2 && tid < 16) a.payload[(it & 1) * 16384 + wg * 16 + tid] = (float)(it * 1000 + wg);
        __syncthreads();
        if (V == 0) {
            if (tid == 0) {
                __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)it * nwg;
                unsigned long long start = 0;
                for (unsigned spins = 0; __hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spins++) {
                    if ((spins & 1023) == 1023) {
                        if (!start) start = now_ticks();
                        else if (now_ticks() - start > 200000ull) { *a.err = 1; sdead = 1; break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else if (V == 1 || V == 2) {
            if (tid == 0) __hip_atomic_store(a.flags + wg, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 64) {   // wave 0 polls all flags, nwg <= 1024: up to 16 per lane
                unsigned long long start = 0;
                for (unsigned spins = 0;; spins++) {
                    bool ok = true;
                    for (int i = tid; i < nwg; i += 64)
                        ok &= __hip_atomic_load(a.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)it;
                    if (__all(ok)) break;
                    if ((spins & 1023) == 1023) {
                        if (!start) start = now_ticks();
                        else if (now_ticks() - start > 200000ull) { *a.err = 2; sdead = 1; break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            if (tid == 0) {
                const int x = wg & 7, per = (nwg + 7 - x) / 8;   // workgroups of this XCD
                const unsigned old = __hip_atomic_fetch_add(a.xcd_ctr + x * 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)it * per) __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)it * 8;
                unsigned long long start = 0;
                for (unsigned spins = 0; __hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spins++) {
                    if ((spins & 1023) == 1023) {
                        if (!start) start = now_ticks();
                        else if (now_ticks() - start > 200000ull) { *a.err = 3; sdead = 1; break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        if (sdead) dead = true;   // a timed-out workgroup stops; the others time out in turn
        if (V == 2 && tid < 16) {
            const int src = (wg * 37 + it) % nwg;
            const float v = a.payload[(it & 1) * 16384 + src * 16 + tid];
            if (v != (float)(it * 1000 + src)) *a.err = 100;
            acc += v;
            // the next iteration overwrites payload: everyone must have read before anyone writes again -> the barrier at the
            // top of the next iteration is NOT enough (a fast workgroup could overwrite before a slow one has read), so the
            // payload alternates between two halves in a real kernel; here the source index changes and values are checked
            // against the iteration, so a too-early overwrite shows up as error 100 (it would read it + 1's value).
        }
    }
    if (acc == 12345.678f) a.payload[0] = acc;
    if (wg == 0 && tid == 0) { a.ticks[0] = t0; a.ticks[1] = now_ticks(); }
}

int main(int argc, char **argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 256, th = argc > 2 ? atoi(argv[2]) : 1024, iters = argc > 3 ? atoi(argv[3]) : 2000;
    if (nwg < 8 || nwg > 1024 || th < 64 || th > 1024) { fprintf(stderr, "bad arguments\n"); return 1; }
    int dev_cus = 0;
    CK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sync_kernel<1>, th, 0));
    printf("CUs %d, resident workgroups per CU at %d threads: %d\n", dev_cus, th, occ);
    if ((long)occ * dev_cus < nwg) { fprintf(stderr, "grid of %d would not be resident at once: refusing\n", nwg); return 1; }
    Args a;
    CK(hipMalloc(&a.counter, 256));
    CK(hipMalloc(&a.flags, 4096));
    CK(hipMalloc(&a.xcd_ctr, 8 * 64));
    CK(hipMalloc(&a.err, 256));
    CK(hipMalloc(&a.payload, 2 * 65536));
    CK(hipMalloc(&a.ticks, 64));
    a.iters = iters;
    for (int v = 0; v < 4; v++) {
        CK(hipMemset(a.counter, 0, 256)); CK(hipMemset(a.flags, 0, 4096)); CK(hipMemset(a.xcd_ctr, 0, 512)); CK(hipMemset(a.err, 0, 256));
        CK(hipMemset(a.payload, 0, 2 * 65536));
        CK(hipDeviceSynchronize());
        switch (v) {
            case 0: hipLaunchKernelGGL(sync_kernel<0>, dim3(nwg), dim3(th), 0, 0, a); break;
            case 1: hipLaunchKernelGGL(sync_kernel<1>, dim3(nwg), dim3(th), 0, 0, a); break;
            case 2: hipLaunchKernelGGL(sync_kernel<2>, dim3(nwg), dim3(th), 0, 0, a); break;
            default: hipLaunchKernelGGL(sync_kernel<3>, dim3(nwg), dim3(th), 0, 0, a); break;
        }
        CK(hipDeviceSynchronize());
        unsigned err = 0; unsigned long long tk[2];
        CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tk, a.ticks, 16, hipMemcpyDeviceToHost));
        printf("variant %d: %d workgroups x %d threads, %d barriers: %.3f us per barrier, err %u\n", v, nwg, th, iters,
               (double)(tk[1] - tk[0]) / 100.0 / iters, err);
        if (err) { fprintf(stderr, "stopping after an error\n"); return 2; }
    }
    return 0;
}
