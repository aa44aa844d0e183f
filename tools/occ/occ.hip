// Occupancy probe: resident workgroups per CU the runtime reports for a 256-thread kernel at several dynamic LDS sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(float *p) {
    extern __shared__ float s[];
    s[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    p[threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d\n", pr.sharedMemPerBlock,
           pr.sharedMemPerMultiprocessor, pr.maxSharedMemoryPerMultiProcessor, pr.regsPerBlock);
    for (size_t lds : {4096, 8192, 10240, 10752, 13312, 16384, 21552, 32768, 65536, 81920}) {
        if (lds > 65536) hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds);
        printf("lds %6zu -> %d workgroups per CU (%s)\n", lds, n, hipGetErrorString(e));
    }
    return 0;
}
