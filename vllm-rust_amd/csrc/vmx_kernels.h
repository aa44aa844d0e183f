// Internal C++ launch interfaces shared between translation units of libvmx (not part of the C ABI).
#pragma once
#include "vmx_internal.h"

// int4_kernels.hip
int vmx_int4_splits(int N, int K);
int vmx_int4_gemm_partial(const void *packed, const uint16_t *x, float *partial, int M, int N, int K, int group, hipStream_t st);
int vmx_awq_convert_into(const uint32_t *awq_qweight, const uint32_t *awq_qzeros, uint32_t *qweight_out, uint32_t *qzeros_out, int K,
                         int N, int group, int Ntot, int col0, hipStream_t st);
