// stmmqr_devutil.h -- small device helpers shared by the kernel translation units (agent-scope accesses for the
// in-launch hand-offs between workgroups: MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup
// visibility").
#pragma once
#include <hip/hip_runtime.h>

typedef double d4 __attribute__((ext_vector_type(4)));

// barrier that orders LDS traffic only: global stores stay in flight (a full __syncthreads would wait for them)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// relaxed agent-scope accesses: sc1 loads / write-through sc1 stores (served by L2, past the CU's L1)
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(char *p, char v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
