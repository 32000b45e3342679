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

// dlarfg scalars from alpha and ss = |x|^2 > 0 without the library sqrt / divisions (700 cycles a column on
// the critical path of every panel): v_rsq_f64 / v_rcp_f64 seeds + Newton steps, results within an ulp or two of the
// correctly rounded ones.  beta = -sign(alpha) sqrt(alpha^2 + ss), tau = (beta - alpha) / beta, scal = 1 / (alpha - beta).
__device__ __forceinline__ void stm_larfg_scalars(double alpha, double ss, double &beta, double &tau, double &scal)
{
    const double total = alpha * alpha + ss;
    double r = __builtin_amdgcn_rsq(total);
    r = r * (1.5 - 0.5 * total * r * r);
    r = r * (1.5 - 0.5 * total * r * r);
    double sq = total * r;
    sq = sq + 0.5 * r * (total - sq * sq);
    beta = -copysign(sq, alpha);
    const double den = alpha - beta;                    // = copysign(|alpha| + sq, alpha): no cancellation
    double ri = __builtin_amdgcn_rcp(den);
    ri = ri * (2.0 - den * ri);
    ri = ri * (2.0 - den * ri);
    scal = ri;
    tau = (beta - alpha) * (-copysign(r, alpha));       // 1 / beta = -sign(alpha) r
}

// The same with the magnitude guard of LAPACK's dlarfg / dnrm2 folded in as ONE power-of-two factor for the whole
// factorization: sg = 2^-e when the largest |entry| of A is 2^e beyond 2^+-300, else 1 (DevCtx::sig, computed on the device
// from A's values).  The callers accumulate their sums with ONE operand scaled,  ss1 = sum (sg x) x,  dots1 = sum (sg x) c,
// which keeps every product representable for entries up to 1e+-160 and beyond (orthogonal transformations keep the entries
// of a front below sqrt(m) max|A|).  Returned: beta, tau and scal = 1/(alpha - beta) in true units, and scals = scal / sg
// so that  scal * (sum x c) = scals * dots1.  With sg = 1 every multiplication here is exact: the same bits as the unguarded form.
__device__ __forceinline__ void stm_larfg_guarded(double alpha, double ss1, double sg, double isg, double &beta, double &tau,
                                                  double &scal, double &scals)
{
    double bs;
    stm_larfg_scalars(alpha * sg, ss1 * sg, bs, tau, scals);
    beta = bs * isg;
    scal = scals * sg;
}

// Sum of n doubles, p[0], p[stride], ..., in THAT order (fixed association: deterministic partial-sum reductions), with the
// loads of eight terms in flight together -- a plain loop waits for every load before the next add, ~0.7 us per term from L2.
// (Padding a short batch with +0.0 is exact: a sum that starts at +0.0 never is -0.0.)
template <bool AGENT>
__device__ __forceinline__ double stm_ordered_sum(const double *p, long long stride, int n)
{
    double v = 0;
    int q = 0;
    // (tall fronts -- 100+ slabs: the bulk in batches of 32 loads, the same order of additions: a batch is a memory round trip)
    for (; q + 32 <= n; q += 32) {
        double t[32];
#pragma unroll
        for (int u = 0; u < 32; u++) {
            const double *a = p + (long long)(q + u) * stride;
            t[u] = AGENT ? ld_agent(a) : *a;
        }
#pragma unroll
        for (int u = 0; u < 32; u++) v += t[u];
    }
    for (; q < n; q += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double *a = p + (long long)min(q + u, n - 1) * stride;
            const double x = AGENT ? ld_agent(a) : *a;
            t[u] = (q + u < n) ? x : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) v += t[u];
    }
    return v;
}

// One lane polls *flag (agent scope) until it is >= `target` (the flags of one factorization only grow); bounded, and it
// gives up at once when *abort is set (some wait of this factorization has already run out: the caller reports that
// through FrontNum::perr and the factorization is repeated without inter-workgroup waits).  Ends with the workgroup's
// acquire: plain loads afterwards see what the publisher stored write-through before its flag.
__device__ __forceinline__ bool stm_wait_ge(const int *flag, int target, int *abort, int *s_ok)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int it = 0; it < (1 << 22); it++) {
            if (ld_agent(flag) >= target) { ok = 1; break; }
            if ((it & 63) == 63 && ld_agent(abort)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) st_agent(abort, 1);
        *s_ok = ok;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return *s_ok != 0;
}

// A bounded wait ran out: recorded on the front (which panel chain failed) AND in the plan-wide word abort[1], so that the host
// learns it from four bytes instead of a copy of every FrontNum (stmmqr_factorize_group of the phased interface).
// (abort is null in the single-front seams of stmmqr_seams.cpp: they read the front's own perr)
#define STM_SET_PERR(c, num) do { st_agent(&(num)->perr, 1); if ((c).abort) st_agent((c).abort + 1, 1); } while (0)

