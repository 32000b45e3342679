// Wave64 reduction primitives for gfx950 (DPP + v_permlane{16,32}_swap; no LDS crossbar traffic).
// Used by the Householder panel kernels: every column step of qr_front's dlarfg/dlarf pair (reference
// STMMQR/qr/qr_kernel.c:1359-1381, 1434-1609) needs one norm and up to seven v'c dot products summed over the
// whole workgroup, and that reduction is the critical path of the step.
#pragma once
#include <hip/hip_runtime.h>

// one DPP step of an fp64 butterfly: v + (v moved by the DPP control), both 32-bit halves moved separately
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi2, lo2);
}
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) { return v + dpp_mov<CTRL>(v); }

// v(l) + v(l ^ 16): v_permlane16_swap exchanges the odd rows of one operand with the even rows of the other
__device__ __forceinline__ double xor16_add(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// v(l) + v(l ^ 32)
__device__ __forceinline__ double xor32_add(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// sum over the 64 lanes of a wave, result in every lane.  quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror give every lane the total of its row of 16; two permlane swaps combine the four rows.
__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    v = xor16_add(v);
    return xor32_add(v);
}

// sum over the lanes that share (lane & 7): row_ror:8 inside the rows of 16, then the two row exchanges
__device__ __forceinline__ double wave_sum_stride8(double v)
{
    v = dpp_add<0x128>(v);
    v = xor16_add(v);
    return xor32_add(v);
}

// Eight sums at once.  Each halving step exchanges one half of the values with a partner lane and keeps the
// other half (7 exchange-adds instead of 8 x 3); afterwards lane l holds the sum over its group of 8 lanes of
// value number  idx(l) = ((l>>1)&1) + 2*(l&1) + 4*((l>>2)&1),  and wave_sum_stride8 finishes the job:
// every lane l ends with the wave total of value idx(l).
__device__ __forceinline__ int red8_idx(int lane) { return ((lane >> 1) & 1) + 2 * (lane & 1) + 4 * ((lane >> 2) & 1); }
__device__ __forceinline__ constexpr int red8_lane(int x) { return ((x >> 1) & 1) + 2 * (x & 1) + 4 * ((x >> 2) & 1); }

// maximum of an int over the 64 lanes of the wave, in every lane
__device__ __forceinline__ int wave_max_int(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ double wave_reduce8_partial(const double (&v)[8])
{
    const int lane = threadIdx.x & 63;
    const bool sA = (lane >> 2) & 1, sB = lane & 1, sC = (lane >> 1) & 1;
    double n4[4], n2[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {                       // partner 7 - (l & 7): the other quad
        const double keep = sA ? v[i + 4] : v[i];
        const double send = sA ? v[i] : v[i + 4];
        n4[i] = keep + dpp_mov<0x141>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {                       // partner l ^ 1
        const double keep = sB ? n4[i + 2] : n4[i];
        const double send = sB ? n4[i] : n4[i + 2];
        n2[i] = keep + dpp_mov<0xB1>(send);
    }
    const double keep = sC ? n2[1] : n2[0];             // partner l ^ 2
    const double send = sC ? n2[0] : n2[1];
    return keep + dpp_mov<0x4E>(send);
}
__device__ __forceinline__ double wave_reduce8(const double (&v)[8])
{
    return wave_sum_stride8(wave_reduce8_partial(v));
}

// value of lane SRC (compile-time) in every lane, as a wave-uniform value
template <int SRC>
__device__ __forceinline__ double lane_bcast(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), SRC),
                            __builtin_amdgcn_readlane(__double2loint(v), SRC));
}
