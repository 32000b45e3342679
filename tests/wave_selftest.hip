// Self-test of the wave64 reduction primitives (csrc/stmmqr_wave.h): built and run by tests/test_gpu_wave.py.
// Integer-valued inputs: every sum is exact, so the comparison is bitwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "stmmqr_wave.h"   // -I <package>/csrc
__global__ void k(const double *in, double *out_sum, double *out8, double *out8b)
{
    const int l = threadIdx.x;
    double v[8];
    for (int x = 0; x < 8; x++) v[x] = in[l * 8 + x];
    out_sum[l] = wave_sum(v[0]);
    const double r = wave_reduce8(v);
    out8[l] = r;
    out8b[l * 8 + 0] = lane_bcast<red8_lane(0)>(r); out8b[l * 8 + 1] = lane_bcast<red8_lane(1)>(r);
    out8b[l * 8 + 2] = lane_bcast<red8_lane(2)>(r); out8b[l * 8 + 3] = lane_bcast<red8_lane(3)>(r);
    out8b[l * 8 + 4] = lane_bcast<red8_lane(4)>(r); out8b[l * 8 + 5] = lane_bcast<red8_lane(5)>(r);
    out8b[l * 8 + 6] = lane_bcast<red8_lane(6)>(r); out8b[l * 8 + 7] = lane_bcast<red8_lane(7)>(r);
}
int main()
{
    double h[512], *d, *o1, *o2, *o3, r1[64], r2[64], r3[512];
    for (int i = 0; i < 512; i++) h[i] = (double)((i * 7919) % 1013) - 500.0;    // exact integer sums
    hipMalloc(&d, sizeof h); hipMalloc(&o1, 512); hipMalloc(&o2, 512); hipMalloc(&o3, 4096);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o1, o2, o3);
    hipMemcpy(r1, o1, 512, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 512, hipMemcpyDeviceToHost);
    hipMemcpy(r3, o3, 4096, hipMemcpyDeviceToHost);
    double s[8] = {0};
    for (int l = 0; l < 64; l++) for (int x = 0; x < 8; x++) s[x] += h[l * 8 + x];
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        if (r1[l] != s[0]) bad++;
        const int idx = ((l >> 1) & 1) + 2 * (l & 1) + 4 * ((l >> 2) & 1);
        if (r2[l] != s[idx]) bad++;
        for (int x = 0; x < 8; x++) if (r3[l * 8 + x] != s[x]) bad++;
    }
    printf("redtest bad=%d (s0=%g got %g)\n", bad, s[0], r1[5]);
    return bad != 0;
}
