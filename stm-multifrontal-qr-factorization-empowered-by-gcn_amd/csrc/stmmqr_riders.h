// stmmqr_riders.h -- the trailing update's tiles in the form they take as RIDERS of the panel launch (k_panel_pc, stmmqr_panel.hip).
#pragma once
#include "stmmqr_kdev.h"


// ------------------------------------------------------------------------------------------------
// Passenger launches (options.lookahead = 2, the default).  The chain of a large front is panel(t) -> T(t) + update of column
// block 0 -> panel(t+1); the rest of update t (column blocks 1..) does not feed panel t+1.  Instead of a second stream (two
// cross-stream hand-offs of ~9 us per step) the two launches of that rest RIDE on the chain's own launches, as extra workgroups
// behind the chain's in dispatch order:
//   B(t)   = k_upd_fw : T(t) + block 0 (k_upd_f's workgroups)          + k_upd_w of blocks 1.. of step t
//   A(t+1) = k_panel_pc: panel(t+1) (k_panel's workgroups, dispatched first) + k_upd_c of blocks 1.. of step t
// Every workgroup does exactly what it does in the serial order (same bits); a launch lasts as long as its longest role.
// Dependencies: k_upd_w(t) needs V(t) [A(t)] and C after k_upd_c(t-1) [A(t)]; k_upd_c(t) needs its partial sums [B(t)] and T(t)
// [B(t)]; block 0 of step t+1 is block 1 of step t: complete after A(t+1), before B(t+1).
// ------------------------------------------------------------------------------------------------
// k_upd_c's tile by a 512-thread workgroup: the two 256-thread halves take alternate 64-row chunks of the slab; every row of C sees
// exactly the operations of k_upd_c.  lds: STM_NB * WS (W2) + per half the V image and the product image (BN * VS each).
#define STM_PC_LDS_DOUBLES (STM_NB * WS + 4 * BN * VS)
__device__ __forceinline__ void dev_upd_c_h2(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                             const double *Wp, const long long *__restrict__ wpoff, int fi, int cb, int sl0,
                                             int rspw, double *dyn_lds, int *s_pd)
{
    const int f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    const PanelDesc *pd = &c.fnum[f].pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    const int nslf = stm_upd_nsl(s);
    const int c0 = pd->pc0 + (cb0 + cb * (1 + c.cbskip)) * BN;
    // (a rider takes rspw consecutive slabs -- the launch's choice: the rows of C are independent, the arithmetic does not change --
    //  so that one prologue, during which nothing else runs on this CU, serves more rows)
    const int sl = sl0 * rspw, spw = rspw;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl * SLAB >= mp) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Ws = dyn_lds, *Vs = Ws + STM_NB * WS + half * (2 * BN * VS), *Cs = Vs + BN * VS;
    double *s_W1 = dyn_lds + STM_NB * WS, *s_T = s_W1 + BN * VS;       // (half 0's images: first written after the prologue)
    if (threadIdx.x < STM_NB) s_pd[threadIdx.x] = ((int)threadIdx.x < nbp) ? pd->pdiag[threadIdx.x] : STM_BIGROW;
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int lrow = tid & 63, lcg = tid >> 6;
    const int rbeg = sl * SLAB, rend = min(mp, (sl + spw) * SLAB);
    int r0 = rbeg + half * RB;
    // (a rider has the CU to itself -- the launch carries the panel's registers and LDS -- so nothing else hides its round trips:
    //  the first two chunks of the half are requested before the prologue, from then on two trips ahead)
    UpdChunk ck0, ck1;
    upd_chunk_load(ck0, Vg, Cg, ld, r0 + lrow, mp, nbp, nc, lcg);
    upd_chunk_load(ck1, Vg, Cg, ld, r0 + 2 * RB + lrow, mp, nbp, nc, lcg);
    {
        const double *W0 = Wp + wpoff[fi] + ((long long)cb * nslf) * (STM_NB * BN);
        const double *T = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
        for (int e = threadIdx.x; e < STM_NB * BN; e += 2 * NT) {
            const double v = W0[e];
            s_W1[(e / BN) * WS + (e % BN)] = v;
            s_T[(e / STM_NB) * WS + (e % STM_NB)] = T[e];          // s_T[col][row] = T(row, col)
        }
        __syncthreads();
        d4 w2 = {0, 0, 0, 0};
        if (half == 0) w2 = dev_w2_tile(s_T, 1, WS, s_W1, wid, lane);
        __syncthreads();                                           // (Ws does not alias the prologue images; the barrier orders
                                                                   //  the reads of s_W1 / s_T before half 0's first V image)
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) Ws[(16 * (wid >> 1) + l4 + 4 * r) * WS + 16 * (wid & 1) + l15] = w2[r];
        }
    }
    const int pfrom = upd_plain_from(s_pd, g1, lane);
    const int nch = (rend - rbeg + RB - 1) / RB, trips = (nch + 1) / 2;
    auto trip = [&](UpdChunk &ck) {
        const bool valid = r0 < rend;
        const int i = r0 + lrow;
        double cc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cc[q] = ck.c[q];
        upd_chunk_v_to_lds(ck, i, mp, nbp, s_pd, g1, lrow, lcg, Vs, valid && r0 > pfrom && r0 + RB <= mp);
        __syncthreads();                                           // (the first trip: Ws too)
        if (r0 + 4 * RB < rend) upd_chunk_load(ck, Vg, Cg, ld, i + 4 * RB, mp, nbp, nc, lcg);      // (this buffer's next trip)
        d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const double a = Vs[(4 * kk + l4) * VS + 16 * wid + l15];
            const double b0 = Ws[(4 * kk + l4) * WS + l15];
            const double b1 = Ws[(4 * kk + l4) * WS + 16 + l15];
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, u1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wid + l4 + 4 * r;
            Cs[l15 * VS + row] = u0[r];
            Cs[(16 + l15) * VS + row] = u1[r];
        }
        __syncthreads();
        if (valid && i < mp) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int col = lcg * 8 + q;
                if (col < nc) Cg[i + col * ld] = cc[q] - Cs[col * VS + lrow];
            }
        }
        r0 += 2 * RB;
    };
    for (int j = 0; j < trips; j += 2) {
        trip(ck0);
        if (j + 1 < trips) trip(ck1);
    }
}
