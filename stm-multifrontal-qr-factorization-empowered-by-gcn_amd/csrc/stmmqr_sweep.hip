// stmmqr_sweep.hip -- the trailing update of the LARGE fronts (>= 16 384 rows): the block reflectors of two / four consecutive panels
// in one sweep over the trailing columns (k_upd_w2 / y2 / c2, k_upd_wq / yq / cq).  Shared device code: stmmqr_kdev.h.
#include "stmmqr_kdev.h"


// ------------------------------------------------------------------------------------------------
// Pair update (large fronts, stm_use_pair): the block reflectors of TWO consecutive panels a = p-1 (even) and b = p applied
// in ONE sweep over the columns beyond panel p+1 -- the trailing update of a 27 000-row front is bound by its three
// passes over C per 32 columns (W = V'C reads it, C -= V W reads and writes it); two panels per sweep make that 1.5.
//   k_upd_w2 : partial  W1 = V1(slab)' C(slab),  W2 = V2(slab)' C(slab)      (the SAME C: before either application)
//              last column block: C := V1  ->  G21 = V2'V1
//              the last slab workgroup of a column block to arrive adds the partials in slab order (as k_upd_w)
//   k_upd_c2 : Y1 = T1' W1,  Y2 = T2' (W2 - G21 Y1)  -- which is V2'(C - V1 Y1), the second application's own W --,
//              C(slab) -= V1 Y1 + V2 Y2
// Exactly H_b' H_a' C in exact arithmetic; the rounding differs from two separate updates, so WHICH fronts take it is a
// property of the front alone (symbolic), like the choice of the panel kernel.  The columns of the next TWO panels are
// updated panel by panel (k_upd_w / k_upd_c on column blocks 0, 1 after an even panel, block 0 after an odd one): the
// panel factorizations need them.  Rows: [g1a, max(pta, ptb)); V1 / V2 are masked by their own diagonals and ends.
// ------------------------------------------------------------------------------------------------
struct UpdChunk2 { double v1[8], v2[8], c[8]; };
__device__ __forceinline__ void upd2_chunk_load(UpdChunk2 &ck, const double *V1g, const double *V2g, const double *Cg, long long ld,
                                                int i, int mp, int nb1, int nb2, int nc, int lcg)
{
    const int ic = min(i, mp - 1);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        ck.v1[q] = V1g[ic + (long long)min(col, nb1 - 1) * ld];
        ck.v2[q] = V2g[ic + (long long)min(col, max(nb2, 1) - 1) * ld];
        ck.c[q] = Cg[ic + (long long)min(col, nc - 1) * ld];
    }
}
template <int STRIDE = VS>
__device__ __forceinline__ void upd2_chunk_to_lds(const UpdChunk2 &ck, int i, int mp, int mp1, int mp2, int nb1, int nb2, int nc,
                                                  const int *s_pd1, const int *s_pd2, int g1, int lrow, int lcg, double *Vs1,
                                                  double *Vs2, double *Cs, bool c_is_v1)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        const int d1 = s_pd1[col] - g1, d2 = s_pd2[col] - g1;   // (BIGROW beyond nb: everything masked)
        const double v1 = (i < mp1 && col < nb1 && i >= d1) ? ((i == d1) ? 1.0 : ck.v1[q]) : 0.0;
        const double v2 = (i < mp2 && col < nb2 && i >= d2) ? ((i == d2) ? 1.0 : ck.v2[q]) : 0.0;
        Vs1[col * STRIDE + lrow] = v1;
        Vs2[col * STRIDE + lrow] = v2;
        Cs[col * STRIDE + lrow] = c_is_v1 ? v1 : ((i < mp && col < nc) ? ck.c[q] : 0.0);
    }
}
// LDS column stride of the chunk images of k_upd_w2: with 68 doubles (2 x 68 = 8 mod 64 dwords) the 16-byte operand reads below
// are conflict-free in every 16-lane group of ds_read_b128 (MI355X_MICROARCH.md, LDS: bank = (a / 4) mod 64, groups
// {0-3,12-15,20-27}, ...); the 8-byte reads of the earlier form were merged by the compiler into ds_read2_b64, which runs at half
// the LDS rate with a 32-bank modulus: two-way conflicts on the 66-double stride (profiles/r03_a_*: 39 % of the LDS cycles)
#define VS2 (RB + 4)
typedef double d2v __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// slabs of 256 rows one workgroup of the pair kernels takes (the pair's rows in slabs -> 1, 2 or 4)
__device__ __forceinline__ int stm_pair_spw(int nsl, int tune)
{
    if (tune & 15) return 1 << ((tune & 15) - 1);
    return (nsl >= 32) ? 4 : (nsl >= 16) ? 2 : 1;
}
// the two panel descriptions of a pair and what both kernels derive from them (uniform per workgroup)
struct PairGeom { int g1, mp, mp1, mp2, nb1, nb2, k1a, k1b, pc0; };
__device__ __forceinline__ bool pair_geom(const FrontNum *num, int p, PairGeom &G)
{
    const PanelDesc *pa = &num->pd[STM_PDI(p - 1)], *pb = &num->pd[STM_PDI(p)];
    G.nb1 = pa->pnb;
    if (G.nb1 <= 0) return false;                               // (panel a did nothing: then b did nothing either)
    G.nb2 = pb->pnb > 0 ? pb->pnb : 0;
    G.g1 = pa->pg1;
    G.mp1 = pa->pt - pa->pg1;
    G.mp2 = G.nb2 > 0 ? pb->pt - pa->pg1 : 0;
    G.mp = max(G.mp1, G.mp2);
    G.k1a = pa->pk1; G.k1b = G.nb2 > 0 ? pb->pk1 : pa->pk1;
    G.pc0 = pa->pc0 + 2 * STM_NB;                               // first column beyond panel p+1
    return G.mp > 0;
}

__global__ __launch_bounds__(NT) void k_upd_w2(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff, int *wcnt)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_pd1[STM_NB], s_pd2[STM_NB];
    __shared__ int s_ticket;
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !(p & 1)) return;
    FrontNum *num = &c.fnum[f];
    PairGeom G;
    if (!pair_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);   // pair column blocks: beyond block 0 of panel p
    if (ncbp <= 0) return;
    const int cb = blockIdx.x, sl = blockIdx.y;
    const bool gram = (cb == (int)gridDim.x - 1);                    // last block of the launch: G21 = V2'V1
    if (!gram && cb >= ncbp) return;
    if (gram && G.nb2 <= 0) return;
    const int c0 = gram ? G.k1a : G.pc0 + cb * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    // A workgroup takes `spw` consecutive slabs of its column block (a property of the PANEL PAIR, like k_upd_c2's rule): its
    // life is then 16 chunks instead of 4 behind the same descriptor chain, ticket and partial-W store (at one slab per workgroup
    // those were most of a workgroup's 25 us for 3.4 us of MFMA work), and there are `spw` times fewer partials to add.
    const int nsl = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_pair_spw(nsl, c.tune);
    if (sl % spw) return;
    const int nc = gram ? G.nb1 : min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // (the chunk images are read 16 bytes at a time: they start on a 16-byte boundary of the LDS, one spare double is allocated)
    double *Vs1 = dyn_lds + ((((unsigned)(uintptr_t)dyn_lds) >> 3) & 1), *Vs2 = Vs1 + STM_NB * VS2, *Cs = Vs2 + STM_NB * VS2;
    const PanelDesc *pa = &num->pd[STM_PDI(p - 1)], *pb = &num->pd[STM_PDI(p)];
    if (tid < STM_NB) {
        s_pd1[tid] = (tid < G.nb1) ? pa->pdiag[tid] : STM_BIGROW;
        s_pd2[tid] = (tid < G.nb2) ? pb->pdiag[tid] : STM_BIGROW;
    }
    __syncthreads();
    const double *Fb = c.Farena + s.foff + G.g1;
    const double *V1g = Fb + (long long)G.k1a * ld, *V2g = Fb + (long long)G.k1b * ld, *Cg = Fb + (long long)c0 * ld;
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    const int rend = min(G.mp, (sl + spw) * SLAB);
    UpdChunk2 ck;
    upd2_chunk_load(ck, V1g, V2g, Cg, ld, sl * SLAB + (tid & 63), G.mp, G.nb1, G.nb2, nc, tid >> 6);
    // The K index of an MFMA is a summation index: lane group l4 takes the rows 8 kk + 2 l4 and 8 kk + 2 l4 + 1 of a chunk for
    // two consecutive MFMAs, so that both operands of both come from ONE 16-byte LDS read each (three ds_read_b128 per four
    // MFMAs instead of six 8-byte reads).
    const d2v *A1p = reinterpret_cast<const d2v *>(Vs1 + (16 * mi + l15) * VS2 + 2 * l4);
    const d2v *A2p = reinterpret_cast<const d2v *>(Vs2 + (16 * mi + l15) * VS2 + 2 * l4);
    const d2v *Bp = reinterpret_cast<const d2v *>(Cs + (16 * ni + l15) * VS2 + 2 * l4);
    for (int r0 = sl * SLAB; r0 < rend; r0 += RB) {
        upd2_chunk_to_lds<VS2>(ck, r0 + (tid & 63), G.mp, G.mp1, G.mp2, G.nb1, G.nb2, nc, s_pd1, s_pd2, G.g1, tid & 63, tid >> 6, Vs1,
                               Vs2, Cs, gram);
        __syncthreads();
        if (r0 + RB < rend) upd2_chunk_load(ck, V1g, V2g, Cg, ld, r0 + RB + (tid & 63), G.mp, G.nb1, G.nb2, nc, tid >> 6);
#pragma unroll
        for (int kk = 0; kk < RB / 8; kk++) {
            const d2v a1 = A1p[4 * kk], a2 = A2p[4 * kk], b = Bp[4 * kk];
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b.x, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b.y, acc2, 0, 0, 0);
        }
        __syncthreads();
    }
    // slot of (column block, slab group): two blocks, W1 then W2; the Gram block is column block ncbp
    const int ngrp = (nsl + spw - 1) / spw;
    double *W0 = Wp + wpoff[fi] + ((long long)(gram ? ncbp : cb) * stm_pair_slots(nslf, c.tune)) * (2 * STM_NB * BN);
    double *W = W0 + (long long)(sl / spw) * (2 * STM_NB * BN);
    if (ngrp == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc1[r];
            W[STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc2[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc1[r]);
        st_agent(&W[STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc2[r]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + (gram ? ncbp : cb);
    if (tid == 0) {
        s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket == ngrp - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_ticket != ngrp - 1) return;
    double v[2 * STM_NB * BN / NT];
#pragma unroll
    for (int q = 0; q < 2 * STM_NB * BN / NT; q++) v[q] = stm_ordered_sum<true>(W0 + tid + q * NT, 2 * STM_NB * BN, ngrp);   // fixed order
#pragma unroll
    for (int q = 0; q < 2 * STM_NB * BN / NT; q++) W0[tid + q * NT] = v[q];
}

// Y of a pair, once per column block (between k_upd_w2 and k_upd_c2): Y1 = T1' W1, Y2 = T2' (W2 - G21 Y1) from the summed W of the
// block, the Gram block G21 of the same launch of k_upd_w2 and the two T factors; the NEGATED 64 x 32 Y goes to the front's Y area
// (DevCtx::Ypend, by absolute column block), in the layout k_upd_c2's lanes read their MFMA operand from.  (It used to be the prologue of EVERY workgroup of
// k_upd_c2: five 8 KB images through LDS and three 32-step triangular loops, 6-8 us in front of 13 us of MFMA work.)
__global__ __launch_bounds__(NT) void k_upd_y2(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff)
{
    extern __shared__ double dyn_lds[];
    const int fi = blockIdx.y, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !(p & 1)) return;
    FrontNum *num = &c.fnum[f];
    PairGeom G;
    const bool live = pair_geom(num, p, G);
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);
    const int cb = blockIdx.x;
    const int tid = threadIdx.x;
    if (!live) return;
    if (cb >= ncbp || G.pc0 + cb * BN >= s.fn) return;
    double *s_W1 = dyn_lds, *s_W2 = s_W1 + STM_NB * WS, *s_T1 = s_W2 + STM_NB * WS, *s_T2 = s_T1 + STM_NB * WS,
           *s_G = s_T2 + STM_NB * WS, *s_Y1 = s_G + STM_NB * WS;
    const int nslp = stm_pair_slots(nslf, c.tune);
    double *W0 = Wp + wpoff[fi] + ((long long)cb * nslp) * (2 * STM_NB * BN);
    const double *Gr = Wp + wpoff[fi] + ((long long)ncbp * nslp) * (2 * STM_NB * BN) + STM_NB * BN;     // W2 part of the Gram block
    const double *T1 = c.Tws + (long long)STM_TSLOT(c.tslot[f], p - 1) * STM_NB * STM_NB;
    const double *T2 = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    const bool has2 = G.nb2 > 0;
    for (int e = tid; e < STM_NB * BN; e += NT) {
        s_W1[(e / BN) * WS + (e % BN)] = W0[e];
        s_W2[(e / BN) * WS + (e % BN)] = has2 ? W0[STM_NB * BN + e] : 0.0;
        s_G[(e / BN) * WS + (e % BN)] = has2 ? Gr[e] : 0.0;                    // G21(a, b) = v2_a' v1_b
        s_T1[(e / STM_NB) * WS + (e % STM_NB)] = T1[e];                          // s_T[col][row] = T(row, col)
        s_T2[(e / STM_NB) * WS + (e % STM_NB)] = has2 ? T2[e] : 0.0;
    }
    __syncthreads();
    const int l = tid & 31, cg = tid >> 5;
    double y[4] = {0, 0, 0, 0};
    for (int q = 0; q <= l; q++) {                             // Y1 = T1' W1
        const double tq = s_T1[l * WS + q];
#pragma unroll
        for (int x = 0; x < 4; x++) y[x] += tq * s_W1[q * WS + cg * 4 + x];
    }
#pragma unroll
    for (int x = 0; x < 4; x++) s_Y1[l * WS + cg * 4 + x] = y[x];
    __syncthreads();
    double z[4];                                               // Z = W2 - G21 Y1  ( = V2'(C - V1 Y1) )
#pragma unroll
    for (int x = 0; x < 4; x++) z[x] = s_W2[l * WS + cg * 4 + x];
    for (int b = 0; b < STM_NB; b++) {
        const double gq = s_G[l * WS + b];
#pragma unroll
        for (int x = 0; x < 4; x++) z[x] -= gq * s_Y1[b * WS + cg * 4 + x];
    }
    __syncthreads();                                           // (everyone has read W2 before Z replaces it)
#pragma unroll
    for (int x = 0; x < 4; x++) s_W2[l * WS + cg * 4 + x] = z[x];
    __syncthreads();
    double y2[4] = {0, 0, 0, 0};
    for (int q = 0; q <= l; q++) {                             // Y2 = T2' Z
        const double tq = s_T2[l * WS + q];
#pragma unroll
        for (int x = 0; x < 4; x++) y2[x] += tq * s_W2[q * WS + cg * 4 + x];
    }
    double *Yo = c.Ypend + c.ypoff[f] + (long long)((G.pc0 + cb * BN) >> 5) * (2 * STM_NB * BN);          // by absolute column block
#pragma unroll
    for (int x = 0; x < 4; x++) {
        Yo[l * BN + cg * 4 + x] = -y[x];
        Yo[(STM_NB + l) * BN + cg * 4 + x] = -y2[x];
    }
}

__global__ __launch_bounds__(NT) void k_upd_c2(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist,
                                               const double *Wp, const long long *__restrict__ wpoff)
{
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !(p & 1)) return;
    const FrontNum *num = &c.fnum[f];
    PairGeom G;
    if (!pair_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1;
    const int cb = blockIdx.x, sl = blockIdx.y;
    if (cb >= ncbp) return;
    const int c0 = G.pc0 + cb * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    const int nsl_all = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_pair_spw(nsl_all, c.tune);
    if (sl % spw) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const PanelDesc *pa = &num->pd[STM_PDI(p - 1)], *pb = &num->pd[STM_PDI(p)];
    const double *Fb = c.Farena + s.foff + G.g1;
    const double *__restrict__ V1g = Fb + (long long)G.k1a * ld, *__restrict__ V2g = Fb + (long long)G.k1b * ld;
    double *__restrict__ Cg = c.Farena + s.foff + G.g1 + (long long)c0 * ld;
    const double *__restrict__ Yn = c.Ypend + c.ypoff[f] + (long long)(c0 >> 5) * (2 * STM_NB * BN);      // -Y, [64][32] (k_upd_y2)
    // Round 3: the application runs WITHOUT LDS and without barriers.  The product is formed transposed,
    //     D(col, row) = C(row, col) - sum_k Y(k, col) V(row, k),
    // so that (i) C is the accumulator operand the MFMA starts from -- lane (l15, l4) holds D(col l4 + 4 r, row l15): for a
    // fixed r the 64 lanes touch 4 columns x 16 consecutive rows, i.e. four full 128-byte segments of the column-major front,
    // loaded from and stored to global memory directly in that layout -- and (ii) V is the B operand B(k, row): 16 consecutive
    // rows of 4 reflector columns per load, again whole segments.  Y (negated, from k_upd_y2) is the A operand and stays in
    // registers for the whole workgroup.  A wave takes every fourth 16-row tile of the workgroup's rows.
    //
    // Two forms of a tile.  INTERIOR tiles -- every row below all 64 unit diagonals, inside both panels' row ranges, a full
    // 32-column block: all but the first 64 rows of a pair, its last rows and the front's last column block -- run a pipeline
    // with NO predication: 24 unconditional loads of the next tile, 32 MFMAs, 8 unconditional stores.  That matters more than
    // the saved mask arithmetic: with a branch around any load or store of the loop the compiler no longer knows how many
    // memory operations are outstanding and waits for vmcnt(0) at the top of EVERY tile -- i.e. for the stores it has just
    // issued (measured: 325 ms per factorization of c5mid for this kernel, 210 ms without its stores, 165 ms with neither
    // loads nor stores).  The other tiles take the general form (masks, clamped loads, predicated stores), one at a time.
    double yn0[2 * STM_NB / 4], yn1[2 * STM_NB / 4];
#pragma unroll
    for (int kk = 0; kk < 2 * STM_NB / 4; kk++) {
        yn0[kk] = Yn[(4 * kk + l4) * BN + l15];
        yn1[kk] = Yn[(4 * kk + l4) * BN + 16 + l15];
    }
    // last row (relative to g1) that holds a unit diagonal; BIGROW: a dead reflector.  One load per lane and a wave reduction (round 4:
    // it was a loop of 64 dependent loads in front of every workgroup's tiles)
    int dmax;
    {
        const int q = lane & 31;
        const int d = (lane < 32) ? ((q < G.nb1) ? pa->pdiag[q] - G.g1 : STM_BIGROW) : ((q < G.nb2) ? pb->pdiag[q] - G.g1 : STM_BIGROW);
        dmax = wave_max_int(d);
    }
    const int rbeg = sl * SLAB, rend = min(G.mp, (sl + spw) * SLAB);
    const int ntile = (rend - rbeg + 15) >> 4;
    const int rfull = min(min(G.mp1, G.mp2), rend);     // rows below this are inside both panels
    // tiles [t_lo, t_hi) are interior (nothing if the column block is ragged or some reflector is dead)
    int t_lo = (dmax >= STM_BIGROW || nc < BN) ? ntile : max(0, (dmax + 1 - rbeg + 15) >> 4);
    int t_hi = (rfull - rbeg) >> 4;
    if (t_lo > ntile) t_lo = ntile;
    if (t_hi < t_lo) t_hi = t_lo;
    struct Tile { double c0[4], c1[4], v1[STM_NB / 4], v2[STM_NB / 4]; };
    // general form of one tile
    auto general_tile = [&](int tix) {
        const int i = rbeg + 16 * tix + l15;                                           // my row (relative to g1)
        const int row = min(i, G.mp - 1);                                              // (clamped: masked afterwards)
        Tile t;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            t.c0[r] = Cg[row + (long long)min(l4 + 4 * r, nc - 1) * ld];
            t.c1[r] = Cg[row + (long long)min(16 + l4 + 4 * r, nc - 1) * ld];
        }
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            t.v1[kk] = V1g[row + (long long)min(4 * kk + l4, G.nb1 - 1) * ld];
            t.v2[kk] = V2g[row + (long long)min(4 * kk + l4, max(G.nb2, 1) - 1) * ld];
        }
        d4 a0 = {t.c0[0], t.c0[1], t.c0[2], t.c0[3]}, a1 = {t.c1[0], t.c1[1], t.c1[2], t.c1[3]};
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const int col = 4 * kk + l4;
            const int d = (col < G.nb1) ? pa->pdiag[col] - G.g1 : STM_BIGROW;
            const double bv = (i < G.mp1 && i >= d) ? ((i == d) ? 1.0 : t.v1[kk]) : 0.0;
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[kk], bv, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[kk], bv, a1, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const int col = 4 * kk + l4;
            const int d = (col < G.nb2) ? pb->pdiag[col] - G.g1 : STM_BIGROW;
            const double bv = (i < G.mp2 && i >= d) ? ((i == d) ? 1.0 : t.v2[kk]) : 0.0;
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[STM_NB / 4 + kk], bv, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[STM_NB / 4 + kk], bv, a1, 0, 0, 0);
        }
        if (i < G.mp) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (l4 + 4 * r < nc) Cg[i + (long long)(l4 + 4 * r) * ld] = a0[r];
                if (16 + l4 + 4 * r < nc) Cg[i + (long long)(16 + l4 + 4 * r) * ld] = a1[r];
            }
        }
    };
    int tix = wid;
    for (; tix < t_lo; tix += NW) general_tile(tix);
    if (tix < t_hi) {
        // interior tiles, four per trip: lane addresses advance by 64 rows per tile.  C comes from HBM and is requested THREE
        // tiles ahead (four register images used in turn -- the trip is unrolled so that none is ever copied: copying the
        // destination of a load in flight would wait for it), V (L2) one tile ahead and BEFORE the C loads of the same step:
        // the memory counter retires in order, so the wait for V(t) covers nothing younger -- C(t+2), the stores of tile
        // t-1, V(t+1) and C(t+3) stay in flight (a counted vmcnt(40)).  The scheduling barriers keep that issue order.
        // Requests beyond this wave's last tile of the trips are clamped to it (loaded again, never used); the 1-3 interior
        // tiles left after the last whole trip take the general form.
        const int nint = (t_hi - 1 - tix) / NW + 1, ntrip = nint >> 2;
        if (ntrip > 0) {
            const double *cp = Cg + (rbeg + 16 * tix + l15) + (long long)l4 * ld;
            const double *v1p = V1g + (rbeg + 16 * tix + l15) + (long long)l4 * ld;
            const double *v2p = V2g + (rbeg + 16 * tix + l15) + (long long)l4 * ld;
            const long long ld4 = 4 * ld;
            struct TC { double c0[4], c1[4]; };
            struct TV { double v1[STM_NB / 4], v2[STM_NB / 4]; };
            auto load_c = [&](TC &t, int off) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    t.c0[r] = cp[off + r * ld4];
                    t.c1[r] = cp[off + (4 + r) * ld4];
                }
            };
            auto load_v = [&](TV &t, int off) {
#pragma unroll
                for (int kk = 0; kk < STM_NB / 4; kk++) {
                    t.v1[kk] = v1p[off + kk * ld4];
                    t.v2[kk] = v2p[off + kk * ld4];
                }
            };
            TC cb4[4];
            TV vb2[2];
            const int step = 16 * NW, offlast = step * (4 * ntrip - 1);
            load_v(vb2[0], 0);
            load_c(cb4[0], 0);
            load_c(cb4[1], min(step, offlast));
            load_c(cb4[2], min(2 * step, offlast));
            for (int trip = 0; trip < ntrip; trip++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int off = step * (4 * trip + q);
                    load_v(vb2[(q + 1) & 1], min(off + step, offlast));
                    load_c(cb4[(q + 3) & 3], min(off + 3 * step, offlast));
                    __builtin_amdgcn_sched_barrier(0);
                    const TC &tc = cb4[q];
                    const TV &tv = vb2[q & 1];
                    d4 a0 = {tc.c0[0], tc.c0[1], tc.c0[2], tc.c0[3]}, a1 = {tc.c1[0], tc.c1[1], tc.c1[2], tc.c1[3]};
#pragma unroll
                    for (int kk = 0; kk < STM_NB / 4; kk++) {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[kk], tv.v1[kk], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[kk], tv.v1[kk], a1, 0, 0, 0);
                    }
#pragma unroll
                    for (int kk = 0; kk < STM_NB / 4; kk++) {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[STM_NB / 4 + kk], tv.v2[kk], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[STM_NB / 4 + kk], tv.v2[kk], a1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double *sp = const_cast<double *>(cp) + off;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        sp[r * ld4] = a0[r];
                        sp[(4 + r) * ld4] = a1[r];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tix += 4 * ntrip * NW;
        }
    }
    for (; tix < ntile; tix += NW) general_tile(tix);
}

// ------------------------------------------------------------------------------------------------
// Quad update (round 4, options.pair_update = 4): the block reflectors of FOUR consecutive panels q0 = p-3 .. q3 = p
// (p = 3 mod 4) in ONE sweep over the columns beyond panel p+1.  The sweeps of a 27 000-row front are bound by what they
// move per MFMA: a 16-row tile of k_upd_c2 brings 4 KB of C in, 4 KB out and 8 KB of V for 32 MFMAs; with four panels it is
// the same C for 64 MFMAs (0.75 instead of 1.5 passes over the trailing columns per panel).
//   k_upd_wq : partial W_i = V_i(slab)' C(slab), i = 1..4 (the same C: before any application); three more column blocks with
//              C := V_1, V_2, V_3 give the Gram blocks G_ij = V_i'V_j (i > j)
//   k_upd_yq : Y_i = T_i' (W_i - sum_{j<i} G_ij Y_j)   -- V_i' of C after the applications before it --, negated, to Ypend
//   k_upd_cq : C(slab) -= sum_i V_i Y_i
// Exactly H_3' H_2' H_1' H_0' C in exact arithmetic.  The columns of the next FOUR panels are updated panel by panel (column
// blocks 0 .. 3-r after panel r of a quad).  Which fronts take it is symbolic (stmmqr_plan::pair_front), as for the pair.
// ------------------------------------------------------------------------------------------------
#define QP 4                       // panels per sweep
#define QRB 32                     // rows per chunk of k_upd_wq (five chunk images: 46 KB of LDS, three workgroups per CU)
#define VSQ (QRB + 4)              // column stride of the chunk images: 16-byte operand reads conflict free (as VS2)
#define YSQ 48                     // row stride of -Y in k_upd_cq's LDS: the two 32-lane halves of an 8-byte read on disjoint banks
// slabs of 256 rows one workgroup of the quad kernels takes: as the pair's rule, and 8 from 64 slabs on (configs[4] stand-in, 203 slabs:
// 3434 -> 3370 ms; c5mid, 106 slabs: +-0; 8 everywhere: c5mid 580 -> 596.  tune >> 8: another threshold, measurement sweeps)
__device__ __forceinline__ int stm_quad_spw(int nsl, int tune)
{
    if (tune & 15) return 1 << ((tune & 15) - 1);
    return (nsl >= ((tune >> 8) ? (tune >> 8) : 64)) ? 8 : stm_pair_spw(nsl, 0);
}
struct QuadGeom { int g1, mp, pc0; int mpi[QP], nb[QP], k1[QP]; };
// (entry i of a per-panel array for a run-time i: selects over constant indices, so that the arrays stay in registers)
__device__ __forceinline__ int qsel(const int (&a)[QP], int i)
{
    int r = a[0];
#pragma unroll
    for (int q = 1; q < QP; q++) r = (i == q) ? a[q] : r;
    return r;
}
__device__ __forceinline__ bool quad_geom(const FrontNum *num, int p, QuadGeom &G)
{
    const PanelDesc *p0 = &num->pd[STM_PDI(p - (QP - 1))];
    if (p0->pnb <= 0) return false;                             // (the first panel did nothing: then none of them did)
    G.g1 = p0->pg1;
    G.pc0 = p0->pc0 + QP * STM_NB;                              // first column beyond panel p+1
    G.mp = 0;
#pragma unroll
    for (int i = 0; i < QP; i++) {
        const PanelDesc *pi = &num->pd[STM_PDI(p - (QP - 1) + i)];
        const int nb = pi->pnb > 0 ? pi->pnb : 0;
        G.nb[i] = nb;
        G.mpi[i] = nb > 0 ? pi->pt - G.g1 : 0;
        G.k1[i] = nb > 0 ? pi->pk1 : p0->pk1;
        G.mp = max(G.mp, G.mpi[i]);
    }
    return G.mp > 0;
}

__global__ __launch_bounds__(NT, 3) void k_upd_wq(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff, int *wcnt)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_pd[QP][STM_NB];
    __shared__ int s_ticket;
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || (p & (QP - 1)) != QP - 1) return;
    FrontNum *num = &c.fnum[f];
    QuadGeom G;
    if (!quad_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);   // sweep column blocks: beyond block 0 of panel p
    if (ncbp <= 0) return;
    const int cbx = blockIdx.x, sl = blockIdx.y;
    const int gj = cbx - ncbp;                                       // >= 0: Gram block, C := V_gj
    if (gj >= QP - 1) return;
    const bool gram = gj >= 0;
    if (gram && qsel(G.nb, gj + 1) <= 0) return;                     // (no panel behind it: nobody reads its products)
    const int c0 = gram ? qsel(G.k1, gj) : G.pc0 + cbx * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    const int nsl = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_quad_spw(nsl, c.tune);
    if (sl % spw) return;
    const int nc = gram ? qsel(G.nb, gj) : min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // five chunk images (the V of the four panels and C) of 32 rows: 46 KB, three workgroups per CU.  (Measured against it: chunks of 64
    // rows in two phases -- panels 0, 1 then 2, 3 against one C image, three images as k_upd_w2, 168 registers -- c5mid 607 -> 621 ms.)
    double *Vs = dyn_lds + ((((unsigned)(uintptr_t)dyn_lds) >> 3) & 1), *Cs = Vs + QP * STM_NB * VSQ;
    __shared__ int s_dm[2];
    if (tid < 2 * 64) {                                              // (the four panels' diagonals; their maximum: BIGROW if any is missing)
        const int i = tid >> 5, q = tid & 31;
        const int d = (q < qsel(G.nb, i)) ? num->pd[STM_PDI(p - (QP - 1) + i)].pdiag[q] : STM_BIGROW;
        s_pd[i][q] = d;
        const int dm = wave_max_int(d);
        if (lane == 0) s_dm[wid] = dm;
    }
    __syncthreads();
    // chunks below every unit diagonal, inside all four panels' rows, with full column blocks need no masks (all but the first 128 rows and
    // the last rows of a quad): their staging is a plain copy -- the masks are ~8 VALU instructions per element, 160 per thread and chunk
    const int dmax = max(s_dm[0], s_dm[1]);
    const int rfull = min(min(G.mpi[0], G.mpi[1]), min(G.mpi[2], G.mpi[3]));
    const bool plain_ok = dmax < STM_BIGROW && nc == BN;
    const double *Fb = c.Farena + s.foff + G.g1;
    const double *Cg = Fb + (long long)c0 * ld;
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc[QP];
#pragma unroll
    for (int i = 0; i < QP; i++) acc[i] = (d4){0, 0, 0, 0};
    const int rend = min(G.mp, (sl + spw) * SLAB);
    const int lrow = tid & (QRB - 1), lcg = tid / QRB;              // staging: thread = (row of the chunk, group of 4 columns)
    double cv[QP][4], cc[4];
    auto chunk_load = [&](int i) {
        const int ic = min(i, G.mp - 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = lcg * 4 + q;
#pragma unroll
            for (int pi = 0; pi < QP; pi++) cv[pi][q] = Fb[ic + (long long)(G.k1[pi] + min(col, max(G.nb[pi], 1) - 1)) * ld];
            cc[q] = Cg[ic + (long long)min(col, nc - 1) * ld];
        }
    };
    auto chunk_to_lds = [&](int i) {
        const int r0c = i - lrow;                                    // first row of the chunk (uniform)
        if (plain_ok && r0c > dmax - G.g1 && r0c + QRB <= rfull) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int col = lcg * 4 + q;
#pragma unroll
                for (int pi = 0; pi < QP; pi++) Vs[(pi * STM_NB + col) * VSQ + lrow] = cv[pi][q];
                Cs[col * VSQ + lrow] = cc[q];                        // (a Gram block's C columns ARE V_gj's: loaded from the same place)
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = lcg * 4 + q;
            double vg = 0.0;
#pragma unroll
            for (int pi = 0; pi < QP; pi++) {
                const int d = s_pd[pi][col] - G.g1;                  // (BIGROW beyond nb: everything masked)
                const double v = (i < G.mpi[pi] && col < G.nb[pi] && i >= d) ? ((i == d) ? 1.0 : cv[pi][q]) : 0.0;
                Vs[(pi * STM_NB + col) * VSQ + lrow] = v;
                vg = (pi == gj) ? v : vg;
            }
            Cs[col * VSQ + lrow] = gram ? vg : ((i < G.mp && col < nc) ? cc[q] : 0.0);
        }
    };
    chunk_load(sl * SLAB + lrow);
    const d2v *Ap = reinterpret_cast<const d2v *>(Vs + (16 * mi + l15) * VSQ + 2 * l4);
    const d2v *Bp = reinterpret_cast<const d2v *>(Cs + (16 * ni + l15) * VSQ + 2 * l4);
    for (int r0 = sl * SLAB; r0 < rend; r0 += QRB) {
        chunk_to_lds(r0 + lrow);
        __syncthreads();
        if (r0 + QRB < rend) chunk_load(r0 + QRB + lrow);
#pragma unroll
        for (int kk = 0; kk < QRB / 8; kk++) {
            const d2v b = Bp[4 * kk];
#pragma unroll
            for (int pi = 0; pi < QP; pi++) {
                const d2v a = Ap[pi * (STM_NB * VSQ / 2) + 4 * kk];
                acc[pi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc[pi], 0, 0, 0);
                acc[pi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc[pi], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // slot of (column block, slab group): QP blocks W_1 .. W_4; the Gram blocks are column blocks ncbp .. ncbp + 2
    const int ngrp = (nsl + spw - 1) / spw;
    double *W0 = Wp + wpoff[fi] + ((long long)cbx * stm_quad_slots(nslf, c.tune)) * (QP * STM_NB * BN);
    double *W = W0 + (long long)(sl / spw) * (QP * STM_NB * BN);
    if (ngrp == 1) {
#pragma unroll
        for (int pi = 0; pi < QP; pi++)
#pragma unroll
            for (int r = 0; r < 4; r++) W[pi * STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc[pi][r];
        return;
    }
#pragma unroll
    for (int pi = 0; pi < QP; pi++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_agent(&W[pi * STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[pi][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + cbx;
    if (tid == 0) {
        s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket == ngrp - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_ticket != ngrp - 1) return;
    double v[QP * STM_NB * BN / NT];
#pragma unroll
    for (int q = 0; q < QP * STM_NB * BN / NT; q++) v[q] = stm_ordered_sum<true>(W0 + tid + q * NT, QP * STM_NB * BN, ngrp);   // fixed order
#pragma unroll
    for (int q = 0; q < QP * STM_NB * BN / NT; q++) W0[tid + q * NT] = v[q];
}

// -Y of a quad, once per column block (between k_upd_wq and k_upd_cq)
__global__ __launch_bounds__(NT) void k_upd_yq(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff)
{
    extern __shared__ double dyn_lds[];
    const int fi = blockIdx.y, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || (p & (QP - 1)) != QP - 1) return;
    FrontNum *num = &c.fnum[f];
    QuadGeom G;
    if (!quad_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);
    const int cb = blockIdx.x, tid = threadIdx.x;
    if (cb >= ncbp || G.pc0 + cb * BN >= s.fn) return;
    double *s_Y = dyn_lds, *s_Z = s_Y + QP * STM_NB * WS, *s_T = s_Z + STM_NB * WS, *s_G = s_T + STM_NB * WS;
    const long long nslp = stm_quad_slots(nslf, c.tune);
    const double *W0 = Wp + wpoff[fi] + ((long long)cb * nslp) * (QP * STM_NB * BN);
    double *Yo = c.Ypend + c.ypoff[f] + (long long)((G.pc0 + cb * BN) >> 5) * (QP * STM_NB * BN);         // by absolute column block
    const int l = tid & 31, cg = tid >> 5;
#pragma unroll 1
    for (int i = 0; i < QP; i++) {
        double y[4] = {0, 0, 0, 0};
        if (qsel(G.nb, i) > 0) {                                // (uniform)
            double z[4];
#pragma unroll
            for (int x = 0; x < 4; x++) z[x] = W0[i * STM_NB * BN + l * BN + cg * 4 + x];
#pragma unroll 1
            for (int j = 0; j < i; j++) {                       // Z = W_i - sum_j G_ij Y_j  ( = V_i' of C after the applications 0 .. i-1 )
                const double *Gr = Wp + wpoff[fi] + ((long long)(ncbp + j) * nslp) * (QP * STM_NB * BN) + i * STM_NB * BN;   // G_ij(a, b) = v_ia' v_jb
                __syncthreads();
                for (int e = tid; e < STM_NB * BN; e += NT) s_G[(e / BN) * WS + (e % BN)] = Gr[e];
                __syncthreads();
                const double *Yj = s_Y + j * STM_NB * WS;
#pragma unroll
                for (int b = 0; b < STM_NB; b++) {
                    const double gq = s_G[l * WS + b];
#pragma unroll
                    for (int x = 0; x < 4; x++) z[x] -= gq * Yj[b * WS + cg * 4 + x];
                }
            }
            const double *T = c.Tws + STM_TSLOT(c.tslot[f], p - (QP - 1) + i) * STM_NB * STM_NB;
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 4; x++) s_Z[l * WS + cg * 4 + x] = z[x];
            for (int e = tid; e < STM_NB * STM_NB; e += NT) s_T[(e / STM_NB) * WS + (e % STM_NB)] = T[e];     // s_T[col][row] = T(row, col)
            __syncthreads();
#pragma unroll
            for (int q = 0; q < STM_NB; q++) {                  // Y_i = T_i' Z  (the trip count as a predicate: the LDS reads up front)
                const double tq = s_T[l * WS + q];
                if (q <= l) {
#pragma unroll
                    for (int x = 0; x < 4; x++) y[x] += tq * s_Z[q * WS + cg * 4 + x];
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            s_Y[i * STM_NB * WS + l * WS + cg * 4 + x] = y[x];
            Yo[(i * STM_NB + l) * BN + cg * 4 + x] = -y[x];
        }
    }
}

__global__ __launch_bounds__(NT, 2) void k_upd_cq(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist)
{
    __shared__ double s_Y[QP * STM_NB * YSQ];
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || (p & (QP - 1)) != QP - 1) return;
    const FrontNum *num = &c.fnum[f];
    QuadGeom G;
    if (!quad_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1;
    const int cb = blockIdx.x, sl = blockIdx.y;
    if (cb >= ncbp) return;
    const int c0 = G.pc0 + cb * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    const int nsl_all = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_quad_spw(nsl_all, c.tune);
    if (sl % spw) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const double *__restrict__ Fb = c.Farena + s.foff + G.g1;
    double *__restrict__ Cg = c.Farena + s.foff + G.g1 + (long long)c0 * ld;
    const double *__restrict__ Yn = c.Ypend + c.ypoff[f] + (long long)(c0 >> 5) * (QP * STM_NB * BN);    // -Y, [128][32] (k_upd_yq)
    // As k_upd_c2: no chunk images and no barriers in the loop -- D(col, row) = C(row, col) - sum_k Y(k, col) V(row, k), C the
    // accumulator operand, V the B operand, both straight from global memory in whole 128-byte segments.  -Y is the A operand: 128 x 32
    // here, kept in LDS (in registers it would leave one wave per SIMD; measured, timing only: 1.52 x the time of k_upd_c2 for twice its
    // MFMAs in registers, 1.33 x from LDS at two waves).
    for (int e = tid; e < QP * STM_NB * BN; e += NT) s_Y[(e >> 5) * YSQ + (e & 31)] = Yn[e];
    int dmax = -1;                                      // last row (relative to g1) that holds a unit diagonal; BIGROW: a dead reflector
    int rfull = G.mp;
#pragma unroll
    for (int i = 0; i < QP / 2; i++) {                  // (one load per lane and a wave reduction for two panels)
        const int pi = 2 * i + (lane >> 5), q = lane & 31;
        const int d = (q < qsel(G.nb, pi)) ? num->pd[STM_PDI(p - (QP - 1) + pi)].pdiag[q] - G.g1 : STM_BIGROW;
        dmax = max(dmax, wave_max_int(d));
    }
#pragma unroll
    for (int i = 0; i < QP; i++) rfull = min(rfull, G.mpi[i]);
    __syncthreads();
    const double *y0p = s_Y + l4 * YSQ + l15, *y1p = y0p + 16;
    const int rbeg = sl * SLAB, rend = min(G.mp, (sl + spw) * SLAB);
    const int ntile = (rend - rbeg + 15) >> 4;
    rfull = min(rfull, rend);                           // rows below this are inside all four panels
    // tiles [t_lo, t_hi) are interior (nothing if the column block is ragged, a panel is short of 32 columns or some reflector is dead)
    int t_lo = (dmax >= STM_BIGROW || nc < BN) ? ntile : max(0, (dmax + 1 - rbeg + 15) >> 4);
    int t_hi = (rfull - rbeg) >> 4;
    if (t_lo > ntile) t_lo = ntile;
    if (t_hi < t_lo) t_hi = t_lo;
    t_lo = __builtin_amdgcn_readfirstlane(t_lo);        // (uniform in fact; dmax came through a wave reduction)
    t_hi = __builtin_amdgcn_readfirstlane(t_hi);
    // general form of one tile (masks, clamped loads, predicated stores)
    auto general_tile = [&](int tix) {
        const int i = rbeg + 16 * tix + l15;                                           // my row (relative to g1)
        const int row = min(i, G.mp - 1);                                              // (clamped: masked afterwards)
        d4 a0, a1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            a0[r] = Cg[row + (long long)min(l4 + 4 * r, nc - 1) * ld];
            a1[r] = Cg[row + (long long)min(16 + l4 + 4 * r, nc - 1) * ld];
        }
#pragma unroll 1
        for (int pi = 0; pi < QP; pi++) {
            const PanelDesc *pp = &num->pd[STM_PDI(p - (QP - 1) + pi)];
            const double *Vg = Fb + (long long)qsel(G.k1, pi) * ld;
            const int nbi = qsel(G.nb, pi), mpi = qsel(G.mpi, pi);
            double v[STM_NB / 4];
#pragma unroll
            for (int kk = 0; kk < STM_NB / 4; kk++) v[kk] = Vg[row + (long long)min(4 * kk + l4, max(nbi, 1) - 1) * ld];
#pragma unroll
            for (int kk = 0; kk < STM_NB / 4; kk++) {
                const int col = 4 * kk + l4;
                const int d = (col < nbi) ? pp->pdiag[col] - G.g1 : STM_BIGROW;
                const double bv = (i < mpi && i >= d) ? ((i == d) ? 1.0 : v[kk]) : 0.0;
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0p[(pi * STM_NB + 4 * kk) * YSQ], bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1p[(pi * STM_NB + 4 * kk) * YSQ], bv, a1, 0, 0, 0);
            }
        }
        if (i < G.mp) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (l4 + 4 * r < nc) Cg[i + (long long)(l4 + 4 * r) * ld] = a0[r];
                if (16 + l4 + 4 * r < nc) Cg[i + (long long)(16 + l4 + 4 * r) * ld] = a1[r];
            }
        }
    };
    for (int tix = wid; tix < t_lo; tix += NW) general_tile(tix);
    // Interior rows in SUPER tiles of 32 rows: lane l15 takes the rows 2 l15 and 2 l15 + 1, so that every load and store of the loop is a
    // 16-byte access (half the vector-memory instructions for the same bytes: the texture-address path was 71 % busy in k_upd_c2) -- the
    // even rows are one MFMA tile, the odd rows another, both fed from the same registers.  V comes in two halves (panels 0, 1 and
    // panels 2, 3 of the quad: 64 registers each), each requested again as soon as its MFMAs are issued; C one super tile ahead; two super
    // tiles per trip, counted waits.  Requests beyond the wave's last super tile are clamped to it (loaded again, never used).
    // (the buffer requests address 128 columns from one base with 32-bit offsets: fronts of more than 2^21 rows take the general tiles)
    const bool buf_ok = ld * (long long)(QP * STM_NB * sizeof(double)) < (1LL << 31) - (1 << 20);
    const int ns = buf_ok ? (t_hi - t_lo) >> 1 : 0;     // interior super tiles of the workgroup
    int sdone = 0;                                      // ... of this wave that the pipeline took
    if (ns > wid) {
        const int nmine = __builtin_amdgcn_readfirstlane((ns - 1 - wid) / NW + 1), ntrip = nmine >> 1;
        if (ntrip > 0) {
            // Buffer addressing: one resource descriptor per operand (base: this wave's first interior row of C's / V's first column),
            // ONE 32-bit lane offset for every request, the column and tile offsets scalar -- with 64-bit lane addresses the 40 address
            // pairs of a super tile do not fit beside the V halves and the C images (spills inside the loop; scratch reloads share the
            // memory counter).  Everything that feeds a scalar offset is made uniform for the compiler (readfirstlane), see DESIGN.md.
            const int widu = __builtin_amdgcn_readfirstlane(wid);
            const long long r0u = rbeg + 16 * t_lo + 32 * widu;
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(Cg + r0u), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc((void *)const_cast<double *>(Fb + (long long)G.k1[0] * ld + r0u), 0, 0x7fffffff, 0x00020000);
            const int voff = (int)((2 * l15 + (long long)l4 * ld) * 8);
            const int ld4b = (int)(4 * ld * 8);                  // bytes between columns c and c + 4 (ld <= 2^26)
            struct TC { d2v c0[4], c1[4]; };
            struct TVh { d2v v[QP * STM_NB / 8]; };
            auto load_c = [&](TC &t, int off) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    t.c0[r] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rc, voff, off * 8 + r * ld4b, 0));
                    t.c1[r] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rc, voff, off * 8 + (4 + r) * ld4b, 0));
                }
            };
            auto load_vh = [&](TVh &t, int off, int h) {
#pragma unroll
                for (int kk = 0; kk < QP * STM_NB / 8; kk++)
                    t.v[kk] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, off * 8 + (h * (QP * STM_NB / 8) + kk) * ld4b, 0));
            };
            TC cb2[2];
            TVh va, vb;
            const int step = 32 * NW, offlast = step * (2 * ntrip - 1);
            load_c(cb2[0], 0);                                   // (in the order of a step -- the scheduler would move C behind V --:
            __builtin_amdgcn_sched_barrier(0);                   //  the counted waits of the loop's first pass are then those of
            load_vh(va, 0, 0);                                   //  every other; the loop header takes the weaker of both)
            __builtin_amdgcn_sched_barrier(0);
            load_vh(vb, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            for (int trip = 0; trip < ntrip; trip++) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int off = step * (2 * trip + q), offn = min(off + step, offlast);
                    load_c(cb2[q ^ 1], offn);
                    __builtin_amdgcn_sched_barrier(0);
                    const TC &tc = cb2[q];
                    d4 a0e = {tc.c0[0].x, tc.c0[1].x, tc.c0[2].x, tc.c0[3].x}, a0o = {tc.c0[0].y, tc.c0[1].y, tc.c0[2].y, tc.c0[3].y};
                    d4 a1e = {tc.c1[0].x, tc.c1[1].x, tc.c1[2].x, tc.c1[3].x}, a1o = {tc.c1[0].y, tc.c1[1].y, tc.c1[2].y, tc.c1[3].y};
#pragma unroll
                    for (int kk = 0; kk < QP * STM_NB / 8; kk++) {
                        const double y0 = y0p[4 * YSQ * kk], y1 = y1p[4 * YSQ * kk];
                        a0e = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, va.v[kk].x, a0e, 0, 0, 0);
                        a1e = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, va.v[kk].x, a1e, 0, 0, 0);
                        a0o = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, va.v[kk].y, a0o, 0, 0, 0);
                        a1o = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, va.v[kk].y, a1o, 0, 0, 0);
                        if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);          // (keeps the -Y operands of at most four steps in registers)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_vh(va, offn, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < QP * STM_NB / 8; kk++) {
                        const double y0 = y0p[4 * YSQ * (QP * STM_NB / 8 + kk)], y1 = y1p[4 * YSQ * (QP * STM_NB / 8 + kk)];
                        a0e = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, vb.v[kk].x, a0e, 0, 0, 0);
                        a1e = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, vb.v[kk].x, a1e, 0, 0, 0);
                        a0o = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, vb.v[kk].y, a0o, 0, 0, 0);
                        a1o = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, vb.v[kk].y, a1o, 0, 0, 0);
                        if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_vh(vb, offn, 1);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double e0 = a0e[r], o0 = a0o[r], e1 = a1e[r], o1 = a1o[r];       // (scalars first: DESIGN.md, compiler findings)
                        const d2v t0 = {e0, o0}, t1 = {e1, o1};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t0), rc, voff, off * 8 + r * ld4b, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t1), rc, voff, off * 8 + (4 + r) * ld4b, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sdone = 2 * ntrip;
        }
        // this wave's super tiles that the pipeline did not take (at most one): as two tiles of the general form
        for (int q = sdone; q < nmine; q++) {
            general_tile(t_lo + 2 * (wid + q * NW));
            general_tile(t_lo + 2 * (wid + q * NW) + 1);
        }
    }
    for (int tix = t_lo + 2 * ns + wid; tix < ntile; tix += NW) general_tile(tix);
}
// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
int stm_launch_update_pair(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncbp, int maxsl, double *Wp,
                           const long long *wpoff, int *wcnt, hipStream_t st)
{
    if (nfr <= 0 || ncbp <= 0 || maxsl <= 0) return 0;
    hipLaunchKernelGGL(k_upd_w2, dim3(ncbp + 1, maxsl, nfr), dim3(NT), (size_t)(3 * BN * VS2 + 2) * sizeof(double), st, c, flist, plist, Wp,
                       wpoff, wcnt);
    hipLaunchKernelGGL(k_upd_y2, dim3(ncbp, nfr), dim3(NT), (size_t)(6 * STM_NB * WS) * sizeof(double), st, c, flist, plist, Wp, wpoff);
    hipLaunchKernelGGL(k_upd_c2, dim3(ncbp, maxsl, nfr), dim3(NT), 0, st, c, flist, plist, (const double *)Wp, wpoff);
    return (int)hipGetLastError();
}
int stm_launch_update_quad(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncbp, int maxsl, double *Wp,
                           const long long *wpoff, int *wcnt, hipStream_t st)
{
    if (nfr <= 0 || ncbp <= 0 || maxsl <= 0) return 0;
    hipLaunchKernelGGL(k_upd_wq, dim3(ncbp + QP - 1, maxsl, nfr), dim3(NT), (size_t)((QP + 1) * BN * VSQ + 2) * sizeof(double), st, c, flist,
                       plist, Wp, wpoff, wcnt);
    hipLaunchKernelGGL(k_upd_yq, dim3(ncbp, nfr), dim3(NT), (size_t)((QP + 3) * STM_NB * WS) * sizeof(double), st, c, flist, plist, Wp, wpoff);
    hipLaunchKernelGGL(k_upd_cq, dim3(ncbp, maxsl, nfr), dim3(NT), 0, st, c, flist, plist);
    return (int)hipGetLastError();
}
int stm_configure_sweep(void)
{
    CK(hipFuncSetAttribute((const void *)k_upd_yq, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    return 0;
}
