// stmmqr_kdev.h -- device code shared by the kernel translation units of the numeric phase (panel / update / sweep / pack):
// the small LDS working set of the panel routines, the 64-row chunk staging of the trailing update, T from the Gram matrix
// (dlarft), W2 = T'W1 on the matrix cores, the one-workgroup block update (qr_larftb / dlarfb, SparseQR_factorize.c:1851-1904),
// the Gram pass of a panel, qr_cpack (:1639-1685).  One family per .hip file (round 5; until then one 5500-line file):
//   stmmqr_assemble.hip  k_amax / k_sigma / k_gather_sx / k_setup / k_assemble        (qr_stranspose2, qr_fsize, qr_assemble)
//   stmmqr_panel.hip     dev_panel / dev_tall_group / dev_wave_panel, k_front_wg, k_panel, k_panel_pc   (qr_front)
//   stmmqr_update.hip    k_update, k_upd_w / k_upd_c / k_upd_f, k_upd_b0w, k_upd_fw                    (qr_larftb)
//   stmmqr_sweep.hip     pair / quad update of the large fronts (k_upd_w2 / y2 / c2, k_upd_wq / yq / cq)
//   stmmqr_pack.hip      k_cpack, k_rh_*, k_zero_slabs, k_panel_msg                              (qr_cpack, qr_rhpack)
//   stmmqr_resident.hip  Q-apply / triangular solves on the resident factors
// (stmmqr_capanel.hip: the Gram-based panel.)  Kernels are compiled one by one, so the split changes no kernel's code.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include "stmmqr_device.h"
#include "stmmqr_kernels.h"
#include "stmmqr_wave.h"
#include "stmmqr_devutil.h"

// small LDS working set of the panel routines, declared once per kernel and shared by every code path
struct PanelShared {
    double red[16];
    int st_out[STM_NB];        // Stair / dead flag of the panel columns, flushed to global once per panel (a global
    int dead[STM_NB];          //  store inside the column loop makes every barrier wait for its completion)
    double nextss;             // |x|^2 of the next column, produced by the wave that just updated it
    int nextss_col;            // ... valid for this panel column (-1: none)
    int stair[STM_NB];
    int diag[STM_NB];
    double tau[STM_NB];
    double G[STM_NB][STM_NB + 1];
    double T[STM_NB][STM_NB + 1];
    double part[8 * 32];
    double top[2][8];
    double rsum[2][64];        // register sub-panel: per-wave sums of the 8 reductions of a column step (two buffers)
    double rsumB[5][64];       // blocked application of a half group of reflectors: up to 40 sums in one exchange
    double Ts[8][9];
    double gp[32];
};

#define NT 256
#define NW (NT / 64)
#define BN 32                 // trailing-update column block
#define RB 64                 // trailing-update row chunk
#define VS (RB + 2)           // LDS row stride of the V / C chunk images (doubles)
#define WS (BN + 1)

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// sum over the whole workgroup of NTH threads; every thread gets the result.  s_red: NTH/64 doubles of LDS.
template <int NTH>
__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    v = wave_sum(v);
    __syncthreads();                       // protect s_red from the previous use
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < NTH / 64; w++) r += s_red[w];
    return r;
}

// inclusive scan of one int per thread across the workgroup; *total = sum.  s_scan: NW ints.
__device__ __forceinline__ int block_incl_scan(int v, int *s_scan, int *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) s_scan[wid] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        int sw = s_scan[w];
        if (w < wid) base += sw;
        tot += sw;
    }
    *total = tot;
    return x + base;
}

#define SLAB STM_UPD_SLAB
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)


// One 64-row chunk of V (panel columns, unit-lower-trapezoidal mask applied) and of C (one column block) goes through
// registers into LDS.  The loads are unconditional on clamped indices (a predicated load is a branch around each
// access, and the 16 loads of a thread would be issued one round trip at a time); the callers issue the loads of the
// next chunk before the MFMA loop of the current one.
struct UpdChunk { double v[8], c[8]; };
__device__ __forceinline__ void upd_chunk_load(UpdChunk &ck, const double *Vg, const double *Cg, long long ld, int i, int mp,
                                               int nbp, int nc, int lcg)
{
    const int ic = min(i, mp - 1);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        ck.v[q] = Vg[ic + (long long)min(col, nbp - 1) * ld];
        ck.c[q] = Cg[ic + (long long)min(col, nc - 1) * ld];
    }
}
__device__ __forceinline__ void upd_chunk_to_lds(const UpdChunk &ck, int i, int mp, int nbp, int nc, const int *s_pd, int g1,
                                                 int lrow, int lcg, double *Vs, double *Cs, bool c_is_v = false, bool plain = false)
{
    if (plain) {                                               // (uniform: a chunk below every unit diagonal, inside the panel's rows,
#pragma unroll                                                 //  full blocks -- nothing to mask: the same values by a plain copy)
        for (int q = 0; q < 8; q++) {
            const int col = lcg * 8 + q;
            Vs[col * VS + lrow] = ck.v[q];
            Cs[col * VS + lrow] = c_is_v ? ck.v[q] : ck.c[q];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        const int d = s_pd[col] - g1;                          // (BIGROW beyond nbp: everything masked)
        const double v = (i < mp && col < nbp && i >= d) ? ((i == d) ? 1.0 : ck.v[q]) : 0.0;
        Vs[col * VS + lrow] = v;
        Cs[col * VS + lrow] = c_is_v ? v : ((i < mp && col < nc) ? ck.c[q] : 0.0);     // (Gram block: C = V)
    }
}

__device__ __forceinline__ void upd_chunk_v_to_lds(const UpdChunk &ck, int i, int mp, int nbp, const int *s_pd, int g1, int lrow,
                                                   int lcg, double *Vs, bool plain = false)
{
    if (plain) {
#pragma unroll
        for (int q = 0; q < 8; q++) Vs[(lcg * 8 + q) * VS + lrow] = ck.v[q];
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        const int d = s_pd[col] - g1;
        Vs[col * VS + lrow] = (i < mp && col < nbp && i >= d) ? ((i == d) ? 1.0 : ck.v[q]) : 0.0;
    }
}
// last row (relative to g1) below which a chunk of a full panel needs no masks; BIGROW: never (a dead reflector, a short panel)
__device__ __forceinline__ int upd_plain_from(const int *s_pd, int g1, int lane)
{
    const int dm = wave_max_int(lane < STM_NB ? s_pd[lane] : -1);
    return dm >= STM_BIGROW ? STM_BIGROW : dm - g1;
}

// ------------------------------------------------------------------------------------------------
// qr_larftb(QR_QTX): C <- (I - V T V')' C for one block of <= BN columns, on fp64 MFMA.
//   C = F(g1:g1+mp, c0:c0+nc), V = F(g1:g1+mp, k1:k1+nbp) with the unit diagonal of reflector j at absolute row
//   diag[j] (STM_BIGROW: no reflector) and zeros above it, T = NB x NB upper triangular (column-major, ld NB).
//   lds: >= 2*BN*VS + STM_NB*WS doubles.  diag / T may live in LDS or global memory.
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[i=(l>>4)+4r][j=l&15].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_T_from_gram(double (*G)[STM_NB + 1], double (*T)[STM_NB + 1], const double *tau, int nc, int tid);

// W2 = T' W1 (T upper triangular, NB x NB; W1 NB x BN) on the matrix cores: wave `wid` (0..3) gets the 16 x 16 tile (wid >> 1, wid & 1)
// of W2 in the MFMA result layout, W2[16 (wid >> 1) + (lane >> 4) + 4 r][16 (wid & 1) + (lane & 15)] = result[r].  T(q, l) is read at
// Tm[q * sq + l * sl] (any memory; it must hold zeros below the diagonal and beyond the panel's reflectors), W1[q][x] at W1[q * WS + x]
// (LDS).  Rows l < 16 only see q < 16 (the skipped products are exact zeros).  EVERY form of the trailing update forms W2 here, so
// that a front gets the same bits whichever form its step uses (the scalar loops this replaces were 2.7 us of every update
// workgroup's prologue: 160 LDS reads per thread).
__device__ __forceinline__ d4 dev_w2_tile(const double *Tm, int sq, int sl, const double *W1, int wid, int lane)
{
    const int mi = wid >> 1, ni = wid & 1, l15 = lane & 15, l4 = lane >> 4;
    d4 acc = {0, 0, 0, 0};
    const int nk = mi ? STM_NB / 4 : STM_NB / 8;
#pragma unroll
    for (int kk = 0; kk < STM_NB / 4; kk++) {
        if (kk < nk) {
            const double a = Tm[(4 * kk + l4) * sq + (16 * mi + l15) * sl];       // A[i = l][k = q] = T(q, l)
            const double b = W1[(4 * kk + l4) * WS + 16 * ni + l15];               // B[k = q][j = x]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
    return acc;
}

// tau != nullptr: T was left to the update by the panel kernel (PanelDesc::t_deferred) -- G = V'V is accumulated beside
// W1 (per 256-row slab, slabs added in order: bit-identical to the Gram block of k_upd_w) and T is built here by every
// workgroup for itself (dev_T_from_gram); Tout / Tkeep (may be null) receive it from the caller's first column block.
// TN = true (qr_larftb seam, method QR_QX only): C <- (I - V T V') C, i.e. W2 = T W1 instead of T' W1.
#define STM_UPD_LDS_DOUBLES (2 * BN * VS + STM_NB * WS)
static_assert(STM_UPD_LDS_DOUBLES == STM_UPD_LDS_HOST, "host sizing of the update kernels' LDS");
template <bool TN = false>
__device__ void dev_update_block(double *F, long long ld, int g1, int mp, int k1, int nbp, const int *diag,
                                 const double *T, int c0, int nc, double *lds, const double *tau = nullptr,
                                 double *Tout = nullptr, double *Tkeep = nullptr)
{
    if (nbp <= 0 || mp <= 0 || nc <= 0) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    double *Vs = lds;                       // [STM_NB][VS]
    double *Cs = Vs + STM_NB * VS;          // [BN][VS]
    double *Ws = Cs + BN * VS;              // [STM_NB][WS]
    __shared__ int s_pd[STM_NB];

    __syncthreads();                        // previous users of lds are done
    if (tid < STM_NB) s_pd[tid] = (tid < nbp) ? diag[tid] : STM_BIGROW;
    __syncthreads();

    const int lrow = tid & 63, lcg = tid >> 6;          // loader mapping: 64 rows x 4 column groups of 8
    const double *Vg = F + g1 + (long long)k1 * ld;
    double *Cg = F + g1 + (long long)c0 * ld;

    // ---- phase 1: W1 = V' C ----
    const int mi = wid >> 1, ni = wid & 1;
    // W1 is accumulated per slab of 256 rows and the slabs are added in order -- exactly the association of the
    // row-parallel form (k_upd_w partials summed by k_upd_c), so that a front gets bit-identical results whichever of the
    // two its level happens to use
    const bool build_t = (tau != nullptr);
    d4 acc = {0, 0, 0, 0}, tot = {0, 0, 0, 0}, gacc = {0, 0, 0, 0}, gtot = {0, 0, 0, 0};
    UpdChunk ck;
    upd_chunk_load(ck, Vg, Cg, ld, lrow, mp, nbp, nc, lcg);
    for (int r0 = 0; r0 < mp; r0 += RB) {
        const int i = r0 + lrow;
        upd_chunk_to_lds(ck, i, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
        __syncthreads();
        if (r0 + RB < mp) upd_chunk_load(ck, Vg, Cg, ld, i + RB, mp, nbp, nc, lcg);
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk++) {
            const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
            const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            if (build_t) {
                const double bv = Vs[(16 * ni + l15) * VS + 4 * kk + l4];
                gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, gacc, 0, 0, 0);
            }
        }
        if ((r0 + RB) % STM_UPD_SLAB == 0 || r0 + RB >= mp) {   // end of a slab (SLAB of the row-parallel form)
#pragma unroll
            for (int r = 0; r < 4; r++) { tot[r] += acc[r]; acc[r] = 0; gtot[r] += gacc[r]; gacc[r] = 0; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; r++) Ws[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = tot[r];
    double *s_Tm = Cs;                       // T(q, l) at s_Tm[q * WS + l] when it is built here (the chunk images are free)
    if (build_t) {
        double (*Gm)[STM_NB + 1] = reinterpret_cast<double (*)[STM_NB + 1]>(Vs);
        __shared__ double s_tau_u[STM_NB];
#pragma unroll
        for (int r = 0; r < 4; r++) Gm[16 * mi + l4 + 4 * r][16 * ni + l15] = gtot[r];
        if (tid < STM_NB) s_tau_u[tid] = (tid < nbp) ? tau[tid] : 0.0;
        __syncthreads();
        dev_T_from_gram(Gm, reinterpret_cast<double (*)[STM_NB + 1]>(s_Tm), s_tau_u, nbp, tid);
        if (Tout || Tkeep)
            for (int e = tid; e < STM_NB * STM_NB; e += (int)blockDim.x) {
                const int a = e % STM_NB, b = e / STM_NB;
                const double tv = (a <= b && a < nbp && b < nbp) ? s_Tm[a * WS + b] : 0.0;
                if (Tout) Tout[e] = tv;
                if (Tkeep) Tkeep[e] = tv;
            }
    }
    __syncthreads();

    // ---- phase 2: W2 = T' W1 (T upper triangular) ----
    {
        const int l = tid & 31, cg = tid >> 5;          // 8 groups x 4 columns
        double w2[4] = {0, 0, 0, 0};
        if (TN) {
            for (int q = l; q < nbp; q++) {                // row l of the upper triangular T
                const double tq = build_t ? s_Tm[l * WS + q] : T[l + q * STM_NB];
#pragma unroll
                for (int x = 0; x < 4; x++) w2[x] += tq * Ws[q * WS + cg * 4 + x];
            }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 4; x++) Ws[l * WS + cg * 4 + x] = w2[x];
        } else {
            d4 t2 = {0, 0, 0, 0};
            if (wid < 4) t2 = build_t ? dev_w2_tile(s_Tm, WS, 1, Ws, wid, lane) : dev_w2_tile(T, 1, STM_NB, Ws, wid, lane);
            __syncthreads();
            if (wid < 4) {
#pragma unroll
                for (int r = 0; r < 4; r++) Ws[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = t2[r];
            }
        }
    }
    __syncthreads();

    // ---- phase 3: C -= V W2 ----
    upd_chunk_load(ck, Vg, Cg, ld, lrow, mp, nbp, nc, lcg);
    for (int r0 = 0; r0 < mp; r0 += RB) {
        const int i = r0 + lrow;
        upd_chunk_to_lds(ck, i, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
        __syncthreads();
        if (r0 + RB < mp) upd_chunk_load(ck, Vg, Cg, ld, i + RB, mp, nbp, nc, lcg);
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
            Cs[l15 * VS + row] -= u0[r];
            Cs[(16 + l15) * VS + row] -= u1[r];
        }
        __syncthreads();
        if (i < mp) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int col = lcg * 8 + q;
                if (col < nc) Cg[i + col * ld] = Cs[col * VS + lrow];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// dlarft('F','C'): T (NB x NB upper triangular, column-major, zero padded) of nc <= NB reflectors stored in the
// columns of Vg (= &F(0, first column), absolute row indexing, leading dimension ld), rows [r0, r1).
// diag[j] = absolute row of the unit diagonal of reflector j (BIGROW: none), tau[j] its coefficient.
//   G = V'V by fp64 MFMA (each wave sweeps every NW-th group of 4 rows; the same register is the A and the B
//   operand of the diagonal tiles), cross-wave sum through LDS scratch, then
//   T(0:b-1,b) = -tau_b T(0:b-1,0:b-1) G(0:b-1,b)   (SURVEY.md A.4).   scratch: >= NW*3*256 doubles.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// dlarft (forward, columnwise) from the Gram matrix G = V'V of a panel, tau = 0 columns included as zero columns:
//   T(a,b) = -tau_b sum_{a <= l < b} T(a,l) G(l,b),  T(b,b) = tau_b.
// Blocked 16 + 16: the two diagonal blocks are independent recurrences (one row of T per lane, two waves side by
// side), the off-diagonal block is T12 = -T11 (G12 T22), two 16 x 16 products by 256 threads -- the chain of
// dependent LDS reads + FMAs is a quarter of the 32-column recurrence's.  G's lower-left block is used as scratch.
// G, T: LDS, row stride STM_NB + 1; needs >= 256 threads; T is written completely (zeros below the diagonal and for
// columns >= nc).  Ends with a barrier.
// ------------------------------------------------------------------------------------------------
// tid: index of the thread among the (at least) 256 that work on THIS G / T (threadIdx.x, or threadIdx.x & 255 when the two halves
// of a 512-thread workgroup each build their own); the barriers are the whole workgroup's either way.
// Round 5: the recurrence is blocked all the way down -- T of 2h columns from the T of its two halves,
//   T = [T11, -T11 (G12 T22); 0, T22]      (dlarft's own merge rule; G12 = V1'V2),
// for h = 1, 2, 4, 8, 16: ten short steps of independent dot products (at most 16 terms, 256 outputs) instead of two 16-step chains of
// dependent LDS round trips + the 16/16 merge (3.7 us -> 1 us: T sits on the chain of every step, between the panel and block 0).
// A reflector with tau = 0 (dead / identity column) gives a zero row and column, as the column-by-column form does.
// X = G12 T22 is kept in G's lower triangle (the block below the diagonal block pair: never read as G).
template <int h>
__device__ __forceinline__ void dev_T_merge(double (*G)[STM_NB + 1], double (*T)[STM_NB + 1], int tid)
{
    const int pr = tid / (h * h), ij = tid % (h * h), i = ij / h, j = ij % h;
    const int A0 = 2 * h * pr, B0 = A0 + h;
    const bool mine = tid < (STM_NB / 2) * h;
    // (T11, T22 hold explicit zeros below their diagonals, so the sums run over all k with no predicates: the extra products
    //  are exact zeros)
    if (mine) {                                                      // X(i, j) = sum_k G12(i, k) T22(k, j)
        double p0 = 0, p1 = 0;
#pragma unroll
        for (int k = 0; k < h; k += 2) {
            p0 += G[A0 + i][B0 + k] * T[B0 + k][B0 + j];
            if (k + 1 < h) p1 += G[A0 + i][B0 + k + 1] * T[B0 + k + 1][B0 + j];
        }
        G[B0 + i][A0 + j] = p0 + p1;
    }
    lds_barrier();                                                   // (LDS only: stores to global memory stay in flight)
    if (mine) {                                                      // T12(i, j) = -sum_k T11(i, k) X(k, j)
        double p0 = 0, p1 = 0;
#pragma unroll
        for (int k = 0; k < h; k += 2) {
            p0 += T[A0 + i][A0 + k] * G[B0 + k][A0 + j];
            if (k + 1 < h) p1 += T[A0 + i][A0 + k + 1] * G[B0 + k + 1][A0 + j];
        }
        T[A0 + i][B0 + j] = -(p0 + p1);
    }
    lds_barrier();
}
__device__ __forceinline__ void dev_T_from_gram(double (*G)[STM_NB + 1], double (*T)[STM_NB + 1], const double *tau, int nc, int tid)
{
    for (int e = tid; e < STM_NB * STM_NB; e += 256) {
        const int a = e >> 5, b = e & 31;
        T[a][b] = (a == b && a < nc) ? tau[a] : 0.0;
    }
    lds_barrier();
    dev_T_merge<1>(G, T, tid);
    dev_T_merge<2>(G, T, tid);
    dev_T_merge<4>(G, T, tid);
    dev_T_merge<8>(G, T, tid);
    dev_T_merge<16>(G, T, tid);
}

template <int NTH>
__device__ void dev_gram_T(const double *Vg, long long ld, int r0, int r1, int nc, const int *diag, const double *tau,
                           double (*s_G)[STM_NB + 1], double (*s_T)[STM_NB + 1], double *Tout, double *scratch, unsigned long long *tl = nullptr)
{
    constexpr int NWV = NTH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const bool two = nc > 16;
    const int d0 = (l15 < nc) ? diag[l15] : STM_BIGROW;
    const int d1 = (16 + l15 < nc) ? diag[16 + l15] : STM_BIGROW;
    // (columns and rows are clamped so that every load is unconditional -- a predicated load is a branch around the
    //  access and the loads of a trip would be issued one round trip at a time -- and masked afterwards)
    const int ncc = max(nc, 1);
    const double *V0 = Vg + (long long)min(l15, ncc - 1) * ld, *V1 = Vg + (long long)min(16 + l15, ncc - 1) * ld;
    d4 g00 = {0, 0, 0, 0}, g01 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
    const int nk = (r1 - r0 + 3) / 4;
    // four row groups per trip: the 8 loads of the NEXT trip are issued before the MFMAs of the current one (the loop is
    // bound by the memory latency of a trip otherwise)
    double x0[4], x1[4], y0[4], y1[4];
    auto load_trip = [&](int kk, double (&p0)[4], double (&p1)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = r0 + 4 * (kk + u * NWV) + l4;
            const int ic = max(min(i, r1 - 1), 0);
            p0[u] = V0[ic]; p1[u] = V1[ic];
        }
    };
    if (wid < nk) load_trip(wid, x0, x1);
    for (int kk = wid; kk < nk; kk += 4 * NWV) {
        const int kn = kk + 4 * NWV;
        if (kn < nk) load_trip(kn, y0, y1);
        double a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = r0 + 4 * (kk + u * NWV) + l4;
            const bool in = (kk + u * NWV < nk) && (i < r1);
            a0[u] = (in && i >= d0) ? ((i == d0) ? 1.0 : x0[u]) : 0.0;
            a1[u] = (in && two && i >= d1) ? ((i == d1) ? 1.0 : x1[u]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], a0[u], g00, 0, 0, 0);
            if (two) {
                g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], a1[u], g01, 0, 0, 0);
                g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], a1[u], g11, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { x0[u] = y0[u]; x1[u] = y1[u]; }
    }
    __syncthreads();                                    // scratch is free
    if (tl && tid == 0) tl[1] = wall_clock64();
    // cross-wave sum in groups of 4 waves (scratch: 4 * 768 doubles)
    for (int grp = 0; grp < NWV / 4; grp++) {
        if ((wid >> 2) == grp) {
            double *sc = scratch + (wid & 3) * 768;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (l4 + 4 * r) * 16 + l15;
                if (grp == 0) { sc[o] = g00[r]; sc[256 + o] = g01[r]; sc[512 + o] = g11[r]; }
                else { sc[o] += g00[r]; sc[256 + o] += g01[r]; sc[512 + o] += g11[r]; }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < 768; e += NTH) {
        const double v = scratch[e] + scratch[768 + e] + scratch[1536 + e] + scratch[2304 + e];
        const int tile = e >> 8, a = (e >> 4) & 15, b = e & 15;
        s_G[a + (tile == 2 ? 16 : 0)][b + (tile >= 1 ? 16 : 0)] = v;
    }
    __syncthreads();
    if (tl && tid == 0) tl[2] = wall_clock64();
    dev_T_from_gram(s_G, s_T, tau, nc, threadIdx.x);
    if (tl && tid == 0) tl[3] = wall_clock64();
    for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
        const int a = e % STM_NB, b = e / STM_NB;
        Tout[e] = (a < nc && b < nc && a <= b) ? s_T[a][b] : 0.0;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// qr_cpack: C = F(rank:, fp:) upper trapezoid -> packed column-major; coalesced on the
// packed side.  part/nparts as in dev_assemble.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_cpack(const DevCtx &c, const FrontSym &s, FrontNum *num, int part, int nparts)
{
    const int tid = threadIdx.x;
    const int rank = num->rank, fm = num->fm;
    const int cn = s.fn - s.fp;
    int cm = min(fm - rank, cn);
    if (cm < 0) cm = 0;
    if (part == 0 && tid == 0) num->cm = cm;
    if (cm <= 0 || cn <= 0) return;
    const long long ld = s.ld;
    const double *Fc = c.Farena + s.foff + rank + (long long)s.fp * ld;
    double *C = c.Carena + s.coff;
    const long long tri = (long long)cm * (cm + 1) / 2;
    const long long csize = tri + (long long)cm * (cn - cm);
    for (long long e = (long long)part * NT + tid; e < csize; e += (long long)nparts * NT) {
        int cj, ci;
        if (e < tri) {
            cj = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while ((long long)cj * (cj + 1) / 2 > e) cj--;
            while ((long long)(cj + 1) * (cj + 2) / 2 <= e) cj++;
            ci = (int)(e - (long long)cj * (cj + 1) / 2);
        } else {
            const long long r = e - tri;
            cj = cm + (int)(r / cm);
            ci = (int)(r % cm);
        }
        C[e] = Fc[ci + cj * ld];
    }
}
