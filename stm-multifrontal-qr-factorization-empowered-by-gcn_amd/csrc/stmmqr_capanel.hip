// stmmqr_capanel.hip -- Gram-based ("communication-avoiding") Householder panel for the large fronts (gfx950).
//
// Replaces the column-by-column reductions of qr_front's panel loop (reference STMMQR/src/qr/SparseQR_factorize.c
// :1434-1609: per column one dlarfg norm + one dlarf sweep) by ONE Gram matrix per panel.
//
// A panel = columns [k1, k1+nbp) x rows [g1, tmax).  Its rows are cut into
//     the top block  At : the nt = min(nbp, tmax-g1) rows that can become pivot rows, kept explicitly (LDS, 32 x 32)
//     the bottom rows B  : everything below, nB rows, cut into slabs of CA_R rows, one workgroup per slab, each slab
//                          resident in LDS for the whole panel.
// Everything qr_front needs from B is an inner product of two of its (updated) columns:
//     |x|^2  = sum_{i>g, top} At(i,j)^2 + b_j'b_j          (dlarfg)
//     v'a_x  = At(g,x) + scal (sum_{i>g, top} At(i,j) At(i,x) + b_j'b_x)      (dlarf, scal = 1/(alpha-beta))
// and the update of B is a column operation  b_x -= c_x b_j, b_j *= scal  -- so B is never touched inside the column
// loop: the loop ("chain") works on G = B'B (32 x 32, downdated by the congruence of each column operation), on At, and
// accumulates the column operations in M (32 x 32 upper triangular); B <- B M is applied once at the end with MFMA.
// One workgroup reduction over the rows per PANEL (the Gram matrix, by MFMA) instead of one per column.
//
// Accuracy.  G is exact when it is formed and loses absolute accuracy eps*|a_x|^2 per step; a column whose remaining norm
// has dropped below 1/CA_K of its norm at the time G was formed (cancellation: the column is nearly dependent on its
// predecessors -- the rounding-noise pivots of over-estimated contribution blocks are the typical case) triggers a
// REFRESH before it is used: B <- B M, M <- I, G <- B'B from the real rows.  The same rule LAPACK's dgeqp3 applies to
// its downdated column norms.  So every norm is accurate to CA_K*eps relative, every reflector orthogonal to that
// accuracy, and the dead-column test |beta| <= tol (:1495) sees an accurate beta.  Validated on the
// fixtures' real fronts against the unblocked column loop by the numpy model of this algorithm (tests/ca_model.py, tests/test_ca_model.py).
//
// Several slabs (nB > CA_R): every slab workgroup forms its partial Gram matrix, stores it write-through and takes a
// ticket; the LAST one to arrive sums the partials in slab order (deterministic), runs the chain alone and publishes M;
// the others apply it to their slab.  Nobody ever waits for a workgroup that has not started yet (the owner is the last
// arriver; the others wait for the owner, which is running) so the protocol cannot deadlock whatever the residency.
// A refresh with several slabs is one more exchange round (owner waits for workgroups that are all resident).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stmmqr_device.h"
#include "stmmqr_kernels.h"
#include "stmmqr_wave.h"
#include "stmmqr_devutil.h"
#include "stmmqr_riders.h"

#define CA_NT 512
#define CA_LD (STM_CA_R + 2)      // column stride of the slab image (doubles): = 2 mod 32, MFMA operand reads conflict free
#define CA_K 32.0                 // refresh when a column's remaining norm^2 fell below 1/CA_K of its norm^2 at the last Gram

enum { CA_UPDATE = 1, CA_NOUPDATE = 2, CA_REFRESH = 3, CA_DONE = 4 };

struct CaShared {
    double At[STM_NB][STM_NB + 1];      // top block, At[i][x] = F(g1 + i, k1 + x)  (during the chain: in the owner's registers)
    double G[STM_NB][STM_NB + 1];       // Gram matrix of the bottom rows (upper triangle + diagonal are maintained)
    double M[STM_NB][STM_NB + 1];       // pending column operations: current B = stored B * M
    // vectors handed from one column step to the next, double buffered by step parity (a step reads one set and writes
    // the other; ONE barrier per step):
    double part[2][16][STM_NB];         // partial sums (16 row groups) of  sum_{i > pivot row} At(i, j) At(i, x)
    double colA[2][STM_NB], rowA[2][STM_NB];   // column j / pivot row of the top block, current values
    double colN[2][STM_NB];             // column j + 1 before step j's update
    double gj[2][STM_NB], mjv[2][STM_NB];      // row j of G (x >= j), column j of M
    double gref[STM_NB];                // norm^2 of each column (rows >= pivot row) when G was last formed
    double tau[STM_NB];
    int stair[STM_NB], st_out[STM_NB], diag[STM_NB], dead[STM_NB];
    int ctl[8];
};

__device__ __forceinline__ double rdlane(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// G <- S'S over the nr rows of the slab image (rows >= nr are zero).  Waves 0..3 / 4..7 take the two halves of the rows,
// one 16 x 16 tile each; the second half lands in cs.M (scratch: M is not live here) and is added.  Ends with a barrier.
__device__ __forceinline__ void ca_gram(CaShared &cs, const double *S, int nr, double sg)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ti = (wid >> 1) & 1, tj = wid & 1, h = wid >> 2;
    const int nk = (nr + 3) >> 2, kh = (nk + 1) >> 1;
    const int ka = h ? kh : 0, kb = h ? nk : kh;
    d4 acc = {0, 0, 0, 0};
    const double *Sa = S + (16 * ti + l15) * CA_LD + l4, *Sb = S + (16 * tj + l15) * CA_LD + l4;
#pragma unroll 4
    for (int kk = ka; kk < kb; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * kk] * sg, Sb[4 * kk], acc, 0, 0, 0);   // (one operand carries the magnitude guard)
    double (*dst)[STM_NB + 1] = h ? cs.M : cs.G;
#pragma unroll
    for (int r = 0; r < 4; r++) dst[16 * ti + l4 + 4 * r][16 * tj + l15] = acc[r];
    __syncthreads();
    for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) cs.G[e >> 5][e & 31] += cs.M[e >> 5][e & 31];
    __syncthreads();
}

// S <- S M (M upper triangular, in cs.M), 16-row tiles, in place.  Ends with a barrier.
__device__ __forceinline__ void ca_apply(CaShared &cs, double *S, int nr)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double bm0[4], bm1[8];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) bm0[kk] = cs.M[4 * kk + l4][l15];
#pragma unroll
    for (int kk = 0; kk < 8; kk++) bm1[kk] = cs.M[4 * kk + l4][16 + l15];
    const int ntile = (nr + 15) >> 4;
    for (int rt = wid; rt < ntile; rt += CA_NT / 64) {
        double a[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) a[kk] = S[(4 * kk + l4) * CA_LD + 16 * rt + l15];
        d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bm0[kk], u0, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 8; kk++) u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bm1[kk], u1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            S[l15 * CA_LD + 16 * rt + l4 + 4 * r] = u0[r];
            S[(16 + l15) * CA_LD + 16 * rt + l4 + 4 * r] = u1[r];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void ca_set_identity(CaShared &cs)
{
    for (int e = threadIdx.x; e < STM_NB * STM_NB; e += CA_NT) cs.M[e >> 5][e & 31] = ((e >> 5) == (e & 31)) ? 1.0 : 0.0;
}

// whole workgroup waits until *flag >= target (bounded), one lane acquires for the CU
__device__ __forceinline__ bool ca_wait_ge(const int *flag, int target, int *s_ok)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int it = 0; it < (1 << 26); it++) {
            if (ld_agent(flag) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_ok = ok;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return *s_ok != 0;
}

// store the 32 x 32 image in LDS (row stride 33) to global memory write-through; every wave drains its stores, barrier
__device__ __forceinline__ void ca_publish_block(const double (*src)[STM_NB + 1], double *dst)
{
    for (int e = threadIdx.x; e < STM_NB * STM_NB; e += CA_NT) st_agent(&dst[e], src[e >> 5][e & 31]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// the body of k_panel_ca: slab workgroup by of the bx-th front of the launch's lists
__device__ __forceinline__ void dev_k_panel_ca(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int defer_ok,
                                               int bx, int by)
{
    const int p = plist[bx];                                    // every front of a step is at its own panel
    __builtin_amdgcn_s_setprio(3);                              // (the panel chain is the critical path: ahead of the update waves
                                                                //  of the side stream that share the CU)
    extern __shared__ double S[];                       // slab image [STM_NB][CA_LD]
    __shared__ CaShared cs;
    __shared__ int s_ok;
    const int f = flist[bx];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !stm_use_ca(s, p, c.panel_algo, c.ca_min_rows)) return;
    const int w = by;
    const int nwf = stm_ca_slabs(s);                    // symbolic: every workgroup of the launch agrees
    if (w >= nwf) return;
    if ((c.dbg & 2048) && w == ((c.dbg >> 20) & 7)) {   // tests: this slab workgroup starts ~1 ms late
        for (int it = 0; it < 4000; it++) __builtin_amdgcn_s_sleep(100);
    }
    FrontNum *num = &c.fnum[f];
    PanelDesc *pd = &num->pd[STM_PDI(p)];
    double *F = c.Farena + s.foff;
    int *St = c.Stair + s.rp;
    double *Tau = c.Tau + s.rp;
    char *Rdead = c.Rdead + s.col1;
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = num->fm, n = s.fn, npiv = s.fp;
    const int k1 = p * STM_NB, k2 = min(n, k1 + STM_NB), nbp = k2 - k1;
#ifdef STMMQR_STAMPS
    // timeline of panel (dbg >> 24) (dbg & 32): wall clock (100 MHz) of thread 0 of slab workgroup w, dbgbuf[16 + 64 w + idx]
    const bool tl_on = (c.dbg & 32) && c.dbgbuf && p == ((c.dbg >> 24) & 127) && tid == 0 && w < 8;
#define TL(idx) do { if (tl_on) c.dbgbuf[16 + 64 * w + (idx)] = wall_clock64(); } while (0)
#define TC(idx) do { if (tl_on && j == 9) c.dbgbuf[16 + 64 * w + (idx)] = clock64(); } while (0)
#else
#define TC(idx) do { } while (0)
#define TL(idx) do { } while (0)
#endif
    TL(0);

    // ---- header: nothing below is modified before every slab workgroup of this front has taken its ticket ----
    const int was_done = num->done;
    const int g1 = num->g, rank0 = num->rank;
    if (was_done) {
        if (w == 0 && tid == 0) { pd->pnb = 0; pd->t_deferred = 0; }
        return;
    }
    if (g1 >= m) {
        // no rows left before the first column of this panel: remaining pivotal columns are dead, the others empty
        // (:1444-1458).  Every slab workgroup sees the same header; workgroup 0 finalises, nobody takes a ticket.
        if (w == 0) {
            for (int kk = k1 + tid; kk < n; kk += CA_NT) {
                if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
                else St[kk] = m;
                Tau[kk] = 0;
            }
            if (tid < STM_NB) pd->pdiag[tid] = STM_BIGROW;
            if (tid == 0) { num->done = 1; pd->pg1 = g1; pd->pt = g1; pd->pk1 = k1; pd->pnb = 0; pd->pc0 = k2; pd->t_deferred = 0; pd->mode = 0; }
        }
        return;
    }
    const int tmax = min(m, max(St[k2 - 1], g1 + nbp));
    const int nt = min(nbp, tmax - g1);                 // >= 1 (g1 < m)
    const int nB = tmax - g1 - nt;
    const int nwact = max(1, (nB + STM_CA_R - 1) / STM_CA_R);
    const int rbase = g1 + nt + w * STM_CA_R;           // first row of my slab
    const int nr = max(0, min(STM_CA_R, nB - w * STM_CA_R));
    const double sg = c.sig ? c.sig[0] : 1.0, isg = c.sig ? c.sig[1] : 1.0;     // magnitude guard: G and every dot carry ONE factor sg
    const int slot = c.tslot[f];
    double *Gp = c.Gp + (long long)slot * (c.gp_slabs + 1) * (STM_NB * STM_NB);    // partial Gram matrices, then the M mailbox

    // ---- load: my slab (coalesced: a thread per row), the top block, the staircase ----
    if (nr > 0) {
        const int i = min(tid, nr - 1);
        double v[STM_NB];
#pragma unroll
        for (int x = 0; x < STM_NB; x++) v[x] = F[(long long)(k1 + min(x, nbp - 1)) * ld + rbase + i];
        if (tid < STM_CA_R) {                           // (the image has STM_CA_R rows per column; rows >= nr are zero)
#pragma unroll
            for (int x = 0; x < STM_NB; x++) S[x * CA_LD + tid] = (x < nbp && tid < nr) ? v[x] : 0.0;
        }
    }
    for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) {
        const int i = e >> 5, x = e & 31;
        cs.At[i][x] = (i < nt && x < nbp) ? F[(long long)(k1 + x) * ld + g1 + i] : 0.0;
        cs.G[i][x] = 0.0;
    }
    if (tid < STM_NB) cs.stair[tid] = (tid < nbp) ? St[k1 + tid] : 0;
    __syncthreads();
    TL(1);

    // ---- round 0: Gram matrix; with several slabs: exchange, the last arriver owns the chain ----
    if (nr > 0) ca_gram(cs, S, nr, sg);
    TL(2);
    bool owner = true;
    if (nwf > 1) {
        if (nr > 0) ca_publish_block(cs.G, Gp + (long long)w * (STM_NB * STM_NB));
        else __syncthreads();
        if (tid == 0) {
            const int tk = __hip_atomic_fetch_add(&num->gcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cs.ctl[7] = (tk == nwf - 1);
            if (tk == nwf - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_agent(&num->gcnt, 0);
            }
        }
        __syncthreads();
        owner = cs.ctl[7] != 0;
        if (!owner && nr == 0) return;                  // nothing to apply, nothing to own
    }

    TL(3);
    int round = 0;
    if (!owner) {
        // ---- slab workgroup: wait for the owner's rounds ----
        for (;;) {
            if (!ca_wait_ge(&num->prog, STM_PROG * p + 1 + round, &s_ok)) { if (tid == 0) STM_SET_PERR(c, num); return; }
            const int final = ld_agent(&pd->sw);        // (stored before the flag of this round; sw: 0 refresh round, 1 final)
            const double *Mg = Gp + (long long)c.gp_slabs * (STM_NB * STM_NB);
            for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) cs.M[e >> 5][e & 31] = ld_agent(&Mg[e]);
            __syncthreads();
            TL(4 + 2 * round);
            ca_apply(cs, S, nr);
            TL(5 + 2 * round);
            if (final) break;
            ca_gram(cs, S, nr, sg);
            ca_publish_block(cs.G, Gp + (long long)w * (STM_NB * STM_NB));
            if (tid == 0) __hip_atomic_fetch_add(&num->gcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            round++;
        }
        if (tid < nr) {
#pragma unroll 8
            for (int x = 0; x < nbp; x++) F[(long long)(k1 + x) * ld + rbase + tid] = S[x * CA_LD + tid];
        }
        return;
    }

    // ================= owner: the chain =================
    if (nwf > 1) {
        // sum of the partial Gram matrices in slab order (fixed order: deterministic whoever owns)
        for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) {
            double acc = 0;
            if (nB > 0)                                  // (no bottom rows: nobody published anything, G = 0)
                acc = stm_ordered_sum<false>(Gp + e, STM_NB * STM_NB, nwact);
            cs.G[e >> 5][e & 31] = acc;
        }
        __syncthreads();
    }
    if (tid < STM_NB) {
        double a2 = 0;
        for (int i = 0; i < nt; i++) a2 += (cs.At[i][tid] * sg) * cs.At[i][tid];
        cs.gref[tid] = cs.G[tid][tid] + a2;
        cs.tau[tid] = 0; cs.diag[tid] = STM_BIGROW; cs.st_out[tid] = 0; cs.dead[tid] = 0;
    }
    // The chain keeps At, G and M distributed over the registers of the 512 threads -- thread (tx, ig) owns rows ig and
    // ig + 16 of column tx of each -- and only the vectors a column step needs travel through LDS.  EVERY wave computes the
    // scalars of a step for itself (same inputs, same instructions: same bits), so a step is: loads, dlarfg scalars,
    // the updates of the thread's six entries, the partial dots of the next column, stores, ONE barrier.
    const int tx = tid & 31, ig = tid >> 5, i0 = ig, i1 = ig + 16, half = lane >> 5;
    double at0 = cs.At[i0][tx], at1 = cs.At[i1][tx];
    double gq0 = cs.G[i0][tx], gq1 = cs.G[i1][tx];
    double m0 = (i0 == tx) ? 1.0 : 0.0, m1 = (i1 == tx) ? 1.0 : 0.0;
    {
        // what the first column step reads: column 0 / row 0 of the top block, row 0 of G, column 0 of M, the partial dots
        // of column 0 below row 0, column 1
        if (tx == 0) { cs.colA[0][i0] = at0; cs.colA[0][i1] = at1; cs.mjv[0][i0] = m0; cs.mjv[0][i1] = m1; }
        if (tx == 1) { cs.colN[0][i0] = at0; cs.colN[0][i1] = at1; }
        if (i0 == 0) { cs.rowA[0][tx] = at0; cs.gj[0][tx] = gq0; }
        const double c0 = cs.At[i0][0], c1 = cs.At[i1][0];
        cs.part[0][ig][tx] = ((i0 > 0) ? (at0 * sg) * c0 : 0.0) + (at1 * sg) * c1;
    }
    const int stairx = cs.stair[tx];
    __syncthreads();

    TL(4);
    const int ntol = min(c.ntol - s.col1, npiv);
    const double tol = c.tol;
    int j = 0, jref = 0, ncols_done = nbp;
    int g = g1, rank = rank0, done = 0, tlast = g1;        // (every thread advances them identically)
    long long iflops = 0, ilen = 0;                        // the reference's flop count: integers, exact in fp64 (FLOP_COUNT :1571)
    int my_st = 0, my_dead = 0, my_diag = STM_BIGROW;      // lane x keeps the results of panel column x
    double my_tau = 0.0;
    while (j < nbp) {
        if (g >= m) { ncols_done = j; break; }
        const int par = j & 1, np = par ^ 1, jn = j + 1;
        const int gi = g - g1, k = k1 + j;
        TC(40);
        // ---- loads, all unconditional ----
        double pq[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pq[q] = cs.part[par][2 * q + half][tx];
        const double gjx = cs.gj[par][tx], rowx = cs.rowA[par][tx];
        const double cA0 = cs.colA[par][i0], cA1 = cs.colA[par][i1], alpha = cs.colA[par][gi];
        const double mj0 = cs.mjv[par][i0], mj1 = cs.mjv[par][i1];
        double bn0 = cs.colN[par][i0], bn1 = cs.colN[par][i1];
        const double gji0 = cs.gj[par][i0], gji1 = cs.gj[par][i1], gjj = cs.gj[par][j];
        const double grefj = cs.gref[j];
        const int t = max(g + 1, __builtin_amdgcn_readlane(stairx, j));
        double d = ((pq[0] + pq[1]) + (pq[2] + pq[3])) + ((pq[4] + pq[5]) + (pq[6] + pq[7]));
        d = xor32_add(d);
        TC(41);
        const double dj = rdlane(d, j);
        const double ssr = dj + gjj;
        const bool unresolved = (ssr <= 0.0) && !(dj == 0.0 && gjj == 0.0);
        const double ss = fmax(ssr, 0.0);
        const double total = alpha * (alpha * sg) + ss;                      // (one factor sg, as ss and gref)
        TC(42);
        if (j != jref && nB > 0 && (unresolved || grefj > CA_K * total)) {
            // ---- REFRESH: G has lost too much of column j: B <- B M, M <- I, G <- B'B from the real rows ----
            cs.M[i0][tx] = m0; cs.M[i1][tx] = m1;
            cs.At[i0][tx] = at0; cs.At[i1][tx] = at1;
            __syncthreads();
            if (nwf > 1) {
                // one more exchange round: publish M, every slab workgroup applies it, forms its Gram matrix, arrives
                ca_publish_block(cs.M, Gp + (long long)c.gp_slabs * (STM_NB * STM_NB));
                if (tid == 0) { st_agent(&pd->sw, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_agent(&num->prog, STM_PROG * p + 1 + round); }
                round++;
                if (nr > 0) {
                    ca_apply(cs, S, nr);
                    ca_gram(cs, S, nr, sg);
                    ca_publish_block(cs.G, Gp + (long long)w * (STM_NB * STM_NB));
                }
                const int others = nwact - (nr > 0 ? 1 : 0);
                if (!ca_wait_ge(&num->gcnt, others, &s_ok)) { if (tid == 0) STM_SET_PERR(c, num); return; }
                if (tid == 0) st_agent(&num->gcnt, 0);
                for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) {
                    cs.G[e >> 5][e & 31] = stm_ordered_sum<true>(Gp + e, STM_NB * STM_NB, nwact);
                }
                __syncthreads();
            } else {
                ca_apply(cs, S, nr);
                ca_gram(cs, S, nr, sg);
            }
            // back to registers: the fresh G, M = I; republish what the retried step reads (row j of G, column j of M)
            gq0 = cs.G[i0][tx]; gq1 = cs.G[i1][tx];
            m0 = (i0 == tx) ? 1.0 : 0.0; m1 = (i1 == tx) ? 1.0 : 0.0;
            if (i0 == j) cs.gj[par][tx] = gq0;
            if (i1 == j) cs.gj[par][tx] = gq1;
            if (tx == j) { cs.mjv[par][i0] = m0; cs.mjv[par][i1] = m1; }
            if (tid < STM_NB) {
                double a2 = 0;
                for (int i = gi; i < nt; i++) a2 += (cs.At[i][tid] * sg) * cs.At[i][tid];
                cs.gref[tid] = cs.G[tid][tid] + a2;
            }
            __syncthreads();
            jref = j;
            if (tid == 0 && c.dbgbuf && (c.dbg & 16)) atomicAdd(&c.dbgbuf[14], 1ull);        // (diagnosis: refresh rounds)
            TL(30 + (j & 15));
            continue;                                   // (the top block did not change: the partial dots of column j stand)
        }
        // ---- dlarfg; ss == 0 (no row below the diagonal, or all of them zero) gives H = I.
        //      sqrt / reciprocals by v_rsq_f64 / v_rcp_f64 + Newton steps (the library forms cost 700 cycles per column)
        const bool ident = (ss == 0.0);
        double bb, tau0, scal0, scals0;
        stm_larfg_guarded(alpha, ss, sg, isg, bb, tau0, scal0, scals0);
        const double beta = ident ? alpha : bb;
        const bool dead = (k < ntol) && (fabs(beta) <= tol);                 // (:1495-1544) column zeroed, g does not advance
        const bool upd = !ident && !dead;
        const double scal = upd ? scal0 : 0.0, scals = upd ? scals0 : 0.0;   // 1 / (alpha - beta), and the same over sg
        const double tau = upd ? tau0 : 0.0;
        TC(43);
        const bool on = upd && tx > j && tx < nbp;
        const double cwx = on ? tau * (rowx + scals * (d + gjx)) : 0.0;       // tau v'a_x
        const double ccx = cwx * scal;
        const double cwn = rdlane(cwx, jn & 31);                             // (lane jn; 0 when jn == nbp: `on` is false there)
        const double ccA = rdlane(ccx, 2 * wid), ccB = rdlane(ccx, 2 * wid + 1);          // cc[i0]: my row group is 2 wid + half
        const double ccC = rdlane(ccx, (2 * wid + 16) & 31), ccD = rdlane(ccx, (2 * wid + 17) & 31);
        const double ci0 = half ? ccB : ccA, ci1 = half ? ccD : ccC;
        const double v0 = (i0 == gi) ? 1.0 : ((i0 > gi && i0 < nt) ? cA0 * scal : 0.0);
        const double v1 = (i1 == gi) ? 1.0 : ((i1 > gi && i1 < nt) ? cA1 * scal : 0.0);
        // ---- the updates of my entries (cwx = ccx = 0 unless this is a live reflector and my column is behind it) ----
        at0 -= v0 * cwx; at1 -= v1 * cwx;
        bn0 -= v0 * cwn; bn1 -= v1 * cwn;
        m0 -= mj0 * ccx; m1 -= mj1 * ccx;
        gq0 -= (i0 > j && tx >= i0) ? (ci0 * gjx + ccx * gji0 - ci0 * ccx * gjj) : 0.0;
        gq1 -= (i1 > j && tx >= i1) ? (ci1 * gjx + ccx * gji1 - ci1 * ccx * gjj) : 0.0;
        const bool isj = (tx == j);
        // my column is the finished one: beta (0 if dead) on the diagonal, v below it (zeros if dead / H = I)
        const double nd = dead ? 0.0 : beta;
        at0 = isj ? ((i0 < gi) ? at0 : ((i0 == gi) ? nd : (upd ? v0 : 0.0))) : at0;
        at1 = isj ? ((i1 < gi) ? at1 : ((i1 == gi) ? nd : (upd ? v1 : 0.0))) : at1;
        m0 = isj ? (upd ? m0 * scal : 0.0) : m0;                              // b_j <- b_j scal; dead / identity: zero column
        m1 = isj ? (upd ? m1 * scal : 0.0) : m1;
        const int gin = dead ? gi : gi + 1;
        cs.part[np][ig][tx] = ((i0 > gin) ? (at0 * sg) * bn0 : 0.0) + ((i1 > gin) ? (at1 * sg) * bn1 : 0.0);
        if (tx == jn) { cs.colA[np][i0] = at0; cs.colA[np][i1] = at1; cs.mjv[np][i0] = m0; cs.mjv[np][i1] = m1; }
        if (tx == jn + 1) { cs.colN[np][i0] = at0; cs.colN[np][i1] = at1; }
        if (i0 == gin) cs.rowA[np][tx] = at0;
        if (i1 == gin) cs.rowA[np][tx] = at1;
        if (i0 == jn) cs.gj[np][tx] = gq0;
        if (i1 == jn) cs.gj[np][tx] = gq1;
        // ---- bookkeeping (registers) ----
        my_st = isj ? (dead ? 0 : t) : my_st;
        my_dead = isj ? (dead ? 1 : 0) : my_dead;
        my_diag = isj ? (dead ? STM_BIGROW : g) : my_diag;
        my_tau = isj ? tau : my_tau;
        if (!dead) {
            iflops += (long long)(t - g) * (3 + 4 * (long long)(n - k - 1));
            ilen += (t - g);
            tlast = t;
            g++;
        }
        if (k == npiv - 1) rank = g;                      // (:1604-1608) also taken on a dead last pivot
        TC(46);
        __syncthreads();
        TC(47);
        j++;
        if ((j & 7) == 0) TL(8 + (j >> 3));
    }
    // the distributed images and the per-column results go back to LDS for the final application / stores
    cs.M[i0][tx] = m0; cs.M[i1][tx] = m1;
    cs.At[i0][tx] = at0; cs.At[i1][tx] = at1;
    if (tid < STM_NB) { cs.st_out[tid] = my_st; cs.dead[tid] = my_dead; cs.diag[tid] = my_diag; cs.tau[tid] = my_tau; }
    const double flops = (double)iflops, lensum = (double)ilen;
    if (tid == 0 && c.dbgbuf && (c.dbg & 16)) { atomicAdd(&c.dbgbuf[13], 1ull); atomicAdd(&c.dbgbuf[15], (unsigned long long)nwact); }   // (panels, slabs)
    __syncthreads();
    TL(20);
    if (ncols_done < nbp || g >= m) {
        // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458; when the rows run out
        // with the last column of the panel the reference notices at the next column: same result)
        done = 1;
        for (int kk = k1 + ncols_done + tid; kk < n; kk += CA_NT) {
            if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
            else St[kk] = m;
            Tau[kk] = 0;
        }
        // (their slab columns: whatever is there stays -- those rows do not exist: g >= m means tmax <= m <= g)
    }

    // ---- final round: B <- B M everywhere ----
    if (nwf > 1) {
        ca_publish_block(cs.M, Gp + (long long)c.gp_slabs * (STM_NB * STM_NB));
        if (tid == 0) { st_agent(&pd->sw, 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_agent(&num->prog, STM_PROG * p + 1 + round); }
    }
    TL(21);
    if (nr > 0) {
        ca_apply(cs, S, nr);
        TL(22);
        if (tid < nr) {
#pragma unroll 8
            for (int x = 0; x < nbp; x++) F[(long long)(k1 + x) * ld + rbase + tid] = S[x * CA_LD + tid];
        }
    }
    TL(23);
    // ---- the owner's results: top block, HStair / HTau / Rdead, the pending block reflector ----
    for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) {
        const int i = e & 31, x = e >> 5;
        if (i < nt && x < nbp) F[(long long)(k1 + x) * ld + g1 + i] = cs.At[i][x];
    }
    if (tid < ncols_done) {
        St[k1 + tid] = cs.st_out[tid];
        Tau[k1 + tid] = cs.tau[tid];
        if (cs.dead[tid]) Rdead[k1 + tid] = 1;
    }
    if (tid < STM_NB) pd->pdiag[tid] = (tid < ncols_done) ? cs.diag[tid] : STM_BIGROW;
    const bool live = tlast > g1;
    // T: left to the trailing update when one follows (k_upd_w / dev_update_block build it from V'V); the last panel of a
    // front (no trailing columns) gets it from k_cpack's extra workgroup (t_deferred = 2); no live reflector: T = 0 here
    const int tdef = !live ? 0 : ((defer_ok && k2 < n) ? 1 : 2);
    if (!live) {
        double *Tout = c.Tws + (long long)STM_TSLOT(slot, p) * STM_NB * STM_NB;
        double *Tkeep = c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr;
        for (int e = tid; e < STM_NB * STM_NB; e += CA_NT) { Tout[e] = 0.0; if (Tkeep) Tkeep[e] = 0.0; }
    }
    if (wid == 0 && lane == 0) {
        num->g = g; num->rank = rank; num->done = done;
        num->flops += flops;
        num->flops_upd += 4.0 * (double)(n - k2) * lensum;
        pd->pg1 = g1; pd->pt = tlast; pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2; pd->t_deferred = tdef; pd->mode = 0;
    }
    TL(24);
#undef TL
#undef TC
}

__global__ __launch_bounds__(CA_NT) void k_panel_ca(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int defer_ok)
{
    dev_k_panel_ca(c, flist, plist, defer_ok, blockIdx.x, blockIdx.y);
}

// A step whose panels are all Gram-based has no k_panel_pc launch for the k_upd_c riders of the step before it (passenger launches,
// stmmqr_riders.h): they ride here instead.  (They ran as a launch of their own before: 77 x 26 us on the chain of the default
// workload.)  A ONE-dimensional grid: the first npan * nw workgroups are the panel's slab workgroups in the order of k_panel_ca's own
// grid (front fastest), the riders follow (column block fastest, then slab group, then front of the PREVIOUS step's lists).  The slab
// workgroups wait for each other, one per CU: consecutive workgroup ids go round the XCDs, so they spread over the chip as in their own
// launch -- as a sub-block of a 3-D grid their ids were a row length apart, all on one XCD, more than it has CUs (measured: the wait
// ran out on c5mini).
__global__ __launch_bounds__(CA_NT) void k_panel_ca_pc(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int npan, int nw,
                                                     int defer_ok, const int *__restrict__ uflist, const int *__restrict__ uplist, int ucb0,
                                                     int uncb, int uy, const double *Wp, const long long *__restrict__ uwpoff, int rspw)
{
    extern __shared__ double ca_dyn[];
    __shared__ int s_pdr[STM_NB];
    const int lin = blockIdx.x;
    if (lin < npan * nw) {
        dev_k_panel_ca(c, flist, plist, defer_ok, lin % npan, lin / npan);
        return;
    }
    const int r = lin - npan * nw;
    dev_upd_c_h2(c, uflist, uplist, ucb0, Wp, uwpoff, r / (uncb * uy), r % uncb, (r / uncb) % uy, rspw, ca_dyn, s_pdr);
}

int stm_ca_lds_bytes(void) { return (int)(STM_NB * CA_LD * sizeof(double)); }
static size_t ca_pc_lds_bytes(void)
{
    const size_t a = (size_t)stm_ca_lds_bytes(), b = (size_t)STM_PC_LDS_DOUBLES * sizeof(double);
    return a > b ? a : b;
}

int stm_configure_capanel(void)
{
    hipError_t e = hipFuncSetAttribute((const void *)k_panel_ca, hipFuncAttributeMaxDynamicSharedMemorySize, stm_ca_lds_bytes());
    if (e != hipSuccess) return (int)e;
    return (int)hipFuncSetAttribute((const void *)k_panel_ca_pc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ca_pc_lds_bytes());
}

// the Gram-based panels of a step + the k_upd_c riders of the step before it (one launch; more fronts than fit a launch: the rest alone)
int stm_launch_panel_ca_pc(const DevCtx &c, const int *flist, const int *plist, int nfr, int nw, int defer_ok, const int *uflist,
                           const int *uplist, int unfr, int ucb0, int uncb, int umaxsl, const double *Wp, const long long *uwpoff, hipStream_t st)
{
    if (nfr <= 0 || unfr <= 0 || uncb <= 0 || umaxsl <= 0) return -1;     // (the caller launches the two separately)
    if (nw < 1) nw = 1;
    int K = 240 / nw;
    if (K < 1) K = 1;
    const int n0 = nfr < K ? nfr : K;
    int rspw = 1;
    {
        static int force = -1;
        if (force < 0) force = getenv("STMMQR_RSPW") ? atoi(getenv("STMMQR_RSPW")) : 0;
        double best = 1e30;
        for (int k = 1; k <= 16; k *= 2) {
            const long wgs = (long)unfr * uncb * ((umaxsl + k - 1) / k);
            const double t = (double)((wgs + 239) / 240) * (5.0 + 1.0 * k);
            if (t < best) { best = t; rspw = k; }
        }
        if (force > 0) rspw = force;
    }
    const int uy = (umaxsl + rspw - 1) / rspw;
    const long total = (long)n0 * nw + (long)uncb * uy * unfr;
    if (total > 0x7fffffffL) return -1;
    hipLaunchKernelGGL(k_panel_ca_pc, dim3((unsigned)total), dim3(CA_NT), ca_pc_lds_bytes(), st, c, flist, plist, n0, nw, defer_ok, uflist, uplist,
                       ucb0, uncb, uy, Wp, uwpoff, rspw);
    for (int i = n0; i < nfr; i += K)
        hipLaunchKernelGGL(k_panel_ca, dim3(nfr - i < K ? nfr - i : K, nw), dim3(CA_NT), (size_t)stm_ca_lds_bytes(), st, c, flist + i, plist + i,
                           defer_ok);
    return (int)hipGetLastError();
}

int stm_launch_panel_ca(const DevCtx &c, const int *flist, const int *plist, int nfr, int nw, int defer_ok, hipStream_t st)
{
    if (nfr <= 0) return 0;
    if (nw < 1) nw = 1;
    // one slab workgroup per CU (LDS); the ticket protocol needs no co-residency, but a launch is kept within the chip
    // anyway so that nobody spins while its partners wait for a CU
    int K = 240 / nw;
    if (K < 1) K = 1;
    for (int i = 0; i < nfr; i += K)
        hipLaunchKernelGGL(k_panel_ca, dim3(nfr - i < K ? nfr - i : K, nw), dim3(CA_NT), (size_t)stm_ca_lds_bytes(), st, c, flist + i, plist + i,
                           defer_ok);
    return (int)hipGetLastError();
}
