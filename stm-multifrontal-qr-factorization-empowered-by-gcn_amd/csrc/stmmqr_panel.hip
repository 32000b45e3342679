// stmmqr_panel.hip -- the Householder panels of the large fronts and the whole small fronts (qr_front,
// STMMQR/src/qr/SparseQR_factorize.c:1383-1618): dev_panel (one workgroup, LDS), dev_tall_group (column pipeline over workgroups),
// dev_wave_panel (a wave per 4 columns), k_front_wg, k_panel, k_panel_pc (panel + k_upd_c riders).  Shared device code: stmmqr_kdev.h.
#include "stmmqr_kdev.h"
#include "stmmqr_riders.h"


// ------------------------------------------------------------------------------------------------
// Apply the reflectors of an LDS-resident sub-panel to the remaining columns of the panel (qr_private_apply1 /
// dlarf semantics, :1359-1381, reflector after reflector).  One wave owns one column: the column's active rows
// [gs, r1) live in that wave's registers (<= 64 per lane, i.e. 4096 rows), V is read from LDS with the
// unit-diagonal / zero mask, each v'c is a wave64 shuffle reduction.  Columns taller than 4096 active rows use
// the slower two-pass form on the column in global memory.
//   Vl[(i - gs) + j*pst] = F(i, k1 + j0 + j) for the sub-panel columns j < sw.
// ------------------------------------------------------------------------------------------------
template <int NTH>
__device__ __forceinline__ void dev_apply_subpanel(const double *Vl, long long pst, int gs, int r1, int sw, const int *diag,
                                   const double *tau, double *Fc /* = &F(0, first remaining column) */, long long ld,
                                   int ncols)
{
    constexpr int NWV = NTH / 64;
    constexpr int MAXQ = 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rows = r1 - gs;
    if (rows <= 0) return;
    const int nq = (rows + 63) >> 6;
    for (int jc = wid; jc < ncols; jc += NWV) {
        double *cg = Fc + jc * ld + gs;                 // cg[r] = F(gs + r, column)
        if (nq <= MAXQ && NTH >= 512) {
            double creg[MAXQ];
#pragma unroll
            for (int q = 0; q < MAXQ; q++) {
                const int r = lane + 64 * q;
                creg[q] = (q < nq && r < rows) ? cg[r] : 0.0;
            }
            for (int i = 0; i < sw; i++) {
                const double ti = tau[i];
                if (ti == 0.0) continue;
                const double *v = Vl + i * pst + lane;  // explicit unit-lower-trapezoidal image: no masks
                double z = 0;
#pragma unroll
                for (int q = 0; q < MAXQ; q++)
                    if (q < nq) z += v[64 * q] * creg[q];
                z = wave_sum(z) * ti;
#pragma unroll
                for (int q = 0; q < MAXQ; q++)
                    if (q < nq) creg[q] -= z * v[64 * q];
            }
#pragma unroll
            for (int q = 0; q < MAXQ; q++) {
                const int r = lane + 64 * q;
                if (q < nq && r < rows) cg[r] = creg[q];
            }
        } else {
            for (int i = 0; i < sw; i++) {
                const double ti = tau[i];
                if (ti == 0.0) continue;
                const double *v = Vl + i * pst;
                double z = 0;
                for (int r = lane; r < rows; r += 64) z += v[r] * cg[r];
                z = wave_sum(z) * ti;
                for (int r = lane; r < rows; r += 64) cg[r] -= z * v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// workgroup-wide sum of K values per thread; every thread gets all K sums.  s_part: (NTH/64)*K doubles.
// ------------------------------------------------------------------------------------------------
template <int NTH, int K>
__device__ __forceinline__ void block_reduce_vec(double (&x)[K], double *s_part)
{
    constexpr int NWV = NTH / 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K; i++) x[i] = wave_sum(x[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < K; i++) s_part[wid * K + i] = x[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; i++) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) r += s_part[w * K + i];
        x[i] = r;
    }
}

// rows [g, tmax) a panel will touch (uniform over the workgroup; 0 when the front is already finished)
__device__ __forceinline__ int panel_rows(const FrontSym &s, const FrontNum *num, const int *St, int p)
{
    if (num->done) return 0;
    const int k1 = p * STM_NB, k2 = min(s.fn, k1 + STM_NB);
    const int g = num->g;
    return max(0, min(num->fm, max(St[k2 - 1], g + (k2 - k1))) - g);
}

// ------------------------------------------------------------------------------------------------
// Register-resident sub-panel (qr_front's column loop, reference :1434-1609, for a sub-panel of <= 8 columns whose
// active rows fit RPT rows per thread).  Thread tid owns rows gs + tid + NTH*r of ALL sub-panel columns, so a
// column step touches no memory except its one workgroup reduction:
//   pass 1 : part[x] = sum_{g<i<t} F(i,k) F(i,k+x)   (x = 0: |x|^2 of dlarfg; x > 0: the v'c of dlarf, unscaled)
//            one 8-value workgroup reduction (halving butterfly + one LDS exchange, ONE barrier, double-buffered)
//   scalar : beta / tau / 1/(alpha-beta), dead-column test -- every thread redundantly
//   pass 2 : v = x * scal ;  c_x -= tau (top_x + scal part_x) v   for the remaining columns of the sub-panel
// The current column always lives in register column 0: a finished column is retired to the LDS image
// (lds[(i-gs) + j*pst], the layout dev_panel's write-back / apply tail expects) and the register columns rotate
// down by one, so the loop body exists once (an unrolled body per column overflows the instruction cache) and every
// register index is static.  State (g, rank, ...) follows dev_panel's conventions.
// ------------------------------------------------------------------------------------------------
template <int NTH, int RPT, int SWT>
__device__ __forceinline__ void dev_subpanel_reg(PanelShared &ps, double *F, long long ld, int *St, double *Tau, char *Rdead,
                                                 int k1, int j0, int sw, int nbp, int gs, int tmax, int m, int n, int npiv,
                                                 int ntol, double tol, int &g, int &rank, double &flops, double &lensum, int &nlive,
                                                 int &tlast, int &done, int &ncols_done, double *lds, long long pst, double sg, double isg)
{
    constexpr int NWV = NTH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int msp = (max(tmax - gs, 0) + 63) & ~63;
    double a[RPT][SWT];
#pragma unroll
    for (int x = 0; x < SWT; x++) {
        const double *src = F + (long long)(k1 + j0 + min(x, sw - 1)) * ld;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int i = gs + tid + NTH * r;
            const double val = src[max(min(i, tmax - 1), 0)];           // unconditional load, masked afterwards
            a[r][x] = (x < sw && i < tmax) ? val : 0.0;
        }
    }
    for (int j = 0; j < sw; j++) {
        const int jp = j0 + j, k = k1 + jp;
        if (!done && g >= m) {
            // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
            for (int kk = k + tid; kk < n; kk += NTH) {
                if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
                else St[kk] = m;
                Tau[kk] = 0;
            }
            for (int jj = jp + tid; jj < nbp; jj += NTH) { ps.diag[jj] = STM_BIGROW; ps.tau[jj] = 0; }
            done = 1;
            ncols_done = jp;
        }
        if (!done) {
            const int t = max(g + 1, ps.stair[jp]);
            const int par = j & 1;
            double part[8];
#pragma unroll
            for (int x = 0; x < 8; x++) part[x] = 0;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = gs + tid + NTH * r;
                const double xv = (i > g && i < t) ? a[r][0] * sg : 0.0;    // (one operand carries the magnitude guard)
#pragma unroll
                for (int x = 0; x < SWT; x++) part[x] += xv * a[r][x];      // (part[SWT..7] stay zero)
            }
            const bool owner = (tid == g - gs);         // holds row g in a[0][.]  (g - gs < 8 <= NTH)
            if (owner) {
#pragma unroll
                for (int x = 0; x < SWT; x++) ps.top[par][x] = a[0][x];
            }
            const double rw = wave_reduce8(part);
            if (lane < 8) ps.rsum[par][wid * 8 + lane] = rw;
            __syncthreads();
            const double rb = wave_sum_stride8((lane < NWV * 8) ? ps.rsum[par][lane] : 0.0);
            double sum[8];
            sum[0] = lane_bcast<red8_lane(0)>(rb); sum[1] = lane_bcast<red8_lane(1)>(rb);
            sum[2] = lane_bcast<red8_lane(2)>(rb); sum[3] = lane_bcast<red8_lane(3)>(rb);
            sum[4] = lane_bcast<red8_lane(4)>(rb); sum[5] = lane_bcast<red8_lane(5)>(rb);
            sum[6] = lane_bcast<red8_lane(6)>(rb); sum[7] = lane_bcast<red8_lane(7)>(rb);
            const double alpha = ps.top[par][0];
            const double ss = sum[0];
            double tau = 0, beta = alpha, scal = 0, scals = 0;
            if (ss != 0.0) stm_larfg_guarded(alpha, ss, sg, isg, beta, tau, scal, scals);   // (no active row below the diagonal: ss == 0 exactly)
            const bool dead = (k < ntol) && (fabs(beta) <= tol);
            if (dead) {
                // zero the column from the diagonal down, no reflector, g does not advance (:1495-1544)
#pragma unroll
                for (int r = 0; r < RPT; r++)
                    if (gs + tid + NTH * r >= g) a[r][0] = 0.0;
                if (tid == 0) { ps.st_out[jp] = 0; ps.dead[jp] = 1; ps.diag[jp] = STM_BIGROW; ps.tau[jp] = 0; }
                if (k == npiv - 1) rank = g;            // (:1604-1608) also taken on a dead last pivot
            } else {
                if (tid == 0) { ps.st_out[jp] = t; ps.dead[jp] = 0; ps.diag[jp] = g; ps.tau[jp] = tau; }
                flops += (double)(t - g) * (3.0 + 4.0 * (double)(n - k - 1));
                lensum += (double)(t - g);
                if (tau != 0.0) {
                    nlive++;
                    double w[8];
#pragma unroll
                    for (int x = 1; x < SWT; x++) w[x] = tau * (ps.top[par][x] + scals * sum[x]);
#pragma unroll
                    for (int r = 0; r < RPT; r++) {
                        const int i = gs + tid + NTH * r;
                        if (i > g && i < t) {
                            const double v = a[r][0] * scal;
                            a[r][0] = v;
#pragma unroll
                            for (int x = 1; x < SWT; x++) a[r][x] -= w[x] * v;
                        }
                    }
                    if (owner) {
#pragma unroll
                        for (int x = 1; x < SWT; x++) a[0][x] -= w[x];
                    }
                }
                if (owner) a[0][0] = beta;
                tlast = t;
                g++;
                if (k == npiv - 1) rank = g;
            }
        }
        // ---- retire register column 0 to the LDS image and rotate ----
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int il = tid + NTH * r;               // row - gs
            if (il < msp) lds[il + j * pst] = a[r][0];
#pragma unroll
            for (int x = 0; x + 1 < SWT; x++) a[r][x] = a[r][x + 1];
            a[r][SWT - 1] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// qr_front, one panel of <= STM_NB columns (reference: the column loop :1434-1609 with the panel policy
// fixed to k1 = p*NB, which changes rounding only).  Executed by one whole workgroup.
//
// The panel F(g:tmax, k1:k2) is processed in sub-panels of `sw` columns that fit the LDS budget (a 4000-row panel
// gets 4-column sub-panels, a 500-row panel is one sub-panel): each sub-panel is staged in LDS, reduced column
// by column there (norms: wave64 shuffles + one LDS step; dlarf: one wave per remaining sub-panel column),
// written back, and its block reflector is applied to the remaining columns of the panel with the fp64-MFMA
// routine above -- a blocked QR inside the panel, so the tall panel streams through the CU once per sub-panel
// instead of once per column.  Panels taller than the LDS budget for one column fall back to in-place work.
// Produces: R and V in F, Tau, Stair, Rdead, the T factor of the whole panel (Tout, NB x NB) and the pending
// block-reflector description in FrontNum (pg1, pt, pk1, pnb, pc0, pdiag).
// ------------------------------------------------------------------------------------------------
template <int NTH, bool INPLACE>
__device__ __forceinline__ void dev_panel(PanelShared &ps, const FrontSym &s, FrontNum *num, double *F, int *St, double *Tau, char *Rdead,
                          int p, double tol, int ntol_global, double *Tout, double *lds, int lds_doubles, int dbg = 0,
                          unsigned long long *dbgbuf = nullptr, double *Tkeep = nullptr, const double *sigp = nullptr)
{
    constexpr int NWV = NTH / 64;
    double *s_red = ps.red;
    int *s_diag = ps.diag;
    double *s_tau = ps.tau;
    double (*s_G)[STM_NB + 1] = ps.G;
    double (*s_T)[STM_NB + 1] = ps.T;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = num->fm, n = s.fn, npiv = s.fp;
    const long long ld = s.ld;
    const int k1 = p * STM_NB;
    const int k2 = min(n, k1 + STM_NB);
    const int nbp = k2 - k1;
    const int was_done = num->done;
    int g = num->g, rank = num->rank;
    __syncthreads();                                   // everyone has read FrontNum before anyone writes it
    if (was_done) {
        if (tid == 0) num->pd[STM_PDI(p)].pnb = 0;
        return;
    }
    const int ntol = min(ntol_global - s.col1, npiv);
    const int g1 = g;
    const double sg = sigp ? sigp[0] : 1.0, isg = sigp ? sigp[1] : 1.0;      // magnitude guard (stm_larfg_guarded)
    if (tid == 0) ps.nextss_col = -1;
    if (tid < nbp) ps.stair[tid] = St[k1 + tid];       // the panel's staircase, once (a global load per column
    __syncthreads();                                   //  step would sit on the critical path)
    const int tmax = min(m, max(ps.stair[nbp - 1], g1 + nbp));
    const int mp = tmax - g1;
    double flops = 0, lensum = 0;
    int done = 0, tlast = g1, ncols_done = nbp;

    // sub-panel width: as many columns as fit in LDS with the rows of the first sub-panel
    // INPLACE (rows too tall for even one LDS column, or nothing to do): work straight on F; otherwise sub-panels of
    // SW columns in LDS.  The two cases are separate instantiations so that every access of the LDS case is a
    // ds_* instruction (a pointer that may be either LDS or global compiles to slow FLAT accesses).
    int SW = INPLACE ? nbp : max(1, min(nbp, lds_doubles / (((mp + 63) & ~63) | 1)));
    constexpr bool in_place = INPLACE;
    // tall sub-panels: all waves split the ROWS of every column step and one workgroup reduction delivers the
    // column norm and all v'c dot products of the rest of the sub-panel (<= 7) at once
    const bool tall = !in_place && mp > 768;
    if (tall) SW = min(SW, 8);
    // register-resident sub-panels (<= 8 rows per thread, <= 8 columns): dev_subpanel_reg
    const int reg_min = 768;
    const bool regpath = !in_place && mp > reg_min && mp <= 8 * NTH && !(dbg & 256);
    if (regpath) SW = min(SW, mp > 4 * NTH ? 4 : 8);     // 8 rows per thread leave registers for 4 columns only

#ifdef STMMQR_STAMPS                                    /* phase timers (debug builds only: they cost 24 live VGPRs) */
    unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tc0 = 0, tc1 = 0;
#define CSTAMP(idx) do { if (dbg & 32) { const unsigned long long t_ = clock64(); tph[idx] += t_ - tc1; tc1 = t_; } } while (0)
#define STAMP(idx) do { if (dbg & 16) { __syncthreads(); const unsigned long long t_ = clock64(); tph[idx] += t_ - tc0; tc0 = t_; } } while (0)
    if (dbg & 16) tc0 = clock64();
#else
#define CSTAMP(idx) do { } while (0)
#define STAMP(idx) do { } while (0)
#endif
    for (int j0 = 0; j0 < nbp && !done; j0 += SW) {
        const int sw = min(SW, nbp - j0);
        const int gs = g;                               // first active row of this sub-panel
        const int ms = tmax - gs;
        long long pst;
        double *Pb;                                     // Pb[(i-roff) + (j-coff)*pst] = F(i, k1+j)
        int roff = 0, coff = 0;                         // (never form a pointer outside the LDS object)
        bool use_reg = false;
        if constexpr (!INPLACE) {
            const int msp = (max(ms, 0) + 63) & ~63;    // zero-padded so that 64-row slots can be read unmasked
            pst = msp | 1;
            use_reg = regpath && ms <= 8 * NTH;
            if (!use_reg)
                for (int j = wid; j < sw; j += NWV) {
                    const double *src = F + (k1 + j0 + j) * ld;
                    // (unconditional loads on a clamped row, masked afterwards: see dev_tall_group)
                    for (int i = lane; i < msp; i += 64) {
                        const double val = src[gs + max(min(i, ms - 1), 0)];
                        lds[i + j * pst] = (i < ms) ? val : 0.0;
                    }
                }
            Pb = lds; roff = gs; coff = j0;
        } else {
            pst = ld;
            Pb = F + k1 * ld;
        }
        int nlive = 0;
        if (use_reg) {
            // ---- register-resident column loop; leaves the finished sub-panel in the LDS image ----
            if (ms <= NTH)
                dev_subpanel_reg<NTH, 1, 8>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
            else if (ms <= 2 * NTH)
                dev_subpanel_reg<NTH, 2, 8>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
            else if (ms <= 4 * NTH)
                dev_subpanel_reg<NTH, 4, 8>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
            else
                dev_subpanel_reg<NTH, 8, 4>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
        } else {
        __syncthreads();
        STAMP(0);
        for (int j = j0; j < j0 + sw; j++) {
            const int k = k1 + j;
            if (g >= m) {
                // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
                for (int kk = k + tid; kk < n; kk += NTH) {
                    if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
                    else St[kk] = m;
                    Tau[kk] = 0;
                }
                for (int jj = j + tid; jj < nbp; jj += NTH) { s_diag[jj] = STM_BIGROW; s_tau[jj] = 0; }
                done = 1;
                ncols_done = j;                         // columns >= j were finalised above, straight in global memory
                break;
            }
            const int t = max(g + 1, ps.stair[j]);
            double *col = Pb + (g - roff) + (j - coff) * pst;   // col[0] = F(g,k)
            const int len = t - g;                      // >= 1
            // ---- dlarfg ----
#ifdef STMMQR_STAMPS
            if (dbg & 32) tc1 = clock64();
#endif
            const double alpha = col[0];                // read before the barriers below: thread 0 overwrites it
            const int nrest = tall ? (j0 + sw - 1 - j) : 0;     // remaining sub-panel columns (tall scheme, <= 7)
            double part[8];
            double top[8];
#pragma unroll
            for (int x = 0; x < 8; x++) { part[x] = 0; top[x] = 0; }
            if (tall) {
#pragma unroll
                for (int x = 1; x < 8; x++)
                    if (x <= nrest) top[x] = col[x * pst];      // F(g, k+x)
                for (int i = 1 + tid; i < len; i += NTH) {
                    const double x0 = col[i], xv = x0 * sg;         // (one operand carries the magnitude guard)
                    part[0] += xv * x0;
#pragma unroll
                    for (int x = 1; x < 8; x++)
                        if (x <= nrest) part[x] += xv * col[i + x * pst];
                }
                CSTAMP(5);
                block_reduce_vec<NTH, 8>(part, ps.part);
                CSTAMP(6);
            } else if (ps.nextss_col == j && !(dbg & 128)) {
                part[0] = ps.nextss;                    // computed by the dlarf sweep of the previous column
                __syncthreads();                        // every wave has read alpha / nextss before anyone writes
            } else {
                double ss0 = 0;
                for (int i = 1 + tid; i < len; i += NTH) { const double xv = col[i]; ss0 += (xv * sg) * xv; }
                part[0] = block_sum<NTH>(ss0, s_red);
            }
            const double ss = part[0];
            double tau = 0, beta = alpha, scal = 0, scals = 0;
            if (len > 1 && ss != 0.0) stm_larfg_guarded(alpha, ss, sg, isg, beta, tau, scal, scals);
            const bool dead = (k < ntol) && (fabs(beta) <= tol);
            if (dead) {
                // zero the column from the diagonal down, no reflector, g does not advance (:1495-1544)
                for (int i = tid; i < tmax - g; i += NTH) col[i] = 0;
                if (tid == 0) { ps.st_out[j] = 0; ps.dead[j] = 1; s_diag[j] = STM_BIGROW; s_tau[j] = 0; }
                if (k == npiv - 1) rank = g;            // (:1604-1608) also taken on a dead last pivot
                __syncthreads();
                continue;
            }
            // (col[0] = beta is stored only after a barrier that every reader of alpha = col[0] has passed)
            if (tid == 0) { ps.st_out[j] = t; ps.dead[j] = 0; s_diag[j] = g; s_tau[j] = tau; if (tall) col[0] = beta; }
            CSTAMP(7);
            flops += (double)len * (3.0 + 4.0 * (double)(n - k - 1));
            lensum += (double)len;
            if (tall) {
                // ---- scale x and apply H_k to the rest of the sub-panel in the same sweep over the rows ----
                if (tau != 0.0) {
                    nlive++;
                    double w[8];
#pragma unroll
                    for (int x = 1; x < 8; x++) w[x] = tau * (top[x] + scals * part[x]);
                    for (int i = 1 + tid; i < len; i += NTH) {
                        const double v = col[i] * scal;
                        col[i] = v;
#pragma unroll
                        for (int x = 1; x < 8; x++)
                            if (x <= nrest) col[i + x * pst] -= w[x] * v;
                    }
                    if (tid == 0) {
#pragma unroll
                        for (int x = 1; x < 8; x++)
                            if (x <= nrest) col[x * pst] -= w[x];
                    }
                }
            } else {
                if (tau != 0.0)
                    for (int i = 1 + tid; i < len; i += NTH) col[i] *= scal;
                __syncthreads();
                if (tid == 0) col[0] = beta;            // no one reads F(g,k) any more in this step
                // ---- dlarf on the rest of the SUB-panel: one wave per column, v'c by DPP reduction (A.3) ----
                if (tau != 0.0) {
                    nlive++;
                    if (!(dbg & 4))
                    for (int jj = j + 1 + wid; jj < j0 + sw; jj += NWV) {
                        double *cc = Pb + (g - roff) + (jj - coff) * pst;
                        double w = (lane == 0) ? cc[0] : 0.0;
                        {
                            // four independent partial sums: the LDS read latency overlaps instead of chaining
                            double w1 = 0, w2 = 0, w3 = 0;
                            int i = 1 + lane;
                            for (; i + 192 < len; i += 256) {
                                w += col[i] * cc[i];
                                w1 += col[i + 64] * cc[i + 64];
                                w2 += col[i + 128] * cc[i + 128];
                                w3 += col[i + 192] * cc[i + 192];
                            }
                            for (; i < len; i += 64) w += col[i] * cc[i];
                            w += w1 + w2 + w3;
                        }
                        w = wave_sum(w) * tau;
                        if (lane == 0) cc[0] -= w;
                        if (jj == j + 1) {
                            // the next column: update it and accumulate its |x|^2 (rows below ITS diagonal g+1,
                            // up to its own staircase) in the same sweep
                            const int lenn = max(g + 2, ps.stair[jj]) - g;      // rows g .. g+lenn-1
                            double sq = 0;
                            for (int i = 1 + lane; i < max(len, lenn); i += 64) {
                                double cv = cc[i];
                                if (i < len) { cv -= w * col[i]; cc[i] = cv; }
                                if (i >= 2 && i < lenn) sq += (cv * sg) * cv;
                            }
                            sq = wave_sum(sq);
                            if (lane == 0) { ps.nextss = sq; ps.nextss_col = jj; }
                        } else {
                            for (int i = 1 + lane; i < len; i += 64) cc[i] -= w * col[i];
                        }
                    }
                }
            }
            tlast = t;
            g++;
            if (k == npiv - 1) rank = g;
            __syncthreads();
            CSTAMP(8);
        }
        }
        __syncthreads();
        STAMP(1);
        if (!in_place) {
            for (int j = wid; j < sw; j += NWV) {
                double *dst = F + (k1 + j0 + j) * ld;
                for (int i = gs + lane; i < tmax; i += 64) dst[i] = lds[(i - gs) + j * pst];
            }
        }
        __syncthreads();
        STAMP(2);
        // ---- apply this sub-panel's reflectors to the remaining columns of the panel; V is still in LDS ----
        if (!in_place && !done && j0 + sw < nbp && nlive > 0 && tlast > gs && !(dbg & 1)) {
            // turn the LDS image into the explicit unit-lower-trapezoidal V (R entries above the diagonals were
            // already written back): the apply loop then needs no masks
            for (int e = tid; e < sw * sw; e += NTH) {
                const int i = e % sw, j = e / sw;       // rows gs..gs+sw-1 are the only ones at or above a diagonal
                const int d = s_diag[j0 + j] - gs;
                if (i < ms) lds[i + j * pst] = (s_tau[j0 + j] == 0.0) ? 0.0 : ((i < d) ? 0.0 : ((i == d) ? 1.0 : lds[i + j * pst]));
            }
            if (nlive < sw)                              // dead / identity columns: whole column is not a reflector
                for (int j = 0; j < sw; j++)
                    if (s_tau[j0 + j] == 0.0)
                        for (int i = tid; i < ms; i += NTH) lds[i + j * pst] = 0.0;
            __syncthreads();
            dev_apply_subpanel<NTH>(lds, pst, gs, tlast, sw, s_diag + j0, s_tau + j0, F + (long long)(k1 + j0 + sw) * ld,
                                    ld, nbp - (j0 + sw));
            __syncthreads();
        }
        STAMP(5);
    }
    __syncthreads();
    if (tid < ncols_done) {                            // HStair / HTau / Rdead of this panel, one coalesced flush
        St[k1 + tid] = ps.st_out[tid];
        Tau[k1 + tid] = s_tau[tid];
        if (ps.dead[tid]) Rdead[k1 + tid] = 1;
    }
    STAMP(3);
    // ---- T of the whole panel for the trailing update ----
    if (!(dbg & 2)) dev_gram_T<NTH>(F + (long long)k1 * ld, ld, g1, tlast, nbp, s_diag, s_tau, s_G, s_T, Tout, lds);
    if (Tkeep)                                           // the same T, kept for the Q-apply on the resident factors
        for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
            const int a = e % STM_NB, b = e / STM_NB;
            Tkeep[e] = (a < nbp && b < nbp && a <= b) ? s_T[a][b] : 0.0;
        }
    STAMP(4);
#ifdef STMMQR_STAMPS
    if ((dbg & 16) && tid == 0 && dbgbuf)
        for (int e = 0; e < 12; e++) atomicAdd(&dbgbuf[e], tph[e]);
#endif
#undef STAMP
#undef CSTAMP
    PanelDesc *pd = &num->pd[STM_PDI(p)];
    if (tid < STM_NB) pd->pdiag[tid] = (tid < nbp) ? s_diag[tid] : STM_BIGROW;
    if (tid == 0) {
        num->g = g; num->rank = rank; num->done = done;
        pd->pg1 = g1; pd->pt = tlast; pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2;
        num->flops += flops;
        num->flops_upd += 4.0 * (double)(n - k2) * lensum;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Tall-panel pipeline (panels with more rows than one workgroup can stream through LDS cheaply).
//
// The panel's columns are cut into sub-panels ("groups") of 8 columns (4 above STM_TALL_WIDE rows) and ONE launch runs
// one workgroup per group, blockIdx.y = b:
//     * the group's rows [g1, tmax) x 8 columns are loaded ONCE into registers (thread tid owns rows g1 + tid + NTH*r),
//     * for s = 0 .. b-1: wait until group s has been factorized (FrontNum::prog, release/acquire at agent scope), then
//       apply its reflectors, read back from F with the unit-diagonal mask -- dlarf semantics, reflector after
//       reflector, each v'C one 8-value workgroup reduction (reference qr_private_apply1, :1359-1381),
//     * factorize the group in the same registers (the column step of dev_subpanel_reg), store every finished column
//       straight to F, publish prog.
// A group only ever waits for groups with a smaller blockIdx.y of the same front, i.e. for workgroups that were
// dispatched before it, so the wait cannot deadlock whatever the residency; the spin is bounded all the same.
// Compared with one workgroup per panel this keeps every column in registers for its whole life inside the panel
// (one read, one write of F per column), spreads the in-panel dlarf over up to eight CUs, overlaps the loads of the
// later groups with the factorization of the earlier ones, and makes the cost of a column step independent of the
// panel height (<= STM_TALL_MAX rows).  The last group also builds T of the whole panel (dev_gram_T) and the
// block-reflector description for the trailing update.
// ------------------------------------------------------------------------------------------------
// (lds_barrier, ld_agent / st_agent: stmmqr_devutil.h)

// whole workgroup: wait until *flag >= target (written by another workgroup of this launch), then acquire.
// Returns false if the bounded spin ran out (never expected; the caller gives up on the panel).
// `seen` (thread 0): a value of the flag loaded earlier (the poll of a consumer that is behind its producer costs a
// memory round trip although the flag has long been set: it loads the flag before its previous block of work instead).
__device__ __forceinline__ bool wait_progress(const int *flag, int target, int seen = -1)
{
    __shared__ int s_ok;
    __syncthreads();                               // (s_ok of a previous wait has been read by everyone)
    if (threadIdx.x == 0) {
        int ok = (seen >= target);
        for (int it = 0; !ok && it < (1 << 26); it++) {
            if (ld_agent(flag) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        s_ok = ok;
        // ONE lane acquires for the whole CU (the invalidate acts on the CU's L1 and on stale other-XCD lines; every
        // wave fencing costs 2-4x as much); the wait holds the barrier until the invalidate has completed.
        // (Dropping the acquire in favour of sc1 loads of the handed-over columns was measured: no gain.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return s_ok != 0;
}
// publish: all global stores of this workgroup happen-before the flag value.
// Everything handed from one workgroup of the panel launch to another is stored WRITE-THROUGH (st_agent: sc1 stores, the
// panel columns included), so no L2 write-back (agent-scope release: buffer_wbl2 writes back every dirty line of the XCD's
// L2, microseconds when other fronts' updates have just run there) is needed: every wave waits for its own stores to be
// acknowledged (a workgroup barrier alone does not wait for them), the barrier joins the waves, one lane stores the flag.
__device__ __forceinline__ void publish_progress(int *flag, int value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_agent(flag, value);
}

template <int NTH>
__device__ __forceinline__ void block_reduce8(PanelShared &ps, int &par, const double (&part)[8], double (&sum)[8])
{
    constexpr int NWV = NTH / 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const double rw = wave_reduce8(part);
    if (lane < 8) ps.rsum[par][wid * 8 + lane] = rw;
    lds_barrier();
    const double rb = wave_sum_stride8((lane < NWV * 8) ? ps.rsum[par][lane] : 0.0);
    sum[0] = lane_bcast<red8_lane(0)>(rb); sum[1] = lane_bcast<red8_lane(1)>(rb);
    sum[2] = lane_bcast<red8_lane(2)>(rb); sum[3] = lane_bcast<red8_lane(3)>(rb);
    sum[4] = lane_bcast<red8_lane(4)>(rb); sum[5] = lane_bcast<red8_lane(5)>(rb);
    sum[6] = lane_bcast<red8_lane(6)>(rb); sum[7] = lane_bcast<red8_lane(7)>(rb);
    par ^= 1;
}

template <int NTH, int RPT, int SWT>
__device__ __forceinline__ void dev_tall_group(PanelShared &ps, const FrontSym &s, FrontNum *num, PanelDesc *pd, double *F,
                                               int *St, double *Tau, char *Rdead, int p, int b, int g1, int tmax,
                                               double tol, int ntol_global, double *Tout, double *lds, int dbg = 0,
                                               unsigned long long *dbgbuf = nullptr, double *Tkeep = nullptr, int defer_ok = 0,
                                               const double *sigp = nullptr, int *abortp = nullptr)
{
    const int tid = threadIdx.x;
    const int m = num->fm, n = s.fn, npiv = s.fp;              // (fm is fixed before the panel kernels run)
    const long long ld = s.ld;
    const int k1 = p * STM_NB, k2 = min(n, k1 + STM_NB), nbp = k2 - k1;
    const int c0 = SWT * b, sw = min(SWT, nbp - c0);           // my columns: k1 + c0 + x, x < sw
    const int ns = (nbp + SWT - 1) / SWT;
    const int rb = g1;                                         // first row of the register image
    int par = 0;
    if (tid < SWT) ps.stair[tid] = (tid < sw) ? St[k1 + c0 + tid] : 0;     // (only this group ever writes these)
#ifdef STMMQR_STAMPS
    unsigned long long ts0 = clock64(), ts1;
    const bool stamp_me = (dbg & 16) && dbgbuf && (b == ns - 1);
#define TSTAMP(idx) do { if (stamp_me) { __syncthreads(); ts1 = clock64(); if (tid == 0) atomicAdd(&dbgbuf[idx], ts1 - ts0); ts0 = ts1; } } while (0)
    // timeline of panel 1 (dbg & 32): wall clock (100 MHz) of thread 0 at the events of every group, dbgbuf[16 + 64 b + idx]
    const bool tl_on = (dbg & 32) && dbgbuf && p == (((dbg >> 24) & 127) ? ((dbg >> 24) & 127) : 1) && tid == 0 && s.parent < 0;      // (the root front only; panel = dbg bits 24-30)
#define TL(idx) do { if (tl_on && ((idx) < 8 || (idx) >= 20) && !((idx) >= 26 && (idx) < 30)) dbgbuf[16 + 64 * b + (idx)] = wall_clock64(); } while (0)
#define TLW(idx) do { if ((dbg & 32) && s.parent < 0 && p == (((dbg >> 24) & 127) ? ((dbg >> 24) & 127) : 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TL(idx); } while (0)
#define TCY(idx) do { if (tl_on && j == 4) dbgbuf[16 + 64 * b + 48 + (idx)] = clock64(); } while (0)
#else
#define TCY(idx) do { } while (0)
#define TSTAMP(idx) do { } while (0)
#define TL(idx) do { } while (0)
#define TLW(idx) do { } while (0)
#endif
    TL(0);

    // (loads are unconditional on a clamped index and masked afterwards: a predicated load becomes a branch around
    //  each access and the 8 x RPT loads would be issued one round trip at a time)
    double a[RPT][SWT];
#pragma unroll
    for (int x = 0; x < SWT; x++) {
        const double *src = F + (long long)(k1 + c0 + min(x, sw - 1)) * ld;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int i = rb + tid + NTH * r;
            const double val = src[min(i, tmax - 1)];
            a[r][x] = (x < sw && i < tmax) ? val : 0.0;
        }
    }
    TSTAMP(6);
    TLW(1);
#ifdef STMMQR_STAMPS
    int tl_h = 0;
#endif
    // ---- apply the reflectors of the groups before mine, as they become available ----
    int prev_done = 0;
    // (8-column groups publish in QUARTERS since round 5: the part on the chain -- the last one of the group before mine -- is two
    //  reflectors instead of four; halves: default workload 92.4 ms, quarters 91.3, single columns 93.8 -- every publish costs the
    //  producer a wait for its stores and a barrier.  -DSTM_TALL_NPART=2 / 8 for the other two)
#ifndef STM_TALL_NPART
#define STM_TALL_NPART 4
#endif
#ifndef STM_TALL_NPART4
#define STM_TALL_NPART4 2
#endif
    constexpr int NPART = (SWT == 8) ? STM_TALL_NPART : (SWT == 4) ? STM_TALL_NPART4 : 2;      // a group publishes its columns in this many parts
    constexpr int HW = SWT / NPART;                            // reflectors per published part
    constexpr int NPV = HW * SWT, NGP = (NPV + 7) / 8;         // V'C products, in exchange groups of eight
    // (a group publishes in NPART parts -- after every HW of its columns -- so that the next group applies the earlier reflectors
    //  while the later ones are still being factorized)
    int seen = -1;
    bool have_chain = false;
    int ch_g = 0, ch_rank = 0, ch_pt = 0, ch_nl = 0;
    double ch_ls = 0, ch_fl = 0;
    for (int sp = 0; sp < b && !prev_done; sp++)
    for (int half = 0; half < NPART; half++) {
        if (!wait_progress(&num->prog, STM_PROG * p + NPART * sp + 1 + half, seen)) {
            if (tid == 0) { st_agent(&num->perr, 1); if (abortp) st_agent(abortp + 1, 1); }      // (= STM_SET_PERR)
            return;
        }
        TSTAMP(7);
#ifdef STMMQR_STAMPS
        tl_h = (sp == b - 1 && half >= NPART - 2) ? half - (NPART - 2) : 8;                       // (only the last two part applications: the ones on the chain)
#endif
        TL(2 + 3 * tl_h);
        const int pc0 = SWT * sp + half * HW;
        // (group sp ran out of rows?  Not num->done: a group that starts late would see the flag of a LATER group and
        //  skip the reflectors of the groups in between)
        if (half == NPART - 1) prev_done = (ld_agent(&pd->done_group) == sp);
        if (half == NPART - 1 && sp == b - 1) {
            // the scalars that travel along the chain of groups were stored before this flag: load them now, the
            // round trip hides behind the block application below (after the loop it would delay my first column)
            ch_g = ld_agent(&num->g); ch_rank = ld_agent(&num->rank); ch_pt = ld_agent(&pd->pt);
            ch_nl = ld_agent(&pd->nlive); ch_ls = ld_agent(&pd->lensum); ch_fl = ld_agent(&num->flops);
            have_chain = true;
        }
        // The HW reflectors of this half are applied as ONE block reflector, C -= V T' (V'C): the HW x SWT products V'C
        // and the strict upper triangle of V'V go through a single workgroup exchange (one barrier instead of one per
        // reflector), T is the HW x HW dlarft recurrence done redundantly by every thread.  tau == 0 / dead columns enter
        // as v = 0, tau = 0.  (The blocking is fixed -- half groups -- so the rounding does not depend on the timing.)
        double v[HW][RPT], tq[HW];
#pragma unroll
        for (int q = 0; q < HW; q++) {
            const int d = ld_agent(&pd->pdiag[pc0 + q]);
            const double tau = ld_agent(&Tau[k1 + pc0 + q]);
            const bool live = !(tau == 0.0 || d >= STM_BIGROW);
            tq[q] = live ? tau : 0.0;
            const double *vc = F + (long long)(k1 + pc0 + q) * ld;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                const double val = vc[min(i, tmax - 1)];       // (rows beyond a column's staircase are zero in F)
                v[q][r] = (!live || i < d || i >= tmax) ? 0.0 : ((i == d) ? 1.0 : val);
            }
        }
#ifdef STMMQR_STAMPS
        if (stamp_me) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(12);
        TLW(3 + 3 * tl_h);
#endif
        if (tid == 0) seen = ld_agent(&num->prog);             // for the next wait: in flight during the block below
        // exchange group 0: the strict upper triangle of V'V; groups 1..: V'C, eight products per group
        const int lane = tid & 63, wid = tid >> 6;
        if constexpr (HW > 1) {
            double gv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) gv[e] = 0.0;
#pragma unroll
            for (int q1 = 0; q1 < HW; q1++) {
#pragma unroll
                for (int q2 = q1 + 1; q2 < HW; q2++) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < RPT; r++) acc += v[q1][r] * v[q2][r];
                    gv[q1 * HW - (q1 * (q1 + 1)) / 2 + (q2 - q1 - 1)] = acc;
                }
            }
            const double rw = wave_reduce8(gv);
            if (lane < 8) ps.rsumB[0][wid * 8 + lane] = rw;
        }
#pragma unroll
        for (int gi = 0; gi < NGP; gi++) {
            double pv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int q = min((gi * 8 + e) / SWT, HW - 1), x = (gi * 8 + e) % SWT;
                double acc = 0.0;
                if (gi * 8 + e < NPV) {
#pragma unroll
                    for (int r = 0; r < RPT; r++) acc += v[q][r] * a[r][x];
                }
                pv[e] = acc;
            }
            const double rw = wave_reduce8(pv);
            if (lane < 8) ps.rsumB[1 + gi][wid * 8 + lane] = rw;
        }
        lds_barrier();
        // T (upper triangular, dlarft forward/columnwise): T(0:j,j) = -tau_j T(0:j,0:j) (V(:,0:j)' v_j)
        double Tq[HW][HW];
        if constexpr (HW == 1) Tq[0][0] = tq[0];
        else {
            const double rs = wave_sum_stride8((lane < (NTH / 64) * 8) ? ps.rsumB[0][lane] : 0.0);
            const double G8[8] = {lane_bcast<red8_lane(0)>(rs), lane_bcast<red8_lane(1)>(rs), lane_bcast<red8_lane(2)>(rs),
                                  lane_bcast<red8_lane(3)>(rs), lane_bcast<red8_lane(4)>(rs), lane_bcast<red8_lane(5)>(rs),
                                  lane_bcast<red8_lane(6)>(rs), lane_bcast<red8_lane(7)>(rs)};
#pragma unroll
            for (int j = 0; j < HW; j++) {
#pragma unroll
                for (int i = 0; i < HW; i++) Tq[i][j] = 0.0;
                Tq[j][j] = tq[j];
#pragma unroll
                for (int i = 0; i < j; i++) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = i; l < j; l++) acc += Tq[i][l] * G8[l * HW - (l * (l + 1)) / 2 + (j - l - 1)];
                    Tq[i][j] = -tq[j] * acc;
                }
            }
        }
        // W = T' (V'C), accumulated group by group (eight products live at a time), then C -= V W
        double Wq[HW][SWT];
#pragma unroll
        for (int q = 0; q < HW; q++) {
#pragma unroll
            for (int x = 0; x < SWT; x++) Wq[q][x] = 0.0;
        }
#pragma unroll
        for (int gi = 0; gi < NGP; gi++) {
            const double rs = wave_sum_stride8((lane < (NTH / 64) * 8) ? ps.rsumB[1 + gi][lane] : 0.0);
            const double P8[8] = {lane_bcast<red8_lane(0)>(rs), lane_bcast<red8_lane(1)>(rs), lane_bcast<red8_lane(2)>(rs),
                                  lane_bcast<red8_lane(3)>(rs), lane_bcast<red8_lane(4)>(rs), lane_bcast<red8_lane(5)>(rs),
                                  lane_bcast<red8_lane(6)>(rs), lane_bcast<red8_lane(7)>(rs)};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (gi * 8 + e < NPV) {
                    const int l = min((gi * 8 + e) / SWT, HW - 1), x = (gi * 8 + e) % SWT;
#pragma unroll
                    for (int q = l; q < HW; q++) Wq[q][x] += Tq[l][q] * P8[e];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; r++) {
#pragma unroll
            for (int q = 0; q < HW; q++) {
#pragma unroll
                for (int x = 0; x < SWT; x++) a[r][x] -= v[q][r] * Wq[q][x];
            }
        }
        TSTAMP(11);
        TL(4 + 3 * tl_h);
    }
    TSTAMP(8);
    // a group before mine ran out of rows (g reached fm): its reflectors were still due on my columns (applied above);
    // nothing is left to factorize.  The group right after it finalises the panel, the others only store.
    if (prev_done) {
#pragma unroll
        for (int x = 0; x < SWT; x++) {
            if (x < sw) {
                double *dst = F + (long long)(k1 + c0 + x) * ld;
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + tid + NTH * r;
                    if (i < tmax) dst[i] = a[r][x];
                }
            }
        }
        if (ld_agent(&pd->done_group) != b - 1) return;
    }
    // ---- factorize my sub-panel (column step as in dev_subpanel_reg; finished columns go straight to F) ----
    // g / rank / pt / nlive / flops travel along the chain of groups: group b reads them after the acquire on group b-1
    if (!have_chain) {
        ch_g = ld_agent(&num->g); ch_rank = ld_agent(&num->rank); ch_pt = ld_agent(&pd->pt);
        ch_nl = ld_agent(&pd->nlive); ch_ls = ld_agent(&pd->lensum); ch_fl = (b == 0) ? num->flops : ld_agent(&num->flops);
    }
    int g = (b == 0) ? g1 : ch_g;
    int rank = ch_rank, done = prev_done, nlive = 0;
    int tlast = (b == 0) ? g1 : ch_pt;
    const int nl_before = (b == 0) ? 0 : ch_nl;
    const double ls_before = (b == 0) ? 0.0 : ch_ls;
    long long iflops = 0, ilen = 0;                            // the reference's flop count: integers, exact in fp64
    const int gs = g;
    const int ntol = min(ntol_global - s.col1, npiv);
    const double sg = sigp ? sigp[0] : 1.0, isg = sigp ? sigp[1] : 1.0;      // magnitude guard (stm_larfg_guarded)
    // HStair / HTau / pdiag / Rdead of my columns [j0, j1): from LDS to global memory, write-through, before a publish
    auto flush_cols = [&](int j0, int j1) {
        lds_barrier();
        if (tid >= j0 && tid < j1) {
            const int kk = k1 + c0 + tid;
            st_agent(&St[kk], ps.st_out[tid]); st_agent(&Tau[kk], ps.tau[tid]); st_agent(&pd->pdiag[c0 + tid], ps.diag[tid]);
            if (ps.dead[tid]) st_agent(&Rdead[kk], (char)1);
        }
    };
    int flushed = 0, jdone = SWT;
    lds_barrier();                                             // ps.stair
    // (fully unrolled over the group's columns: column j works on the register columns j .. SWT-1 only -- no dot products or
    //  updates of columns that have already been retired, no rotation of the register image; same operations on the live
    //  columns in the same order, so the same bits as the rotating loop)
#pragma unroll
    for (int j = 0; j < SWT; j++) {
        if (j >= sw || prev_done) continue;
        const int jp = c0 + j, k = k1 + jp;
        if (j > 0 && j % HW == 0 && b + 1 < ns) {
            flush_cols(min(flushed, jdone), min(j, jdone)); flushed = j;
            publish_progress(&num->prog, STM_PROG * p + NPART * b + j / HW);   // the part before column j is in F
        }
        if (!done && g >= m) {
            // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
            for (int kk = k + tid; kk < n; kk += NTH) {
                if (kk < npiv) { st_agent(&Rdead[kk], (char)1); st_agent(&St[kk], 0); }
                else st_agent(&St[kk], m);
                st_agent(&Tau[kk], 0.0);
            }
            for (int jj = jp + tid; jj < STM_NB; jj += NTH) st_agent(&pd->pdiag[jj], STM_BIGROW);
            done = 1;
            jdone = j;
        }
        if (!done) {
            // Straight-line column step: the dead-column and tau == 0 cases are folded into the scalars (a branch
            // around the update would make the compiler copy the whole register image at the join).
            TCY(0);
            const int t = max(g + 1, ps.stair[j]);
            double part[8], sum[8];
#pragma unroll
            for (int x = 0; x < 8; x++) part[x] = 0;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                const double xv = (i > g && i < t) ? a[r][j] * sg : 0.0;   // (one operand carries the magnitude guard)
#pragma unroll
                for (int x = j; x < SWT; x++) part[x - j] += xv * a[r][x];
            }
            const bool owner = (tid == g - rb);                // holds row g in a[0][.]  (g - rb < STM_NB <= NTH)
            const int tpar = par;
            if (owner) {
#pragma unroll
                for (int x = j; x < SWT; x++) ps.top[tpar][x - j] = a[0][x];
            }
            TCY(1);
            block_reduce8<NTH>(ps, par, part, sum);
            TCY(2);
            const double alpha = ps.top[tpar][0];
            const double ss = sum[0];
            // dlarfg; ss == 0 (no active row below the diagonal, or all of them zero) gives H = I
            double bb, tau0, scal0, scals0;
            stm_larfg_guarded(alpha, ss, sg, isg, bb, tau0, scal0, scals0);
            const bool ident = (ss == 0.0);
            const double beta = ident ? alpha : bb;
            const bool dead = (k < ntol) && (fabs(beta) <= tol);    // (:1495-1544) column zeroed, g does not advance
            const bool upd = !ident && !dead;
            const double tau = upd ? tau0 : 0.0;
            const double scal = upd ? scal0 : 0.0, scals = upd ? scals0 : 0.0;
            TCY(3);
            double w[8];
#pragma unroll
            for (int x = 1; x < SWT - j; x++) w[x] = tau * (ps.top[tpar][x] + scals * sum[x]);    // 0 unless upd
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                const bool act = (i > g && i < t);
                const double v = act ? a[r][j] * scal : 0.0;    // (upd false: the entries are zero already, or dead)
                a[r][j] = act ? v : ((dead && i >= g) ? 0.0 : a[r][j]);
#pragma unroll
                for (int x = j + 1; x < SWT; x++) a[r][x] -= w[x - j] * v;
            }
            if (owner) {
#pragma unroll
                for (int x = j + 1; x < SWT; x++) a[0][x] -= w[x - j];
                a[0][j] = dead ? 0.0 : beta;
            }
            if (tid == 0) {          // (flushed to global memory before each publish: flush_cols below)
                ps.st_out[j] = dead ? 0 : t; ps.tau[j] = tau; ps.diag[j] = dead ? STM_BIGROW : g; ps.dead[j] = dead ? 1 : 0;
            }
            if (!dead) {
                iflops += (long long)(t - g) * (3 + 4 * (long long)(n - k - 1));
                ilen += (t - g);
                nlive += (tau != 0.0);
                tlast = t;
                g++;
            }
            if (k == npiv - 1) rank = g;                       // (:1604-1608) also taken on a dead last pivot
            TCY(4);
        }
        // ---- retire register column j to F ----
        {
            double *dst = F + (long long)k * ld;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                if (i < tmax) st_agent(&dst[i], a[r][j]);       // write-through: read by the other groups of this launch
            }
        }
        TCY(5);
        TL(20 + j);
    }
    TSTAMP(9);
    const double flops = (double)iflops, lensum = (double)ilen;
    if (!prev_done) flush_cols(min(flushed, jdone), min(sw, jdone));
    // ---- sub-panel bookkeeping; the last sub-panel (or the group after one that ran out of rows) finalises ----
    const bool last = prev_done || b == ns - 1;
    const int nl_total = nl_before + nlive;
    if (tid == 0 && !prev_done) {
        st_agent(&num->g, g); st_agent(&num->rank, rank); st_agent(&num->done, done);
        st_agent(&num->flops, ch_fl + flops);
        st_agent(&pd->sg[b], gs); st_agent(&pd->st[b], tlast); st_agent(&pd->pt, tlast); st_agent(&pd->nlive, nl_total);
        st_agent(&pd->lensum, ls_before + lensum);
        if (done) st_agent(&pd->done_group, b);
    }
    if (!last) {
        publish_progress(&num->prog, STM_PROG * p + NPART * b + NPART);
        TL(30);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the write-through column stores of every wave: acknowledged)
    __syncthreads();                                           // all stores of this workgroup are complete and visible
    // T of a panel with trailing columns is left to the trailing update: the row-parallel form builds it from a Gram
    // block per slab (k_upd_w; the last slab to arrive runs the recurrence), the one-workgroup form from the Gram matrix
    // it accumulates beside W (dev_update_block) -- same association, bit-identical T -- so the Gram pass over the
    // whole panel (one CU filling rows x 32 columns: 6-11 us) and the recurrence leave the critical path.  The rule
    // depends on this front alone (defer_ok: an update launch follows this panel launch; k2 < n: this front takes part
    // in it), so the results do not depend on which fronts share a level.
    const int defer_t = (defer_ok && k2 < n && tlast > g1) ? 1 : 0;   // (no live reflector: T = 0 is written here)
    if (tid == 0) {
        // (pd->mode stays 1: it belongs to the header, and a column group that starts late -- the finalising group is
        //  not always the last one -- must still find it there)
        pd->t_deferred = defer_t;
        pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2;
        num->flops_upd += 4.0 * (double)(n - k2) * (prev_done ? ls_before : ls_before + lensum);
    }
    if (defer_t) {
        publish_progress(&num->prog, STM_PROG * p + NPART * b + NPART);
        return;
    }
    if (tid < STM_NB) {
        ps.diag[tid] = (tid < nbp) ? ld_agent(&pd->pdiag[tid]) : STM_BIGROW;    // (columns past a `done` point were reset there)
        ps.tau[tid] = (tid < nbp) ? ld_agent(&Tau[k1 + tid]) : 0.0;
    }
    __syncthreads();
    dev_gram_T<NTH>(F + (long long)k1 * ld, ld, g1, tlast, nbp, ps.diag, ps.tau, ps.G, ps.T, Tout, lds
#ifdef STMMQR_STAMPS
                    , ((dbg & 32) && dbgbuf && p == 1) ? dbgbuf + 16 + 64 * b + 40 : nullptr
#endif
                    );
    if (Tkeep)
        for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
            const int ai = e % STM_NB, bi = e / STM_NB;
            Tkeep[e] = (ai < nbp && bi < nbp && ai <= bi) ? ps.T[ai][bi] : 0.0;
        }
    TSTAMP(10);
    TL(31);
    // (when an earlier group ran out of rows the groups after mine are still storing their columns: the kernel
    //  boundary orders those stores before the trailing update)
    publish_progress(&num->prog, STM_PROG * p + NPART * b + NPART);
#undef TSTAMP
#undef TL
#undef TLW
#undef TCY
}

// ------------------------------------------------------------------------------------------------
// Wave-pipelined panel (short panels: at most STM_WP_ROWS rows).  ONE workgroup, one WAVE per group of WP_SW columns,
// every wave holds all the rows of its columns in registers (lane l: rows rb + l + 64 r).  The column step of qr_front
// (SparseQR_factorize.c:1434-1609: dlarfg, dlarf on the rest of the group) then needs no workgroup barrier and no LDS
// round trip: the 8 sums of a step are one DPP reduction inside the wave.  Finished columns go to an image of the panel
// in LDS; the waves after the owner apply them (dlarf, one wave reduction each) as they appear -- the hand-off is an LDS
// flag per column (release / acquire at workgroup scope), polled by the consumer.  A wave only ever waits for waves
// before it, all of them resident in the same workgroup: no bounded waits, no global-memory flags.
// Which panels come here is decided from the front's own rows when the panel starts (k_panel, mode 2), so the results
// do not depend on the fronts that share the step or the device.
// ------------------------------------------------------------------------------------------------
#define WP_SW 4                          // columns per wave: 8 waves = the 512 threads of k_panel
struct WaveShared {
    int ready[STM_NB];                   // column j of the panel is in the LDS image, its scalars below are valid
    int d[STM_NB], t[STM_NB];            // unit-diagonal row of reflector j (STM_BIGROW: none) / one past its last row
    double tau[STM_NB];
    // the scalars that travel along the chain of columns: state AFTER the last finished column
    int g, rank, done, tlast, nlive, jdone;
    long long iflops, ilen;
};

// (SW: columns per wave -- 4 in k_panel's 512-thread workgroups, 8 in the 256-thread workgroups of k_front_wg)
template <int RPT, int SW = WP_SW>
__device__ __forceinline__ void dev_wave_panel(PanelShared &ps, WaveShared &wsh, const FrontSym &s, FrontNum *num, PanelDesc *pd,
                                               double *F, int *St, double *Tau, char *Rdead, int p, int g1, int tmax, double tol,
                                               int ntol_global, double *Tout, double *lds, double *Tkeep, int defer_ok,
                                               const double *sigp)
{
    constexpr int NTH = 64 * (STM_NB / SW), RS = 64 * RPT;
    static_assert(SW == 4 || SW == 8, "a wave takes 4 or 8 columns");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = num->fm, n = s.fn, npiv = s.fp;
    const long long ld = s.ld;
    const int k1 = p * STM_NB, k2 = min(n, k1 + STM_NB), nbp = k2 - k1;
    const int c0 = SW * w, sw = max(0, min(SW, nbp - c0));
    const int rb = g1;
    const double flops_before = num->flops;
    if (tid < STM_NB) {
        wsh.ready[tid] = 0; wsh.d[tid] = STM_BIGROW; wsh.t[tid] = 0; wsh.tau[tid] = 0.0;
        ps.stair[tid] = (tid < nbp) ? St[k1 + tid] : 0;
        ps.st_out[tid] = 0; ps.dead[tid] = 0;
    }
    if (tid == 0) {
        wsh.g = g1; wsh.rank = num->rank; wsh.done = 0; wsh.tlast = g1; wsh.nlive = 0; wsh.jdone = STM_NB;
        wsh.iflops = 0; wsh.ilen = 0;
    }
    double a[RPT][SW];
#pragma unroll
    for (int x = 0; x < SW; x++) {
        const double *src = F + (long long)(k1 + c0 + min(x, max(sw, 1) - 1)) * ld;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int i = rb + lane + 64 * r;
            const double val = src[min(i, tmax - 1)];
            a[r][x] = (x < sw && i < tmax) ? val : 0.0;
        }
    }
    __syncthreads();
    auto wait_col = [&](int j) {
        while (__hip_atomic_load(&wsh.ready[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    };
    const int ntol = min(ntol_global - s.col1, npiv);
    const double sg = sigp ? sigp[0] : 1.0, isg = sigp ? sigp[1] : 1.0;
    if (sw > 0) {
        // ---- the reflectors of the waves before mine, one dlarf each, as they appear ----
        for (int j = 0; j < c0; j++) {
            wait_col(j);
            const double tau = wsh.tau[j];
            const int d = wsh.d[j];
            if (tau == 0.0 || d >= STM_BIGROW) continue;
            const double *vs = lds + j * RS;
            double v[RPT], pv[8];
#pragma unroll
            for (int x = 0; x < 8; x++) pv[x] = 0.0;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + lane + 64 * r;
                const double val = vs[lane + 64 * r];           // (rows beyond a column's staircase are zero)
                v[r] = (i < d) ? 0.0 : ((i == d) ? 1.0 : val);
#pragma unroll
                for (int x = 0; x < SW; x++) pv[x] += v[r] * a[r][x];
            }
            const double rw = wave_reduce8(pv);
            double wv[SW];
            wv[0] = tau * lane_bcast<red8_lane(0)>(rw); wv[1] = tau * lane_bcast<red8_lane(1)>(rw);
            wv[2] = tau * lane_bcast<red8_lane(2)>(rw); wv[3] = tau * lane_bcast<red8_lane(3)>(rw);
            if constexpr (SW == 8) {
                wv[4] = tau * lane_bcast<red8_lane(4)>(rw); wv[5] = tau * lane_bcast<red8_lane(5)>(rw);
                wv[6] = tau * lane_bcast<red8_lane(6)>(rw); wv[7] = tau * lane_bcast<red8_lane(7)>(rw);
            }
#pragma unroll
            for (int r = 0; r < RPT; r++) {
#pragma unroll
                for (int x = 0; x < SW; x++) a[r][x] -= wv[x] * v[r];
            }
        }
        if (c0 > 0) wait_col(c0 - 1);                            // (a skipped reflector above was still waited for)
        int g = wsh.g, rank = wsh.rank, done = wsh.done, tlast = wsh.tlast, nlive = wsh.nlive, jdone = wsh.jdone;
        long long iflops = wsh.iflops, ilen = wsh.ilen;
        // ---- my columns ----
#pragma unroll
        for (int x = 0; x < SW; x++) {
            if (x >= sw) continue;
            const int jp = c0 + x, k = k1 + jp;
            if (!done && g >= m) {
                // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
                for (int kk = k + lane; kk < n; kk += 64) {
                    if (kk < npiv) { Rdead[kk] = (char)1; St[kk] = 0; }
                    else St[kk] = m;
                    Tau[kk] = 0.0;
                }
                for (int jj = jp + lane; jj < STM_NB; jj += 64) pd->pdiag[jj] = STM_BIGROW;
                done = 1;
                jdone = jp;
            }
            if (!done) {
                const int t = max(g + 1, ps.stair[jp]);
                double part[8];
#pragma unroll
                for (int e = 0; e < 8; e++) part[e] = 0.0;
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + lane + 64 * r;
                    const double xv = (i > g && i < t) ? a[r][x] * sg : 0.0;   // (one operand carries the magnitude guard)
#pragma unroll
                    for (int y = x; y < SW; y++) part[y - x] += xv * a[r][y];
                }
                const double rw = wave_reduce8(part);
                const double sum0 = lane_bcast<red8_lane(0)>(rw), sum1 = lane_bcast<red8_lane(1)>(rw);
                const double sum2 = lane_bcast<red8_lane(2)>(rw), sum3 = lane_bcast<red8_lane(3)>(rw);
                double sum[SW] = {sum0, sum1, sum2, sum3};
                if constexpr (SW == 8) {
                    sum[4] = lane_bcast<red8_lane(4)>(rw); sum[5] = lane_bcast<red8_lane(5)>(rw);
                    sum[6] = lane_bcast<red8_lane(6)>(rw); sum[7] = lane_bcast<red8_lane(7)>(rw);
                }
                const int ol = g - rb;                            // lane that holds row g in a[0][.]  (g - rb < STM_NB)
                double top[SW];
#pragma unroll
                for (int y = x; y < SW; y++)
                    top[y - x] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[0][y]), ol),
                                                  __builtin_amdgcn_readlane(__double2loint(a[0][y]), ol));
                const double alpha = top[0], ss = sum[0];
                double bb, tau0, scal0, scals0;
                stm_larfg_guarded(alpha, ss, sg, isg, bb, tau0, scal0, scals0);
                const bool ident = (ss == 0.0);
                const double beta = ident ? alpha : bb;
                const bool dead = (k < ntol) && (fabs(beta) <= tol);    // (:1495-1544) column zeroed, g does not advance
                const bool upd = !ident && !dead;
                const double tau = upd ? tau0 : 0.0;
                const double scal = upd ? scal0 : 0.0, scals = upd ? scals0 : 0.0;
                double wv[SW];
#pragma unroll
                for (int y = 1; y < SW - x; y++) wv[y] = tau * (top[y] + scals * sum[y]);     // 0 unless upd
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + lane + 64 * r;
                    const bool act = (i > g && i < t);
                    const double v = act ? a[r][x] * scal : 0.0;
                    a[r][x] = act ? v : ((dead && i >= g) ? 0.0 : a[r][x]);
#pragma unroll
                    for (int y = x + 1; y < SW; y++) a[r][y] -= wv[y - x] * v;
                }
                if (lane == ol) {
#pragma unroll
                    for (int y = x + 1; y < SW; y++) a[0][y] -= wv[y - x];
                    a[0][x] = dead ? 0.0 : beta;
                }
                if (lane == 0) {
                    ps.st_out[jp] = dead ? 0 : t; ps.tau[jp] = tau; ps.diag[jp] = dead ? STM_BIGROW : g; ps.dead[jp] = dead ? 1 : 0;
                    wsh.tau[jp] = tau; wsh.d[jp] = dead ? STM_BIGROW : g; wsh.t[jp] = t;
                }
                if (!dead) {
                    iflops += (long long)(t - g) * (3 + 4 * (long long)(n - k - 1));
                    ilen += (t - g);
                    nlive += (tau != 0.0);
                    tlast = t;
                    g++;
                }
                if (k == npiv - 1) rank = g;                       // (:1604-1608) also taken on a dead last pivot
            }
            // ---- the column goes to F and, for the waves after mine, to the LDS image ----
            {
                double *dst = F + (long long)k * ld;
                double *vs = lds + jp * RS;
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + lane + 64 * r;
                    if (i < tmax) dst[i] = a[r][x];
                    vs[lane + 64 * r] = a[r][x];
                }
            }
            if (x == sw - 1 || done) {
                // hand the chain to the next wave (after my last column; at once when the rows ran out: the columns after this
                // one carry no reflector, their owners only store)
                if (lane == 0) {
                    wsh.g = g; wsh.rank = rank; wsh.done = done; wsh.tlast = tlast; wsh.nlive = nlive; wsh.jdone = jdone;
                    wsh.iflops = iflops; wsh.ilen = ilen;
                }
            }
            if (lane == 0) {
                __hip_atomic_store(&wsh.ready[jp], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    // ---- bookkeeping of the panel (what the last column group of the pipeline does) ----
    const int g = wsh.g, tlast = wsh.tlast, jdone = wsh.jdone, done = wsh.done;
    const double lensum = (double)wsh.ilen;
    if (tid < min(nbp, jdone)) {
        const int kk = k1 + tid;
        St[kk] = ps.st_out[tid]; Tau[kk] = ps.tau[tid]; pd->pdiag[tid] = ps.diag[tid];
        if (ps.dead[tid]) Rdead[kk] = (char)1;
    }
    const int defer_t = (defer_ok && k2 < n && tlast > g1) ? 1 : 0;   // (as dev_tall_group: T of a panel with trailing columns
                                                                      //  is built by the update that follows)
    if (tid == 0) {
        num->g = g; num->rank = wsh.rank; num->done = done;
        num->flops = flops_before + (double)wsh.iflops;
        pd->sg[0] = g1; pd->st[0] = tlast; pd->pt = tlast; pd->nlive = wsh.nlive; pd->lensum = lensum;
        pd->done_group = done ? 0 : -1;
        pd->t_deferred = defer_t;
        pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2;
        num->flops_upd += 4.0 * (double)(n - k2) * lensum;
    }
    if (defer_t) return;
    __syncthreads();                                               // pdiag / Tau / F of this workgroup: visible to all its threads
    if (tid < STM_NB) {
        ps.diag[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;  // (columns past a `done` point were reset there)
        ps.tau[tid] = (tid < nbp) ? Tau[k1 + tid] : 0.0;
    }
    __syncthreads();
    dev_gram_T<NTH>(F + (long long)k1 * ld, ld, g1, tlast, nbp, ps.diag, ps.tau, ps.G, ps.T, Tout, lds);
    if (Tkeep)
        for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
            const int ai = e % STM_NB, bi = e / STM_NB;
            Tkeep[e] = (ai < nbp && bi < nbp && ai <= bi) ? ps.T[ai][bi] : 0.0;
        }
}

// ------------------------------------------------------------------------------------------------
// small fronts: one workgroup runs the whole front (all panels, all trailing updates, C pack)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_front_wg(DevCtx c, const int *__restrict__ flist, int lds_doubles)
{
    extern __shared__ double dyn_lds[];
    __shared__ double s_Tw[STM_NB * STM_NB];
    __shared__ PanelShared ps;
    __shared__ WaveShared wsh;
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    FrontNum *num = &c.fnum[f];
    double *F = c.Farena + s.foff;
    auto Tkeep = [&](int p) { return c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : (double *)nullptr; };
    for (int p = 0; p < s.npanels; p++) {
        // A panel of at most 128 rows -- every panel of the fronts of a sparse matrix's lower tree levels: epb1's 825 small fronts have at
        // most 38 rows -- goes to the wave-pipelined panel (dev_wave_panel: a wave per 8 columns, the rows along the lanes, no workgroup
        // barrier and no LDS round trip per column) instead of the sub-panel code: k_front_wg 138 -> ... us per tree level on epb1.  The
        // condition is the front's own (its rows when the panel starts): results do not depend on the launch.
        {
            const int k2w = min(s.fn, (p + 1) * STM_NB), g1w = num->g;
            const int tmaxw = min(num->fm, max((c.Stair + s.rp)[k2w - 1], g1w + (k2w - p * STM_NB)));
            const int rowsw = tmaxw - g1w;
            if (!num->done && rowsw <= 128 && !(c.dbg & (64 | 16384)) && max(STM_NB * (rowsw <= 64 ? 64 : 128), 3072) <= lds_doubles) {
                PanelDesc *pdw = &num->pd[STM_PDI(p)];
                __syncthreads();                                   // (everyone has read FrontNum / Stair before the panel writes them)
                if (threadIdx.x == 0) { pdw->mode = 2; pdw->pg1 = g1w; pdw->tmax = tmaxw; }
#define WAVE_ARGS ps, wsh, s, num, pdw, F, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, g1w, tmaxw, c.tol, c.ntol, s_Tw, dyn_lds, Tkeep(p), 0, c.sig
                if (rowsw <= 64) dev_wave_panel<1, 8>(WAVE_ARGS);
                else dev_wave_panel<2, 8>(WAVE_ARGS);
#undef WAVE_ARGS
                __syncthreads();
                const int k2 = min(s.fn, (p + 1) * STM_NB);
                const int ncb = (s.fn - k2 + BN - 1) / BN;
                for (int cb = 0; cb < ncb; cb++)
                    dev_update_block(F, s.ld, pdw->pg1, pdw->pt - pdw->pg1, pdw->pk1, pdw->pnb, pdw->pdiag, s_Tw, k2 + cb * BN,
                                     min(BN, s.fn - (k2 + cb * BN)), dyn_lds);
                __syncthreads();
                continue;
            }
        }
        if ((c.dbg & 64) || panel_rows(s, num, c.Stair + s.rp, p) > lds_doubles - 65)
            dev_panel<NT, true>(ps, s, num, F, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, s_Tw,
                                dyn_lds, lds_doubles, c.dbg, nullptr, Tkeep(p), c.sig);
        else
            dev_panel<NT, false>(ps, s, num, F, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, s_Tw,
                                 dyn_lds, lds_doubles, c.dbg, nullptr, Tkeep(p), c.sig);
        const int k2 = min(s.fn, (p + 1) * STM_NB);
        const int ncb = (s.fn - k2 + BN - 1) / BN;
        const PanelDesc *pd = &num->pd[STM_PDI(p)];
        for (int cb = 0; cb < ncb; cb++)
            dev_update_block(F, s.ld, pd->pg1, pd->pt - pd->pg1, pd->pk1, pd->pnb, pd->pdiag, s_Tw, k2 + cb * BN,
                             min(BN, s.fn - (k2 + cb * BN)), dyn_lds);
        __syncthreads();
    }
    dev_cpack(c, s, num, 0, 1);
}

#define NTP 512               // threads of the large-front panel kernel (8 waves, <= 256 VGPRs each)

// ------------------------------------------------------------------------------------------------
// large fronts: panel and trailing update are separate launches (many workgroups per update)
// ------------------------------------------------------------------------------------------------
// the body of k_panel: column group b of panel p of front f (fi: the front's place in the launch's lists)
__device__ __forceinline__ void dev_k_panel(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int fi, int b,
                                            int nsub, int defer_ok, int lds_doubles, PanelShared &ps, WaveShared &wsh, double *dyn_lds)
{
    const int f = flist[fi];
    const int p = plist[fi];                                   // every front of a step is at its own panel
    __builtin_amdgcn_s_setprio(3);                             // (critical path: ahead of the side stream's update waves)
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || stm_use_ca(s, p, c.panel_algo, c.ca_min_rows)) return;      // (the Gram-based panel kernel takes those)
    FrontNum *num = &c.fnum[f];
    double *F = c.Farena + s.foff;
    double *T = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    int *St = c.Stair + s.rp;
    double *Tkeep = c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr;
    PanelDesc *pd = &num->pd[STM_PDI(p)];
    if ((c.dbg & 2048) && b == ((c.dbg >> 20) & 7)) {          // tests: column group (dbg >> 20) & 7 starts ~1 ms late
        for (int it = 0; it < 4000; it++) __builtin_amdgcn_s_sleep(100);
    }
    if ((c.dbg & 4096) && b > 0) {                              // tests: every column group but the first gives up at once
        if (threadIdx.x == 0) STM_SET_PERR(c, num);
        return;
    }
    bool tall = stm_tall_panel(s, p, c.tall_min) && !(c.dbg & 256);
    if (!tall && b > 0) return;
    if (tall) {
        const int k1 = p * STM_NB, k2 = min(s.fn, k1 + STM_NB), nbp = k2 - k1;
        int mode, g1, tmax, w;
        if (b == 0) {
            // group 0 decides the mode and publishes the panel-wide constants (the header) to the other groups
            const int was_done = num->done;
            g1 = num->g;
            tmax = min(num->fm, max(St[k2 - 1], g1 + nbp));
            w = (tmax - g1 > STM_TALL_XWIDE) ? 2 : (tmax - g1 > STM_TALL_WIDE) ? 4 : STM_SW;
            // mode 0: nothing to do or the whole panel is done below by this workgroup (it does not fit the register
            // image, or needs more groups than were launched: more rows than the full-rank estimate)
            // (the groups the PLAN launches for this front -- not the launch's, which other fronts may have raised)
            mode = (was_done || tmax - g1 > STM_TALL_MAX || (nbp + w - 1) / w > min(nsub, stm_tall_launches(s, p, c.tall_min))) ? 0 : 1;
            // mode 2: a short panel -- this workgroup alone, a wave per 4 columns (dev_wave_panel).  The condition belongs to
            // the front: the rows its staircase reaches now (in a sparse front the first panels are far shorter than the front:
            // 89 % of the pipeline panels of the xenon1 stand-in have at most 512 rows).  Measured and dropped: panels of up
            // to 1024 rows with 16 rows per lane and a 16-column ring image in LDS (default workload 120.6 -> 154 ms).
            // (the LDS test never fails for a launch sized by the host's rule, STM_NB * STM_WP_ROWS doubles for every launch with
            //  a pipeline panel: it keeps a smaller launch safe)
            if (!was_done && tmax - g1 <= STM_WP_ROWS && !(c.dbg & 16384) &&
                STM_NB * ((tmax - g1 <= 128) ? 128 : (tmax - g1 <= 256) ? 256 : 512) <= lds_doubles)
                mode = 2;
            if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0 && !was_done) {       // diagnosis: panels by their actual rows
                const int rws = tmax - g1;
                atomicAdd(&c.dbgbuf[32 + (rws <= 128 ? 0 : rws <= 256 ? 1 : rws <= 512 ? 2 : rws <= 1024 ? 3 : rws <= 2048 ? 4 : rws <= 4096 ? 5 : 6)], 1ull);
                if (rws <= 512 && stm_panel_rows_est(s, p) > STM_WP_ROWS) atomicAdd(&c.dbgbuf[39], 1ull);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                st_agent(&pd->mode, mode); st_agent(&pd->pg1, g1); st_agent(&pd->pt, g1); st_agent(&pd->tmax, tmax);
                st_agent(&pd->nlive, 0); st_agent(&pd->sw, w); st_agent(&pd->done_group, -1);
                if (was_done) { pd->pnb = 0; pd->t_deferred = 0; }
            }
            publish_progress(&num->hdr, p + 1);
            if (was_done) return;
        } else {
            if (!wait_progress(&num->hdr, p + 1)) { if (threadIdx.x == 0) STM_SET_PERR(c, num); return; }
            mode = ld_agent(&pd->mode); g1 = ld_agent(&pd->pg1); tmax = ld_agent(&pd->tmax); w = ld_agent(&pd->sw);
            if (mode != 1 || b * w >= nbp) return;
        }
        if (mode == 2) {
            const int rows = tmax - g1;
#define WAVE_ARGS ps, wsh, s, num, pd, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, g1, tmax, c.tol, c.ntol, T, dyn_lds, Tkeep, defer_ok, c.sig
            if (rows <= 128) dev_wave_panel<2>(WAVE_ARGS);
            else if (rows <= 256) dev_wave_panel<4>(WAVE_ARGS);
            else dev_wave_panel<8>(WAVE_ARGS);
#undef WAVE_ARGS
            return;
        }
        if (mode == 1) {
            const int rows = tmax - g1;
#define TALL_ARGS ps, s, num, pd, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, b, g1, tmax, c.tol, c.ntol, T, dyn_lds, c.dbg, c.dbgbuf, Tkeep, defer_ok, c.sig, c.abort
            if (w == 2) dev_tall_group<NTP, 16, 2>(TALL_ARGS);
            else if (w == 4) dev_tall_group<NTP, 8, 4>(TALL_ARGS);
            else if (rows <= NTP) dev_tall_group<NTP, 1, 8>(TALL_ARGS);
            else if (rows <= 2 * NTP) dev_tall_group<NTP, 2, 8>(TALL_ARGS);
            else dev_tall_group<NTP, 4, 8>(TALL_ARGS);
#undef TALL_ARGS
            return;
        }
    }
    // One workgroup does the whole panel.  The LDS it stages sub-panels in is a property of the FRONT (stm_front_lds: what
    // the front would get alone), never of the launch -- the sub-panel width, and with it the rounding, must not depend on
    // the fronts that share the step.  The host sizes the launch for the fronts PLANNED to come here; a panel that was
    // planned for the pipeline and fell back (more rows than the full-rank estimate, recovery of a timed-out wait) works
    // in place when the launch is smaller than that.
    const int lds_front = stm_front_lds(s);
    if ((c.dbg & 64) || lds_front > lds_doubles || panel_rows(s, num, St, p) > lds_front - 65)
        dev_panel<NTP, true>(ps, s, num, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, T, dyn_lds,
                             lds_front, c.dbg, c.dbgbuf, Tkeep, c.sig);
    else
        dev_panel<NTP, false>(ps, s, num, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, T, dyn_lds,
                              lds_front, c.dbg, c.dbgbuf, Tkeep, c.sig);
    if (threadIdx.x == 0) { pd->mode = 0; pd->t_deferred = 0; }
}

__global__ __launch_bounds__(NTP) void k_panel(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int nsub,
                                               int defer_ok, int lds_doubles)
{
    extern __shared__ double dyn_lds[];
    __shared__ PanelShared ps;
    __shared__ WaveShared wsh;
    const unsigned long long t0c = clock64(), t0w = wall_clock64();
    dev_k_panel(c, flist, plist, blockIdx.x, blockIdx.y, nsub, defer_ok, lds_doubles, ps, wsh, dyn_lds);
    if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0) {           // diagnosis: shader clock held during the panels (cycles / 100 MHz ticks)
        atomicAdd(&c.dbgbuf[44], clock64() - t0c); atomicAdd(&c.dbgbuf[45], wall_clock64() - t0w); atomicAdd(&c.dbgbuf[46], 1ull);
    }
}

// A launch: the panel pipeline's workgroups (blockIdx.z = 0: front blockIdx.x, column group blockIdx.y) and, behind them in dispatch
// order, k_upd_c's tiles of the previous step's update beyond block 0 (blockIdx.z - 1 = front of THAT step's lists, blockIdx.x =
// column block, blockIdx.y = slab).
__global__ __launch_bounds__(NTP) void k_panel_pc(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int npan,
                                                  int nsub, int defer_ok, int lds_doubles, const int *__restrict__ uflist,
                                                  const int *__restrict__ uplist, int ucb0, const double *Wp,
                                                  const long long *__restrict__ uwpoff, int rspw)
{
    extern __shared__ double dyn_lds[];
    __shared__ PanelShared ps;
    __shared__ WaveShared wsh;
    if (blockIdx.z == 0) {
        if ((int)blockIdx.x >= npan || (int)blockIdx.y >= nsub) return;
        const unsigned long long t0c = clock64(), t0w = wall_clock64();
        dev_k_panel(c, flist, plist, blockIdx.x, blockIdx.y, nsub, defer_ok, lds_doubles, ps, wsh, dyn_lds);
        if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0) {
            atomicAdd(&c.dbgbuf[44], clock64() - t0c); atomicAdd(&c.dbgbuf[45], wall_clock64() - t0w); atomicAdd(&c.dbgbuf[46], 1ull);
        }
        return;
    }
    dev_upd_c_h2(c, uflist, uplist, ucb0, Wp, uwpoff, (int)blockIdx.z - 1, blockIdx.x, blockIdx.y, rspw, dyn_lds, ps.stair);
}
// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
int stm_launch_front_wg(const DevCtx &c, const int *flist, int nfr, int lds_doubles, hipStream_t st)
{
    if (nfr <= 0) return 0;
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();
    hipLaunchKernelGGL(k_front_wg, dim3(nfr), dim3(NT), bytes, st, c, flist, (int)(bytes / sizeof(double)));
    return (int)hipGetLastError();
}
int stm_launch_panel(const DevCtx &c, const int *flist, const int *plist, int nfr, int nsub, int defer_ok, int lds_doubles, hipStream_t st)
{
    if (nfr <= 0) return 0;
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();   // in-panel MFMA update + Gram scratch
    // One workgroup per column group of the panel pipeline (blockIdx.y); fronts whose panel is not pipelined use group 0.
    // The column groups of a pipelined panel wait for each other, but only ever for groups with a smaller blockIdx.y of the
    // same front, i.e. for workgroups that the in-order dispatch has already started (the assumption of every
    // decoupled-look-back scan); the waits are bounded and a wait that runs out is recovered (stmmqr_factorize_device).
    // Oversubscribed launches (more workgroups than the GPU holds: groups start late) are exercised by the tests;
    // STMMQR_DBG bit 9 + STMMQR_CHUNK launch the fronts in chunks instead (tests).
    int K = nfr;
    if (nsub > 1 && (c.dbg & 512)) K = getenv("STMMQR_CHUNK") ? atoi(getenv("STMMQR_CHUNK")) : 1;
    if (K < 1) K = 1;
    for (int i = 0; i < nfr; i += K)
        hipLaunchKernelGGL(k_panel, dim3(nfr - i < K ? nfr - i : K, nsub), dim3(NTP), bytes, st, c, flist + i, plist + i, nsub, defer_ok,
                           (int)(bytes / sizeof(double)));
    return (int)hipGetLastError();
}
// Passenger launches (k_panel_pc / k_upd_fw above).  A: the panels of a step + k_upd_c of the column blocks ucb0 .. ucb0 + uncb - 1 of the
// fronts (uflist, uplist, uwpoff: the PREVIOUS step's lists) out of the passengers' workspace.
int stm_launch_panel_pc(const DevCtx &c, const int *flist, const int *plist, int nfr, int nsub, int defer_ok, int lds_doubles,
                        const int *uflist, const int *uplist, int unfr, int ucb0, int uncb, int umaxsl, const double *Wp,
                        const long long *uwpoff, hipStream_t st)
{
    if (unfr <= 0 || uncb <= 0 || umaxsl <= 0 || (c.dbg & 512)) {
        if (unfr > 0 && uncb > 0 && umaxsl > 0)
            (void)stm_launch_update_c(c, uflist, uplist, unfr, ucb0, uncb, umaxsl, Wp, uwpoff, st);
        return stm_launch_panel(c, flist, plist, nfr, nsub, defer_ok, lds_doubles, st);
    }
    if (nfr <= 0 && !(nfr < 0)) {
        return stm_launch_update_c(c, uflist, uplist, unfr, ucb0, uncb, umaxsl, Wp, uwpoff, st);
    }
    if (nfr < 0) nfr = 0;                                         // (measurements: the riders alone, in their rider form)
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();
    const int lds_arg = (int)(bytes / sizeof(double));            // what k_panel would be told: its rules must not see the passengers
    if (bytes < (size_t)STM_PC_LDS_DOUBLES * sizeof(double)) bytes = (size_t)STM_PC_LDS_DOUBLES * sizeof(double);
    // slabs per rider workgroup: a rider has its CU to itself (the launch carries the panel's registers and LDS), so its descriptor
    // chain and prologue (~5 us) overlap with nothing; per slab ~2.4 us.  Rounds of ~240 workgroups (urows: the launch's tiles, a bound)
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
    hipLaunchKernelGGL(k_panel_pc, dim3(nfr > uncb ? nfr : uncb, nsub > uy ? nsub : uy, 1 + unfr), dim3(NTP), bytes, st, c, flist,
                       plist, nfr, nsub, defer_ok, lds_arg, uflist, uplist, ucb0, Wp, uwpoff, rspw);
    return (int)hipGetLastError();
}
int stm_configure_update(void);
int stm_configure_sweep(void);
int stm_configure_resident(void);
int stm_configure_kernels(void)
{
    // allow the panel kernels to ask for up to 144 KiB of dynamic LDS (160 KiB per CU on gfx950)
    CK(hipFuncSetAttribute((const void *)k_front_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
    CK(hipFuncSetAttribute((const void *)k_panel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_panel_pc, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    { int e = stm_configure_update(); if (e) return e; }
    { int e = stm_configure_sweep(); if (e) return e; }
    return stm_configure_resident();
}
