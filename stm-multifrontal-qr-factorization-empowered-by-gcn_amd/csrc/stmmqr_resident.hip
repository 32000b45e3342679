// stmmqr_resident.hip -- SURVEY 8 (f1): Q-apply and triangular solves on the factors resident in HBM.  Shared device code: stmmqr_kdev.h.
#include "stmmqr_kdev.h"


// ================================================================================================
// Q-apply and triangular solve on the factors that are still resident in HBM
// (reference: qr_private_Happly / QR_qmult, STMMQR/src/qr/SparseQR.c:1455-1790; qr_rsolve :2218-2517).
// The Householder vectors are read in place from the front arena (unit diagonal of the q-th live reflector of a front
// at front row q, entries below it up to the column's HStair), rows of a front are global rows through Hii.
// Work vector W: one entry per row of A, indexed by the row id of S = A(P,:)  (Hii holds exactly these ids).
// ================================================================================================
#ifndef QA_NT
#define QA_NT 512
#endif
#define QA_NW (QA_NT / 64)
// inclusive scan of one int per thread over NWV waves; *total = sum.  s_scan: NWV ints.
template <int NWV>
__device__ __forceinline__ int qa_incl_scan(int v, int *s_scan, int *total)
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
    for (int w = 0; w < NWV; w++) {
        int sw = s_scan[w];
        if (w < wid) base += sw;
        tot += sw;
    }
    *total = tot;
    return x + base;
}
// live-reflector numbering of one front: dq[k] = number of live reflectors before column k, or -1 if column k has none
// (dead pivot column, or the rows ran out).  Mirrors the enumeration of the packed format (qr_rhpack :1691-1784).
__device__ void qa_number_reflectors(const FrontSym &s, int fm, const int *St, const double *Tau, int *dq, int *s_scan)
{
    const int tid = threadIdx.x;
    const int per = (s.fn + QA_NT - 1) / QA_NT;
    const int k0 = tid * per, k1 = min(s.fn, k0 + per);
    int cnt = 0;
    for (int k = k0; k < k1; k++) cnt += (k >= s.fp || St[k] != 0);
    int total;
    const int incl = qa_incl_scan<QA_NW>(cnt, s_scan, &total);
    int d = incl - cnt;
    for (int k = k0; k < k1; k++) {
        const bool live = (k >= s.fp || St[k] != 0);
        dq[k] = (live && d < fm && Tau[k] != 0.0) ? d : -1;     // (tau == 0: H = I, nothing to apply)
        d += live;
    }
    __syncthreads();
}

// method 0: x <- H_last ... H_1 x = Q'x (fronts leaves -> root, reflectors ascending); method 1: x <- Q x (reverse)
__global__ __launch_bounds__(QA_NT) void k_qapply(DevCtx c, const int *__restrict__ flist, int method, double *W, int *err)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[QA_NW];
    __shared__ double s_red[QA_NW];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    const int tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    const double *Tau = c.Tau + s.rp;
    const int *Hi = c.Hii + s.hip;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    double *xs = dyn_lds;                               // [fm]
    int *dq = (int *)(xs + ((fm + 1) & ~1));            // [fn]
    for (int i = tid; i < fm; i += QA_NT) xs[i] = W[Hi[i]];
    qa_number_reflectors(s, fm, St, Tau, dq, s_scan);
    // one reflector after the other: v'x by a workgroup reduction, then the rank-1 update of the LDS-resident x
    // (requesting the next column ahead of the reduction was measured and did not pay: 21.9 -> 26 ms on the stand-in)
    const int kbeg = method ? s.fn - 1 : 0, kend = method ? -1 : s.fn, kinc = method ? -1 : 1;
    for (int k = kbeg; k != kend; k += kinc) {
        const int d = dq[k];
        if (d < 0) continue;
        const double tau = Tau[k];
        const int h = min(d + 1, fm), t = St[k];
        const double *v = F + (long long)k * ld;
        double part = 0;
        for (int i = h + tid; i < t; i += QA_NT) part += v[i] * xs[i];
        double sdot = block_sum<QA_NT>(part, s_red);
        sdot = (sdot + xs[d]) * tau;
        __syncthreads();                                // everyone has read xs[d]
        for (int i = h + tid; i < t; i += QA_NT) xs[i] -= sdot * v[i];
        if (tid == 0) xs[d] -= sdot;
        __syncthreads();
    }
    for (int i = tid; i < fm; i += QA_NT) W[Hi[i]] = xs[i];
    (void)err;
}

// Blocked form of k_qapply with the T factors the factorization kept (DevCtx::Tall): per panel of <= 32 reflectors
//     w = V'x  (one sweep over the panel rows, 32 dot products per thread, ONE workgroup reduction of 32 values),
//     y = T'w  (Q'x)  or  T w  (Q x),      x -= V y  (second sweep).
// V is read in place with the unit-diagonal / staircase mask (reflector j of the panel: diagonal row dq[k], entries up
// to HStair[k]); identity and dead columns have zero rows/columns in T and are masked out of V.
// R right-hand sides of a batch per workgroup (template): every entry of V is read once per sweep for all of them; per vector the
// arithmetic and its order are those of one vector alone (the first sweep takes the panel in R passes of 32 / R reflectors so that the 32
// running sums stay in registers).
template <int R>
__global__ __launch_bounds__(QA_NT) void k_qapply_t(DevCtx c, const int *__restrict__ flist, int method, double *W, RhsBatch B, int nb)
{
    const int nr = min(R, nb - (int)blockIdx.y * R);       // right-hand sides of this workgroup (uniform)
    W += (long long)blockIdx.y * R * B.w;
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[QA_NW];
    __shared__ int s_d[STM_NB], s_t[STM_NB];
    __shared__ double s_part[R][QA_NW][STM_NB], s_w[R][STM_NB], s_y[R][STM_NB], s_T[STM_NB][STM_NB + 1];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0 || s.qbig) return;         // (qbig: k_qbig_* below)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *St = c.Stair + s.rp;
    const double *Tau = c.Tau + s.rp;
    const int *Hi = c.Hii + s.hip;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int xst = (fm + 1) & ~1;
    double *xs = dyn_lds;                               // [R][xst]
    int *dq = (int *)(xs + (size_t)R * xst);            // [fn]
    for (int i = tid; i < fm; i += QA_NT) {
        const int h = Hi[i];
#pragma unroll
        for (int r = 0; r < R; r++) xs[r * xst + i] = (r < nr) ? W[(long long)r * B.w + h] : 0.0;
    }
    qa_number_reflectors(s, fm, St, Tau, dq, s_scan);
    const int pbeg = method ? s.npanels - 1 : 0, pend = method ? -1 : s.npanels, pinc = method ? -1 : 1;
    for (int p = pbeg; p != pend; p += pinc) {
        const int k1 = p * STM_NB, nbp = min(STM_NB, s.fn - k1);
        if (tid < STM_NB) {
            const int d = (tid < nbp) ? dq[k1 + tid] : -1;
            s_d[tid] = (d >= 0) ? d : STM_BIGROW;
            s_t[tid] = (d >= 0) ? St[k1 + tid] : 0;
        }
        __syncthreads();
        int r0 = STM_BIGROW, r1 = 0;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) { r0 = min(r0, s_d[j]); r1 = max(r1, max(s_t[j], (s_d[j] < STM_BIGROW) ? s_d[j] + 1 : 0)); }
        if (r0 >= STM_BIGROW) { __syncthreads(); continue; }   // no live reflector in this panel (uniform)
        const double *Vp = F + (long long)k1 * ld;
        // T of this panel: requested now (coalesced), parked in LDS after the first sweep
        double treg[(STM_NB * STM_NB + QA_NT - 1) / QA_NT];
        {
            const double *T = c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB;
#pragma unroll
            for (int q = 0; q < (STM_NB * STM_NB + QA_NT - 1) / QA_NT; q++) treg[q] = T[min(tid + QA_NT * q, STM_NB * STM_NB - 1)];
        }
        // ---- w = V'x ----
        constexpr int HB = STM_NB / R;                     // reflectors per pass of the first sweep (R x HB running sums per thread)
#pragma unroll
        for (int h0 = 0; h0 < STM_NB; h0 += HB) {
            double acc[R][HB];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int j = 0; j < HB; j++) acc[r][j] = 0;
            for (int i = r0 + tid; i < r1; i += QA_NT) {
                double xi[R];
#pragma unroll
                for (int r = 0; r < R; r++) xi[r] = xs[r * xst + i];
#pragma unroll
                for (int j = 0; j < HB; j++) {
                    const double val = Vp[i + (long long)min(h0 + j, nbp - 1) * ld];        // unconditional, masked below
                    const double v = (i > s_d[h0 + j] && i < s_t[h0 + j]) ? val : ((i == s_d[h0 + j]) ? 1.0 : 0.0);
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r][j] += v * xi[r];
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                double part[8];
#pragma unroll
                for (int q = 0; q < HB / 8; q++) {
#pragma unroll
                    for (int x = 0; x < 8; x++) part[x] = acc[r][8 * q + x];
                    const double rw = wave_reduce8(part);                             // lane l: total of value red8_idx(l)
                    if (lane < 8) s_part[r][wid][h0 + 8 * q + red8_idx(lane)] = rw;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < (STM_NB * STM_NB + QA_NT - 1) / QA_NT; q++) {
            const int e = tid + QA_NT * q;
            if (e < STM_NB * STM_NB) s_T[e % STM_NB][e / STM_NB] = treg[q];                // s_T[row][col], padded rows: no bank conflicts below
        }
        __syncthreads();
        if (tid < STM_NB * R) {
            const int r = tid / STM_NB, j = tid % STM_NB;
            double v = 0;
#pragma unroll
            for (int w = 0; w < QA_NW; w++) v += s_part[r][w][j];
            s_w[r][j] = v;
        }
        __syncthreads();
        // ---- y = T'w (Q'x) or T w (Q x); T upper triangular ----
        if (tid < STM_NB * R) {
            const int r = tid / STM_NB, j = tid % STM_NB;
            double y = 0;
            if (method == 0) { for (int q = 0; q <= j; q++) y += s_T[q][j] * s_w[r][q]; }
            else { for (int q = j; q < STM_NB; q++) y += s_T[j][q] * s_w[r][q]; }
            s_y[r][j] = y;
        }
        __syncthreads();
        // ---- x -= V y ----
        // (in passes of HB reflectors as well: the R x HB entries of y a pass needs stay in registers; a row's x goes through LDS
        //  between two passes -- the same subtractions in the same order)
#pragma unroll
        for (int h0 = 0; h0 < STM_NB; h0 += HB) {
            for (int i = r0 + tid; i < r1; i += QA_NT) {
                double a[R];
#pragma unroll
                for (int r = 0; r < R; r++) a[r] = xs[r * xst + i];
#pragma unroll
                for (int j = 0; j < HB; j++) {
                    const double val = Vp[i + (long long)min(h0 + j, nbp - 1) * ld];
                    const double v = (i > s_d[h0 + j] && i < s_t[h0 + j]) ? val : ((i == s_d[h0 + j]) ? 1.0 : 0.0);
#pragma unroll
                    for (int r = 0; r < R; r++) a[r] -= v * s_y[r][h0 + j];
                }
#pragma unroll
                for (int r = 0; r < R; r++) xs[r * xst + i] = a[r];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < fm; i += QA_NT) {
        const int h = Hi[i];
#pragma unroll
        for (int r = 0; r < R; r++)
            if (r < nr) W[(long long)r * B.w + h] = xs[r * xst + i];
    }
}

// One front of the back substitution R x = y (fronts root -> leaves; reference qr_rsolve, SparseQR.c:2218-2470):
// y = the first rm rows of the front's slice of W (rm = live pivot columns), x of the non-pivotal columns comes from
// the ancestors, a dead pivot column gets x = 0 (basic solution), the live pivot columns form an rm x rm upper triangle
// whose row q is the q-th live column.
// ------------------------------------------------------------------------------------------------
// Q-apply for the large fronts (FrontSym::qbig): one workgroup streams a front's V at the fill rate of ONE CU
// (~45 GB/s: 10 ms for a 7818 x 7818 front), so the rows of such a front are split over workgroups (QB_ROWS rows
// each, one row per thread) and the panels become launches: launch k applies panel p_prev (x -= V y with y = T'w or
// T w, w = the slab partials of the previous launch summed in slab order) and forms the slab partials of
// w = V'x for panel p_next.  x lives in a device buffer Xf (gathered from / scattered to the work vector by
// k_qbig_prep / k_qbig_finish), the reflector numbering in Dq.
// ------------------------------------------------------------------------------------------------
#define QB_ROWS STM_QB_ROWS   // rows (= threads) of a step workgroup (128-row workgroups were measured: slower)
// All split fronts of a tree level advance together: blockIdx.y = index into the level's descriptor list
// (QbDesc: front, offsets of its slices of Xf / Dq / Wq, number of row slabs), launch k handles the k-th panel of each.
__global__ __launch_bounds__(QA_NT) void k_qbig_prep(DevCtx c, const QbDesc *__restrict__ qd, const double *W, double *Xf0, int *Dq0, RhsBatch B)
{
    W += (long long)blockIdx.y * B.w; Xf0 += (long long)blockIdx.y * B.xf;      // (the reflector numbering Dq0 is the same for every right-hand side)
    __shared__ int s_scan[QA_NW];
    const QbDesc d = qd[blockIdx.x];
    const FrontSym s = c.fs[d.f];
    const int fm = c.fnum[d.f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    const int *Hi = c.Hii + s.hip;
    double *Xf = Xf0 + d.xoff;
    for (int i = threadIdx.x; i < fm; i += QA_NT) Xf[i] = W[Hi[i]];
    qa_number_reflectors(s, fm, c.Stair + s.rp, c.Tau + s.rp, Dq0 + d.dqoff, s_scan);
}
__global__ __launch_bounds__(256) void k_qbig_finish(DevCtx c, const QbDesc *__restrict__ qd, double *W, const double *Xf0, RhsBatch B)
{
    W += (long long)blockIdx.z * B.w; Xf0 += (long long)blockIdx.z * B.xf;
    const QbDesc d = qd[blockIdx.y];
    const FrontSym s = c.fs[d.f];
    const int fm = c.fnum[d.f].fm;
    const int *Hi = c.Hii + s.hip;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < fm) W[Hi[i]] = Xf0[d.xoff + i];
}
__global__ __launch_bounds__(QB_ROWS) void k_qbig_step(DevCtx c, const QbDesc *__restrict__ qd, int k, int method, double *Xf0,
                                                     const int *Dq0, double *Wq0, RhsBatch B)
{
    Xf0 += (long long)blockIdx.z * B.xf; Wq0 += (long long)blockIdx.z * B.wq;
    __shared__ int s_d[2][STM_NB], s_t[2][STM_NB];
    __shared__ double s_part[QB_ROWS / 64][STM_NB], s_w[STM_NB], s_y[STM_NB], s_T[STM_NB][STM_NB + 1];
    const QbDesc qdd = qd[blockIdx.y];
    const int f = qdd.f, nslab = qdd.nslab;
    if ((int)blockIdx.x >= nslab) return;
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    // launch k: apply the (k-1)-th panel of the order, form the partials of the k-th (Q'x: ascending, Q x: descending)
    const int np = min(s.npanels, qdd.np_live);
    if (k > np) return;
    const int pp[2] = {(k >= 1) ? (method ? np - k : k - 1) : -1, (k < np) ? (method ? np - 1 - k : k) : -1};
    double *Xf = Xf0 + qdd.xoff;
    const int *Dq = Dq0 + qdd.dqoff;
    double *Wq = Wq0 + qdd.wqoff;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *St = c.Stair + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int i = blockIdx.x * QB_ROWS + tid, ic = min(i, fm - 1);
    // round trip 1: the reflector descriptions of both panels
    if (tid < 2 * STM_NB) {
        const int w = tid >> 5, j = tid & 31, p = pp[w];
        int d = -1, t = 0;
        if (p >= 0 && p * STM_NB + j < s.fn) { d = Dq[p * STM_NB + j]; t = St[p * STM_NB + j]; }
        s_d[w][j] = (d >= 0) ? d : STM_BIGROW;
        s_t[w][j] = (d >= 0) ? t : 0;
    }
    double x = (i < fm) ? Xf[i] : 0.0;
    __syncthreads();
    bool on[2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        int r0 = STM_BIGROW, r1 = 0;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) {
            r0 = min(r0, s_d[w][j]);
            r1 = max(r1, max(s_t[w][j], (s_d[w][j] < STM_BIGROW) ? s_d[w][j] + 1 : 0));
        }
        on[w] = (pp[w] >= 0 && r0 < STM_BIGROW && (int)blockIdx.x * QB_ROWS < r1 && ((int)blockIdx.x + 1) * QB_ROWS > r0);   // (uniform)
    }
    // round trip 2: everything both phases read, requested together (the launch is a chain of memory round trips)
    double v0[STM_NB] = {}, v1[STM_NB] = {}, treg[STM_NB * STM_NB / QB_ROWS], wsum = 0;
    if (on[0]) {
        const int nbp = min(STM_NB, s.fn - pp[0] * STM_NB);
        const double *Vp = F + (long long)(pp[0] * STM_NB) * ld;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) v0[j] = Vp[ic + (long long)min(j, nbp - 1) * ld];      // unconditional, masked below
        const double *T = c.Tall + (long long)(s.tpan + pp[0]) * STM_NB * STM_NB;
#pragma unroll
        for (int q = 0; q < STM_NB * STM_NB / QB_ROWS; q++) treg[q] = T[tid + QB_ROWS * q];
        if (tid < STM_NB) {
            const double *wp = Wq + (long long)(pp[0] & 1) * nslab * STM_NB;
            wsum += stm_ordered_sum<false>(wp + tid, STM_NB, nslab);                            // fixed order: deterministic
        }
    }
    if (on[1]) {
        const int nbp = min(STM_NB, s.fn - pp[1] * STM_NB);
        const double *Vp = F + (long long)(pp[1] * STM_NB) * ld;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) v1[j] = Vp[ic + (long long)min(j, nbp - 1) * ld];
    }
    if (on[0]) {
#pragma unroll
        for (int q = 0; q < STM_NB * STM_NB / QB_ROWS; q++) s_T[(tid + QB_ROWS * q) % STM_NB][(tid + QB_ROWS * q) / STM_NB] = treg[q];
        if (tid < STM_NB) s_w[tid] = wsum;
        __syncthreads();
        if (tid < STM_NB) {
            double y = 0;
            if (method == 0) { for (int q = 0; q <= tid; q++) y += s_T[q][tid] * s_w[q]; }
            else { for (int q = tid; q < STM_NB; q++) y += s_T[tid][q] * s_w[q]; }
            s_y[tid] = y;
        }
        __syncthreads();
        double a = x;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) {
            const double v = (i > s_d[0][j] && i < s_t[0][j]) ? v0[j] : ((i == s_d[0][j]) ? 1.0 : 0.0);
            a -= v * s_y[j];
        }
        if (i < fm && a != x) Xf[i] = a;
        x = (i < fm) ? a : 0.0;
    }
    if (pp[1] >= 0) {
        double acc[STM_NB];
#pragma unroll
        for (int j = 0; j < STM_NB; j++) {
            const double v = (i < fm && i > s_d[1][j] && i < s_t[1][j]) ? v1[j] : ((i == s_d[1][j]) ? 1.0 : 0.0);
            acc[j] = on[1] ? v * x : 0.0;
        }
        {
            double part[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int xx = 0; xx < 8; xx++) part[xx] = acc[8 * q + xx];
                const double rw = wave_reduce8(part);                             // lane l: total of value red8_idx(l)
                if (lane < 8) s_part[wid][8 * q + red8_idx(lane)] = rw;
            }
        }
        __syncthreads();
        if (tid < STM_NB) {
            double v = 0;
#pragma unroll
            for (int w = 0; w < QB_ROWS / 64; w++) v += s_part[w][tid];
            Wq[((long long)(pp[1] & 1) * nslab + blockIdx.x) * STM_NB + tid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Grouped split Q-apply (round 4): FOUR panels (128 reflectors) of a large front per launch instead of one.  The launches of
// k_qbig_step are a chain (launch k needs the x of launch k-1) of ~10 us each whatever they do -- two dependent memory round trips --,
// 1473 of them per Q'b on the default workload.  The product of the four block reflectors of a group is ONE block reflector,
//     H_0 H_1 H_2 H_3 = I - V T4 V',    T4 = [T_0 X_01 X_02 X_03; 0 T_1 X_12 X_13; 0 0 T_2 X_23; 0 0 0 T_3]   (128 x 128),
// with the blocks X_ij of the compact WY recurrence ( [X_0j; ..; X_{j-1,j}] = -T4(0:j, 0:j) [G_0j; ..; G_{j-1,j}] T_j, G_ij = V_i'V_j ).
// T4 of every group of every split front is built ONCE per factorization, at the first Q-apply (k_qt4_number, k_qt4_build: the Gram
// matrix of the group's 128 columns by MFMA, then sixteen 32 x 32 block products), and kept in both layouts (T4 for Q x, its transpose
// for Q'x: the matrix-vector product of a launch reads it with the lanes along the output index).
// ------------------------------------------------------------------------------------------------
#define QG 4
#define QGN (QG * STM_NB)
#define QT4_DOUBLES (2 * QGN * QGN)                  // per group: T4 column-major, then T4 row-major
struct Qt4Item { int f, g; long long off, dqo; };    // T4 of group g of front f at T4all + off; the front's reflector numbering at Dq4 + dqo

__global__ __launch_bounds__(QA_NT) void k_qt4_number(DevCtx c, const int *__restrict__ fl, const long long *__restrict__ dqo, int *Dq4)
{
    __shared__ int s_scan[QA_NW];
    const int f = fl[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    qa_number_reflectors(s, fm, c.Stair + s.rp, c.Tau + s.rp, Dq4 + dqo[blockIdx.x], s_scan);
}

#define QT4_VS 36
__global__ __launch_bounds__(512) void k_qt4_build(DevCtx c, const Qt4Item *__restrict__ items, const int *__restrict__ Dq4, double *T4all)
{
    extern __shared__ double lds[];                   // Gram phase: the chunk image [QGN][QT4_VS]; block phase: five 32 x 33 blocks
    __shared__ int s_d[QGN], s_t[QGN], s_rng[2][8], s_live[QG];
    const Qt4Item it = items[blockIdx.x];
    const int f = it.f, g = it.g;
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    double *T4c = T4all + it.off, *T4r = T4c + QGN * QGN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int k0 = g * QGN;
    if (tid < QGN) {
        const int k = k0 + tid;
        int d = -1, t = 0;
        if (fm > 0 && k < s.fn) { d = Dq4[it.dqo + k]; t = c.Stair[s.rp + k]; }
        s_d[tid] = (d >= 0) ? d : STM_BIGROW;
        s_t[tid] = (d >= 0) ? max(t, d + 1) : 0;
        const int lo = wave_max_int(-s_d[tid]), hi = wave_max_int(s_t[tid]);
        if (lane == 0) { s_rng[0][wid] = -lo; s_rng[1][wid] = hi; }
    }
    __syncthreads();
    if (tid < QG) {
        int lv = 0;
        for (int j = 0; j < STM_NB; j++) lv |= (s_d[STM_NB * tid + j] < STM_BIGROW);
        s_live[tid] = lv;
    }
    const int rmin = min(s_rng[0][0], s_rng[0][1]), rmax = min(fm, max(s_rng[1][0], s_rng[1][1]));
    if (rmin >= STM_BIGROW || rmax <= rmin) {                       // no live reflector in the group: T4 = 0
        for (int e = tid; e < QT4_DOUBLES; e += 512) st_agent(&T4c[e], 0.0);
        return;
    }
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    // ---- G = V'V (128 x 128) over the rows [rmin, rmax): wave w the tile row w ----
    d4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = (d4){0, 0, 0, 0};
    double *Vs = lds;
    const int srow = tid & 31, scg = tid >> 5;
    for (int r0 = rmin; r0 < rmax; r0 += 32) {
        const int i = r0 + srow, ic = min(i, fm - 1);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int col = scg * 8 + q, k = min(k0 + col, s.fn - 1);
            const double val = F[ic + (long long)k * ld];
            const int d = s_d[col], t = s_t[col];
            Vs[col * QT4_VS + srow] = (i == d) ? 1.0 : ((i > d && i < t) ? val : 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const double a = Vs[(16 * wid + l15) * QT4_VS + 4 * kk + l4];
#pragma unroll
            for (int tc = 0; tc < 8; tc++) {
                const double b = Vs[(16 * tc + l15) * QT4_VS + 4 * kk + l4];
                acc[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[tc], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // G to the T4c area (column-major; overwritten block column by block column below), write-through: read back by other waves
#pragma unroll
    for (int tc = 0; tc < 8; tc++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_agent(&T4c[(16 * wid + l4 + 4 * r) + QGN * (16 * tc + l15)], acc[tc][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- the blocks of T4 ----
    double *Ab = lds, *Bb = Ab + 32 * 33, *Yb = Bb + 32 * 33;        // Yb: three blocks
    const int oa = tid & 31, ob = tid >> 5;                          // my two outputs of a block product: (oa, ob), (oa, ob + 16)
    const int np = s.npanels;
    // T of panel 4 g + i (column-major); zero beyond the front's panels and for a panel without a live reflector -- the factorization
    // never wrote T of the panels behind the one where the rows ran out (the per-panel kernels never read it either)
    auto load_T = [&](int i, double *dst) {
        const int p = QG * g + i;
        const bool live = p < np && s_live[i];
        const double *T = c.Tall + (long long)(s.tpan + min(p, np - 1)) * STM_NB * STM_NB;
        for (int e = tid; e < 32 * 32; e += 512) dst[(e & 31) * 33 + (e >> 5)] = live ? T[e] : 0.0;           // dst[row][col]
    };
    auto load_blk = [&](int i, int j, double *dst) {                 // block (i, j) of the T4c area
        for (int e = tid; e < 32 * 32; e += 512) dst[(e & 31) * 33 + (e >> 5)] = ld_agent(&T4c[(32 * i + (e & 31)) + QGN * (32 * j + (e >> 5))]);
    };
    auto mm = [&](const double *A, const double *B, double &o0, double &o1) {      // += A B at my two outputs
#pragma unroll 8
        for (int q = 0; q < 32; q++) {
            const double a = A[oa * 33 + q];
            o0 += a * B[q * 33 + ob];
            o1 += a * B[q * 33 + ob + 16];
        }
    };
    for (int j = 1; j < QG; j++) {
        load_T(j, Bb);
        for (int i = 0; i < j; i++) {                                // Y_i = G_ij T_j
            __syncthreads();
            load_blk(i, j, Ab);
            __syncthreads();
            double y0 = 0, y1 = 0;
            mm(Ab, Bb, y0, y1);
            Yb[i * 32 * 33 + oa * 33 + ob] = y0;
            Yb[i * 32 * 33 + oa * 33 + ob + 16] = y1;
        }
        for (int i = 0; i < j; i++) {                                // X_ij = - sum_{l = i .. j-1} T4_il Y_l
            double x0 = 0, x1 = 0;
            for (int l = i; l < j; l++) {
                __syncthreads();
                if (l == i) load_T(i, Ab);
                else load_blk(i, l, Ab);
                __syncthreads();
                mm(Ab, Yb + l * 32 * 33, x0, x1);
            }
            st_agent(&T4c[(32 * i + oa) + QGN * (32 * j + ob)], -x0);
            st_agent(&T4c[(32 * i + oa) + QGN * (32 * j + ob + 16)], -x1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // diagonal blocks, zeros below them, and the row-major copy
    for (int i = 0; i < QG; i++) {
        __syncthreads();
        load_T(i, Ab);
        __syncthreads();
        for (int e = tid; e < 32 * 32; e += 512) st_agent(&T4c[(32 * i + (e & 31)) + QGN * (32 * i + (e >> 5))], Ab[(e & 31) * 33 + (e >> 5)]);
    }
    for (int e = tid; e < QGN * QGN; e += 512) {
        const int r = e & (QGN - 1), cc = e >> 7;
        if ((r >> 5) > (cc >> 5)) st_agent(&T4c[e], 0.0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = tid; e < QGN * QGN; e += 512) {
        const int cc = e & (QGN - 1), r = e >> 7;                    // T4r[c + 128 r] = T4(r, c)
        T4r[e] = ld_agent(&T4c[r + QGN * cc]);
    }
}

// Several right-hand sides per workgroup (R of the batch, template): V, its masks and T4 are loaded ONCE and applied to R vectors --
// with one vector per workgroup a batch of 32 read every reflector 32 times (73.9 us per launch against 24.6 for one vector on the
// default workload).  Per vector the arithmetic and its order are those of R = 1: a batch gives the bits of its vectors one by one.
template <int R>
__global__ __launch_bounds__(QB_ROWS) void k_qbig_step4(DevCtx c, const QbDesc *__restrict__ qd, const long long *__restrict__ t4off, int k,
                                                      int method, double *Xf0, const int *Dq0, double *Wq0, const double *T4all, RhsBatch B, int nb)
{
    const int nr = min(R, nb - (int)blockIdx.z * R);               // vectors of this workgroup (uniform)
    Xf0 += (long long)blockIdx.z * R * B.xf; Wq0 += (long long)blockIdx.z * R * B.wq4;
    __shared__ int s_d[2][QGN], s_t[2][QGN], s_rng[2][2][2];
    __shared__ double s_part[R][QB_ROWS / 64][QGN], s_w[R][QGN], s_y[R][QGN], s_yp[R][QB_ROWS / QGN][QGN];
    const QbDesc qdd = qd[blockIdx.y];
    const int f = qdd.f, nslab = qdd.nslab;
    if ((int)blockIdx.x >= nslab) return;
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    const int ng = (min(s.npanels, qdd.np_live) + QG - 1) / QG;
    if (k > ng) return;
    // launch k: apply the (k-1)-th group of the order, form the partials of the k-th (Q'x: ascending, Q x: descending)
    const int gg[2] = {(k >= 1) ? (method ? ng - k : k - 1) : -1, (k < ng) ? (method ? ng - 1 - k : k) : -1};
    double *Xf = Xf0 + qdd.xoff;
    const int *Dq = Dq0 + qdd.dqoff;
    double *Wq = Wq0 + (long long)qdd.wqoff * QG;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *St = c.Stair + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int i = blockIdx.x * QB_ROWS + tid, ic = min(i, fm - 1);
    if (tid < 2 * QGN) {
        const int w = tid >> 7, j = tid & (QGN - 1), kc = gg[w] * QGN + j;
        int d = -1, t = 0;
        if (gg[w] >= 0 && kc < s.fn) { d = Dq[kc]; t = St[kc]; }
        s_d[w][j] = (d >= 0) ? d : STM_BIGROW;
        s_t[w][j] = (d >= 0) ? max(t, d + 1) : 0;
        const int lo = wave_max_int(-s_d[w][j]), hi = wave_max_int(s_t[w][j]);
        if (lane == 0) { s_rng[w][(tid >> 6) & 1][0] = -lo; s_rng[w][(tid >> 6) & 1][1] = hi; }
    }
    double x[R];
#pragma unroll
    for (int r = 0; r < R; r++) x[r] = (i < fm && r < nr) ? Xf[(long long)r * B.xf + i] : 0.0;
    __syncthreads();
    bool on[2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const int r0 = min(s_rng[w][0][0], s_rng[w][1][0]), r1 = max(s_rng[w][0][1], s_rng[w][1][1]);
        on[w] = (gg[w] >= 0 && r0 < STM_BIGROW && (int)blockIdx.x * QB_ROWS < r1 && ((int)blockIdx.x + 1) * QB_ROWS > r0);      // (uniform)
    }
    // Both groups' columns go through ONE pipeline of 16 sub-blocks of 16 columns (8 of the group to apply, then 8 of the group whose
    // partials are formed: their loads do not depend on x), four sub-blocks (64 loads per thread) in flight: a launch is a handful of
    // memory round trips whatever it does, so what counts is how many loads each of them carries.
    constexpr int SB = 16, NSB = QGN / SB, DEPTH = 4;
    auto load_v = [&](int idx, double (&v)[SB]) {       // sub-block idx of the pipeline (uniform: nothing is loaded for an idle phase)
        const int w = idx / NSB, sb = idx % NSB;
        if (!on[w]) return;
        const double *Vp = F + (long long)(gg[w] * QGN + sb * SB) * ld;
        const int nbp = s.fn - (gg[w] * QGN + sb * SB);
#pragma unroll
        for (int j = 0; j < SB; j++) v[j] = (nbp > 0) ? Vp[ic + (long long)min(j, nbp - 1) * ld] : 0.0;     // unconditional, masked below
    };
    double buf[DEPTH][SB];
#pragma unroll
    for (int q = 0; q < DEPTH; q++)
#pragma unroll
        for (int j = 0; j < SB; j++) buf[q][j] = 0.0;
    if (on[0]) {
        // y = T4' w (Q'x) or T4 w (Q x): lanes along the output index, four parts of 32 terms each -- requested first
        const double *M = T4all + t4off[blockIdx.y] + (long long)gg[0] * QT4_DOUBLES + (method ? 0 : QGN * QGN);
        const int o = tid & (QGN - 1), part = tid >> 7;
        double m[32];
#pragma unroll
        for (int q = 0; q < 32; q++) m[q] = M[o + QGN * (32 * part + q)];
        if (tid < QGN) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r < nr) s_w[r][tid] = stm_ordered_sum<false>(Wq + (long long)r * B.wq4 + (long long)(gg[0] & 1) * nslab * QGN + tid, QGN, nslab);      // fixed order
        }
#pragma unroll
        for (int q = 0; q < DEPTH; q++) load_v(q, buf[q]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (r >= nr) continue;
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
            for (int q = 0; q < 32; q += 4) {
                p0 += m[q] * s_w[r][32 * part + q];
                p1 += m[q + 1] * s_w[r][32 * part + q + 1];
                p2 += m[q + 2] * s_w[r][32 * part + q + 2];
                p3 += m[q + 3] * s_w[r][32 * part + q + 3];
            }
            s_yp[r][part][o] = (p0 + p1) + (p2 + p3);
        }
        __syncthreads();
        if (tid < QGN) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r < nr) s_y[r][tid] = (s_yp[r][0][tid] + s_yp[r][1][tid]) + (s_yp[r][2][tid] + s_yp[r][3][tid]);
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int q = 0; q < DEPTH; q++) load_v(q, buf[q]);
    }
    double a[R];
#pragma unroll
    for (int r = 0; r < R; r++) a[r] = x[r];
#pragma unroll
    for (int idx = 0; idx < 2 * NSB; idx++) {
        double (&v)[SB] = buf[idx % DEPTH];
        const int w = idx / NSB, sb = idx % NSB;
        if (idx == NSB) {                               // between the phases: x after the group that was applied
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (on[0] && i < fm && r < nr && a[r] != x[r]) Xf[(long long)r * B.xf + i] = a[r];
                x[r] = (i < fm) ? a[r] : 0.0;
            }
        }
        if (w == 0) {
            if (on[0]) {
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const int d = s_d[0][sb * SB + j], t = s_t[0][sb * SB + j];
                    const double vv = (i > d && i < t) ? v[j] : ((i == d) ? 1.0 : 0.0);
#pragma unroll
                    for (int r = 0; r < R; r++) a[r] -= vv * s_y[r][sb * SB + j];
                }
            }
        } else if (gg[1] >= 0) {
            double vm[SB];
#pragma unroll
            for (int j = 0; j < SB; j++) {
                const int d = s_d[1][sb * SB + j], t = s_t[1][sb * SB + j];
                vm[j] = (i < fm && i > d && i < t) ? v[j] : ((i == d) ? 1.0 : 0.0);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r >= nr) continue;
                double part[8];
#pragma unroll
                for (int q = 0; q < SB / 8; q++) {
#pragma unroll
                    for (int xx = 0; xx < 8; xx++) part[xx] = on[1] ? vm[8 * q + xx] * x[r] : 0.0;
                    const double rw = wave_reduce8(part);                             // lane l: total of value red8_idx(l)
                    if (lane < 8) s_part[r][wid][sb * SB + 8 * q + red8_idx(lane)] = rw;
                }
            }
        }
        if (idx + DEPTH < 2 * NSB) load_v(idx + DEPTH, v);
    }
    if (gg[1] >= 0) {
        __syncthreads();
        if (tid < QGN) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (r >= nr) continue;
                double vsum = 0;
#pragma unroll
                for (int w = 0; w < QB_ROWS / 64; w++) vsum += s_part[r][w][tid];
                Wq[(long long)r * B.wq4 + ((long long)(gg[1] & 1) * nslab + blockIdx.x) * QGN + tid] = vsum;
            }
        }
    }
}

#define RS_NT 1024               // the back substitution streams R through one workgroup: more loads in flight
template <int R>                                           // (R right-hand sides of the batch per workgroup: R is read once for all of them)
__global__ __launch_bounds__(RS_NT) void k_rsolve(DevCtx c, const int *__restrict__ flist, const int *__restrict__ Rj,
                                                  const double *W, double *X, int *err, RhsBatch B, int nb)
{
    const int nr = min(R, nb - (int)blockIdx.y * R);
    W += (long long)blockIdx.y * R * B.w; X += (long long)blockIdx.y * R * B.x;
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[RS_NT / 64];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const FrontNum nm = c.fnum[f];
    const int fp = s.fp, fn = s.fn, fm = nm.fm;
    if (fp <= 0 || s.qbig) return;                      // (qbig: k_rbig_* below)
    const int tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    const int *Hi = c.Hii + s.hip;
    const int *rj = Rj + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int ast = (fp + 1) & ~1, ost = (fn - fp + 1) & ~1;
    double *acc = dyn_lds;                              // [R][ast]
    double *xo = acc + (size_t)R * ast;                 // [R][ost] x of the non-pivotal columns
    int *lc = (int *)(xo + (size_t)R * ost);            // [fp] live pivot columns, compact
    // ---- live pivot columns (HStair != 0 and a row left for the diagonal), dead ones: x = 0 ----
    int rm;
    {
        const int per = (fp + RS_NT - 1) / RS_NT;
        const int k0 = min(fp, tid * per), k1 = min(fp, k0 + per);
        int cnt = 0;
        for (int k = k0; k < k1; k++) cnt += (St[k] != 0);
        int total;
        const int incl = qa_incl_scan<RS_NT / 64>(cnt, s_scan, &total);
        int q = incl - cnt;
        for (int k = k0; k < k1; k++) {
            if (St[k] != 0 && q < fm) lc[q] = k;
            else if (St[k] == 0) {
#pragma unroll
                for (int r = 0; r < R; r++)
                    if (r < nr) X[(long long)r * B.x + s.col1 + k] = 0.0;
            }
            q += (St[k] != 0);
        }
        rm = min(total, fm);
    }
    for (int k = fp + tid; k < fn; k += RS_NT) {
        const int j = rj[k];
#pragma unroll
        for (int r = 0; r < R; r++) xo[r * ost + k - fp] = (r < nr) ? X[(long long)r * B.x + j] : 0.0;
    }
    __syncthreads();
    if (rm != nm.rank && tid == 0) atomicExch(err, 1);  // (cannot happen: same rule as the factorization)
    // acc = y - R12 x_others : thread per row, columns streamed (coalesced over the rows)
    for (int i = tid; i < rm; i += RS_NT) {
        double a0[R], a1[R], a2[R], a3[R];              // (four partial sums, 16 loads in flight: as k_rbig_init)
        const int h = Hi[i];
#pragma unroll
        for (int r = 0; r < R; r++) { a0[r] = (r < nr) ? W[(long long)r * B.w + h] : 0.0; a1[r] = a2[r] = a3[r] = 0; }
        int k = fp;
        for (; k + 16 <= fn; k += 16) {
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                const double f0 = F[i + (long long)(k + u) * ld], f1 = F[i + (long long)(k + u + 1) * ld];
                const double f2 = F[i + (long long)(k + u + 2) * ld], f3 = F[i + (long long)(k + u + 3) * ld];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    a0[r] -= f0 * xo[r * ost + k + u - fp];
                    a1[r] -= f1 * xo[r * ost + k + u + 1 - fp];
                    a2[r] -= f2 * xo[r * ost + k + u + 2 - fp];
                    a3[r] -= f3 * xo[r * ost + k + u + 3 - fp];
                }
            }
        }
        for (; k < fn; k++) {
            const double f0 = F[i + (long long)k * ld];
#pragma unroll
            for (int r = 0; r < R; r++) a0[r] -= f0 * xo[r * ost + k - fp];
        }
#pragma unroll
        for (int r = 0; r < R; r++) acc[r * ast + i] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
    }
    __syncthreads();
    // triangle: blocked back substitution over the compact list.  Per block of QS_NB live columns: the diagonal triangle
    // goes to LDS and one wave per right-hand side solves it there (no global latency inside the 32 dependent steps), then every
    // thread updates its rows of acc with the block's columns (coalesced over the rows, 32 independent loads in flight).
    constexpr int QS_NB = 32;
    __shared__ double s_tri[QS_NB][QS_NB + 1];
    __shared__ double s_x[R][QS_NB];
    for (int kb = ((max(rm, 1) - 1) / QS_NB) * QS_NB; kb >= 0 && rm > 0; kb -= QS_NB) {
        const int nbk = min(QS_NB, rm - kb);
        for (int e = tid; e < QS_NB * QS_NB; e += RS_NT) {
            const int i = e % QS_NB, j = e / QS_NB;
            s_tri[i][j] = (i < nbk && j < nbk && i <= j) ? F[(kb + i) + (long long)lc[kb + j] * ld] : 0.0;
        }
        __syncthreads();
        if (tid < 64 * R) {
            // wave r, lane i owns row i of the triangle (i < nbk) of right-hand side r: x_j for j = nbk-1 .. 0
            const int i = tid & 63, r = tid >> 6;
            double a = (i < nbk) ? acc[r * ast + kb + i] : 0.0;
            for (int j = nbk - 1; j >= 0; j--) {
                const double aj = __shfl(a, j, 64);
                const double xj = aj / s_tri[j][j];
                if (i < j) a -= s_tri[i][j] * xj;
                if (i == j) s_x[r][j] = xj;
            }
        }
        __syncthreads();
        if (tid < nbk) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (r < nr) X[(long long)r * B.x + s.col1 + lc[kb + tid]] = s_x[r][tid];
        }
        for (int i = tid; i < kb; i += RS_NT) {
            double a[R];
#pragma unroll
            for (int r = 0; r < R; r++) a[r] = acc[r * ast + i];
#pragma unroll 8
            for (int j = 0; j < nbk; j++) {
                const double fv = F[i + (long long)lc[kb + j] * ld];
#pragma unroll
                for (int r = 0; r < R; r++) a[r] -= fv * s_x[r][j];
            }
#pragma unroll
            for (int r = 0; r < R; r++) acc[r * ast + i] = a[r];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// R' x = b (qr_private_rtsolve, reference SparseQR.c:2522-2700), fronts leaves -> root, one workgroup per front.
// Every front carries a vector u over its fn columns: b minus the contributions of the rows solved so far.  A front
// starts with u = b on its pivotal columns and 0 elsewhere, ADDS its children's pass vectors (their u on the columns they
// hand up, mapped through Rjrel -- the assembly's column map, children in order: deterministic, no atomics), solves its
// triangle R11' x = u(pivots) forwards over the live pivot columns, and passes u(non-pivotal) - R12' x up.
//   Bp: b in R's column order (length n); U: the pass vectors, slot Rp[f] + k (rjsize); Xr: x in R's global row order
//   (rowbase[f] = rows of R above front f's).  Dead pivot columns have no equation (the squeezed R of the reference).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_NT) void k_rtsolve(DevCtx c, const int *__restrict__ flist, const double *__restrict__ Bp,
                                                   double *U, double *Xr, const int *__restrict__ rowbase, RhsBatch B)
{
    Bp += (long long)blockIdx.y * B.x; U += (long long)blockIdx.y * B.u; Xr += (long long)blockIdx.y * B.w;
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[RS_NT / 64];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const FrontNum nm = c.fnum[f];
    const int fp = s.fp, fn = s.fn, fm = nm.fm;
    const int tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    double *u = dyn_lds;                                // [fn]
    double *x = u + ((fn + 1) & ~1);                    // [min(fp, fm)] the front's rows of x
    int *lc = (int *)(x + ((min(fp, max(fm, 1)) + 2) & ~1));   // [fp] live pivot columns, compact
    int rm;
    {
        const int per = (fp + RS_NT - 1) / RS_NT;
        const int k0 = min(fp, tid * per), k1 = min(fp, k0 + per);
        int cnt = 0;
        for (int k = k0; k < k1; k++) cnt += (St[k] != 0);
        int total;
        const int incl = qa_incl_scan<RS_NT / 64>(cnt, s_scan, &total);
        int q = incl - cnt;
        for (int k = k0; k < k1; k++) {
            if (St[k] != 0 && q < fm) lc[q] = k;
            q += (St[k] != 0);
        }
        rm = min(total, fm);
    }
    for (int k = tid; k < fn; k += RS_NT) u[k] = (k < fp) ? Bp[s.col1 + k] : 0.0;
    __syncthreads();
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const FrontSym cs = c.fs[ch];
        const int pc = cs.rp + cs.fp, cn = cs.fn - cs.fp;
        for (int cj = tid; cj < cn; cj += RS_NT) u[c.Rjrel[pc + cj]] += U[pc + cj];     // (distinct targets within a child)
        __syncthreads();
    }
    // forward substitution over the live pivot columns in blocks of 32: (a) the block's right-hand sides lose the rows
    // solved before (32 columns x 32 row lanes per pass), (b) one wave solves the 32 x 32 lower triangle R' in LDS
    constexpr int QS_NB = 32;
    __shared__ double s_tri[QS_NB][QS_NB + 1];
    __shared__ double s_rhs[QS_NB];
    const int lane32 = tid & 31, grp = tid >> 5;                          // 32 groups of 32 lanes
    for (int kb = 0; kb < rm; kb += QS_NB) {
        const int nb = min(QS_NB, rm - kb);
        {
            const int j = grp;                                            // column of the block
            double a = 0;
            if (j < nb) {
                const double *col = F + (long long)lc[kb + j] * ld;
                for (int r = lane32; r < kb; r += 32) a += col[r] * x[r];
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane32 == 0 && j < nb) s_rhs[j] = u[lc[kb + j]] - a;
        }
        for (int e = tid; e < QS_NB * QS_NB; e += RS_NT) {
            const int i = e % QS_NB, j = e / QS_NB;                       // R(kb + i, column j of the block), i <= j
            s_tri[i][j] = (i < nb && j < nb && i <= j) ? F[(kb + i) + (long long)lc[kb + j] * ld] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {
            // lane j owns equation j: x_j = (rhs_j - sum_{i<j} R(i,j) x_i) / R(j,j)
            const int j = tid;
            double a = (j < nb) ? s_rhs[j] : 0.0;
            for (int i = 0; i < nb; i++) {
                const double ai = __shfl(a, i, 64);
                const double xi = ai / s_tri[i][i];
                if (j > i && j < nb) a -= s_tri[i][j] * xi;
                if (j == i) x[kb + i] = xi;
            }
        }
        __syncthreads();
    }
    // pass up: u(non-pivotal) -= R12' x ; a thread per column, the front's rows of R streamed
    for (int k = fp + tid; k < fn; k += RS_NT) {
        const double *col = F + (long long)k * ld;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int r = 0;
        for (; r + 4 <= rm; r += 4) { a0 += col[r] * x[r]; a1 += col[r + 1] * x[r + 1]; a2 += col[r + 2] * x[r + 2]; a3 += col[r + 3] * x[r + 3]; }
        for (; r < rm; r++) a0 += col[r] * x[r];
        U[s.rp + k] = u[k] - ((a0 + a1) + (a2 + a3));
    }
    for (int r = tid; r < rm; r += RS_NT) Xr[rowbase[f] + r] = x[r];
}

// scatter: out[perm[i]] = in[i]   gather: out[i] = in[perm[i]]   (perm == nullptr: identity)
// ------------------------------------------------------------------------------------------------
// Back substitution for the large fronts (FrontSym::qbig), rows split over workgroups as in k_qbig_*: prep (live pivot
// columns -> Lc, rm), init (acc = y - R12 x_others, a row per thread), then one launch per block of 32 live columns
// from the last to the first: every active workgroup solves the 32 x 32 triangle for itself (one wave, LDS) and updates
// its rows above the block; workgroup 0 stores x.  All split fronts of a level advance together (blockIdx.y).
// acc = the front's slice of Xf, Lc = its slice of Dq (the Q-apply has finished with both).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_NT) void k_rbig_prep(DevCtx c, const QbDesc *__restrict__ qd, double *X, int *Lc0, int *Rm, int *err, RhsBatch B)
{
    X += (long long)blockIdx.y * B.x;                      // (the live-column lists Lc0 / Rm are the same for every right-hand side)
    __shared__ int s_scan[RS_NT / 64];
    const QbDesc d = qd[blockIdx.x];
    const FrontSym s = c.fs[d.f];
    const FrontNum nm = c.fnum[d.f];
    const int fp = s.fp, fm = nm.fm, tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    int *lc = Lc0 + d.dqoff;
    const int per = (fp + RS_NT - 1) / RS_NT;
    const int k0 = min(fp, tid * per), k1 = min(fp, k0 + per);
    int cnt = 0;
    for (int k = k0; k < k1; k++) cnt += (St[k] != 0);
    int total;
    const int incl = qa_incl_scan<RS_NT / 64>(cnt, s_scan, &total);
    int q = incl - cnt;
    for (int k = k0; k < k1; k++) {
        if (St[k] != 0 && q < fm) lc[q] = k;
        else if (St[k] == 0) X[s.col1 + k] = 0.0;       // dead pivot column: basic solution
        q += (St[k] != 0);
    }
    if (tid == 0) {
        const int rm = min(total, fm);
        Rm[blockIdx.x] = rm;
        if (rm != nm.rank) atomicExch(err, 1);
    }
}
template <int R>                                           // (R right-hand sides of the batch per workgroup: R12 is read once for all of them)
__global__ __launch_bounds__(STM_QB_ROWS) void k_rbig_init(DevCtx c, const QbDesc *__restrict__ qd, const int *__restrict__ Rj,
                                                           const double *W, const double *X, double *Acc0, const int *Rm, RhsBatch B, int nb)
{
    const int nr = min(R, nb - (int)blockIdx.z * R);
    W += (long long)blockIdx.z * R * B.w; X += (long long)blockIdx.z * R * B.x; Acc0 += (long long)blockIdx.z * R * B.xf;
    // y - R12 x2 for the rows of the live pivot columns.  A workgroup takes 64 rows; its eight waves share the non-pivotal columns
    // (chunks of 16, wave w the chunks w, w + 8, ...) and their partial sums are added in wave order (round 4: a thread per row ran
    // through all the columns alone -- 250 dependent round trips on a front with 4000 of them, 276 us per launch).
    __shared__ double s_p[R][STM_QB_ROWS / 64][64];
    const QbDesc d = qd[blockIdx.y];
    const FrontSym s = c.fs[d.f];
    const int rm = Rm[blockIdx.y];
    const int lrow = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lrow;
    if (blockIdx.x * 64 >= rm) return;
    const int ic = min(i, rm - 1);
    const int *rj = Rj + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    // (coalesced over the rows; X[rj[k]] uniform; four partial sums and an unrolled body keep 16 loads in flight)
    double a0[R], a1[R], a2[R], a3[R];
#pragma unroll
    for (int r = 0; r < R; r++) a0[r] = a1[r] = a2[r] = a3[r] = 0;
    const long long xs = (nr > 1) ? B.x : 0;                // (vectors beyond the batch read the first one's x: never stored)
    const int nfull = (s.fn - s.fp) / 16;
    for (int ch = cl; ch < nfull; ch += STM_QB_ROWS / 64) {
        const int k = s.fp + 16 * ch;
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            const double f0 = F[ic + (long long)(k + u) * ld], f1 = F[ic + (long long)(k + u + 1) * ld];
            const double f2 = F[ic + (long long)(k + u + 2) * ld], f3 = F[ic + (long long)(k + u + 3) * ld];
            const int j0 = rj[k + u], j1 = rj[k + u + 1], j2 = rj[k + u + 2], j3 = rj[k + u + 3];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double *Xr = X + (r < nr ? r : 0) * xs;
                a0[r] -= f0 * Xr[j0];
                a1[r] -= f1 * Xr[j1];
                a2[r] -= f2 * Xr[j2];
                a3[r] -= f3 * Xr[j3];
            }
        }
    }
    if (cl == 0)
        for (int k = s.fp + 16 * nfull; k < s.fn; k++) {
            const double f0 = F[ic + (long long)k * ld];
            const int j0 = rj[k];
#pragma unroll
            for (int r = 0; r < R; r++) a0[r] -= f0 * X[(r < nr ? r : 0) * xs + j0];
        }
#pragma unroll
    for (int r = 0; r < R; r++) s_p[r][cl][lrow] = (a0[r] + a1[r]) + (a2[r] + a3[r]);
    __syncthreads();
    if (cl < nr && i < rm) {                               // (wave r adds up right-hand side r)
        double a = W[(long long)cl * B.w + c.Hii[s.hip + i]];
#pragma unroll
        for (int w = 0; w < STM_QB_ROWS / 64; w++) a += s_p[cl][w][lrow];
        Acc0[(long long)cl * B.xf + d.xoff + i] = a;
    }
}
__global__ __launch_bounds__(STM_QB_ROWS) void k_rbig_step(DevCtx c, const QbDesc *__restrict__ qd, int kstep, double *X, double *Acc0,
                                                           const int *Lc0, const int *Rm, RhsBatch B)
{
    X += (long long)blockIdx.z * B.x; Acc0 += (long long)blockIdx.z * B.xf;
    constexpr int QS_NB = 32;
    __shared__ double s_tri[QS_NB][QS_NB + 1];
    __shared__ double s_x[QS_NB];
    const QbDesc d = qd[blockIdx.y];
    const int rm = Rm[blockIdx.y];
    const int nblk = (rm + QS_NB - 1) / QS_NB;
    if (kstep >= nblk) return;
    const int kb = (nblk - 1 - kstep) * QS_NB, nb = min(QS_NB, rm - kb);
    const int sl = blockIdx.x, tid = threadIdx.x;
    if (sl != 0 && sl * STM_QB_ROWS >= kb) return;      // no row above the block (workgroup 0 always runs: it stores x)
    const FrontSym s = c.fs[d.f];
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int *lc = Lc0 + d.dqoff;
    double *acc = Acc0 + d.xoff;
    // my row's entries of the block's columns: requested before the triangle is solved
    const int i = sl * STM_QB_ROWS + tid;
    double rv[QS_NB];
#pragma unroll
    for (int j = 0; j < QS_NB; j++) rv[j] = F[min(i, max(kb - 1, 0)) + (long long)lc[kb + min(j, nb - 1)] * ld];
    double a = (i < kb) ? acc[i] : 0.0;
    for (int e = tid; e < QS_NB * QS_NB; e += STM_QB_ROWS) {
        const int ti = e % QS_NB, tj = e / QS_NB;
        s_tri[ti][tj] = (ti < nb && tj < nb && ti <= tj) ? F[(kb + ti) + (long long)lc[kb + tj] * ld] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        // lane r owns row r of the triangle (r < nb): x_j for j = nb-1 .. 0
        double t = (tid < nb) ? acc[kb + tid] : 0.0;
        for (int j = nb - 1; j >= 0; j--) {
            const double aj = __shfl(t, j, 64);
            const double xj = aj / s_tri[j][j];
            if (tid < j) t -= s_tri[tid][j] * xj;
            if (tid == j) s_x[j] = xj;
        }
    }
    __syncthreads();
    if (sl == 0 && tid < nb) X[s.col1 + lc[kb + tid]] = s_x[tid];
    if (i < kb) {
#pragma unroll
        for (int j = 0; j < QS_NB; j++) a -= ((j < nb) ? rv[j] : 0.0) * s_x[min(j, nb - 1)];
        acc[i] = a;
    }
}

__global__ __launch_bounds__(256) void k_perm(const double *__restrict__ in, const int *__restrict__ perm, double *out, int n,
                                               int scatter, long long sin, long long sout)
{
    in += (long long)blockIdx.y * sin; out += (long long)blockIdx.y * sout;      // (vector blockIdx.y of a batch)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = perm ? perm[i] : i;
    if (scatter) out[j] = in[i];
    else out[i] = in[j];
}
// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
int stm_launch_qapply(const DevCtx &c, const int *flist, int nfr, int method, double *W, int lds_bytes, int *err, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_qapply, dim3(nfr), dim3(QA_NT), (size_t)lds_bytes, st, c, flist, method, W, err);
    return (int)hipGetLastError();
}
int stm_launch_qapply_t(const DevCtx &c, const int *flist, int nfr, int method, double *W, int lds_bytes, hipStream_t st, int nb,
                        const RhsBatch &B)
{
    if (nfr <= 0) return 0;
    // (lds_bytes: what ONE right-hand side needs; R of them share a workgroup while R x that stays within 64 KB)
    if (nb >= 3 && 4L * lds_bytes <= 65536)
        hipLaunchKernelGGL(k_qapply_t<4>, dim3(nfr, (nb + 3) / 4), dim3(QA_NT), (size_t)4 * lds_bytes, st, c, flist, method, W, B, nb);
    else if (nb >= 2 && 2L * lds_bytes <= 131072)
        hipLaunchKernelGGL(k_qapply_t<2>, dim3(nfr, (nb + 1) / 2), dim3(QA_NT), (size_t)2 * lds_bytes, st, c, flist, method, W, B, nb);
    else
        hipLaunchKernelGGL(k_qapply_t<1>, dim3(nfr, nb), dim3(QA_NT), (size_t)lds_bytes, st, c, flist, method, W, B, nb);
    return (int)hipGetLastError();
}
// the split fronts of one level: prep, max(npanels) + 1 steps, finish
int stm_launch_qapply_big(const DevCtx &c, const QbDesc *qd, int nq, int max_npanels, int max_nslab, int max_fm, int method, double *W,
                          double *Xf, int *Dq, double *Wq, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nq <= 0 || max_npanels <= 0) return 0;
    hipLaunchKernelGGL(k_qbig_prep, dim3(nq, nb), dim3(QA_NT), 0, st, c, qd, (const double *)W, Xf, Dq, B);
    for (int k = 0; k <= max_npanels; k++)
        hipLaunchKernelGGL(k_qbig_step, dim3(max_nslab, nq, nb), dim3(QB_ROWS), 0, st, c, qd, k, method, Xf, (const int *)Dq, Wq, B);
    hipLaunchKernelGGL(k_qbig_finish, dim3((max_fm + 255) / 256, nq, nb), dim3(256), 0, st, c, qd, W, (const double *)Xf, B);
    return (int)hipGetLastError();
}
int stm_qt4_doubles(void) { return QT4_DOUBLES; }
// T4 of every group of every split front (items) -- once per factorization
int stm_launch_qt4_build(const DevCtx &c, const int *fl, const long long *dqo, int nfronts, const void *items, int nitems, int *Dq4, double *T4all,
                         hipStream_t st)
{
    if (nfronts <= 0 || nitems <= 0) return 0;
    hipLaunchKernelGGL(k_qt4_number, dim3(nfronts), dim3(QA_NT), 0, st, c, fl, dqo, Dq4);
    const size_t lds = sizeof(double) * (size_t)((QGN * QT4_VS > 5 * 32 * 33) ? QGN * QT4_VS : 5 * 32 * 33);
    hipLaunchKernelGGL(k_qt4_build, dim3(nitems), dim3(512), lds, st, c, (const Qt4Item *)items, (const int *)Dq4, T4all);
    return (int)hipGetLastError();
}
int stm_launch_qapply_big4(const DevCtx &c, const QbDesc *qd, const long long *t4off, int nq, int max_npanels, int max_nslab, int max_fm,
                           int method, double *W, double *Xf, int *Dq, double *Wq4, const double *T4all, hipStream_t st, int nb,
                           const RhsBatch &B)
{
    if (nq <= 0 || max_npanels <= 0) return 0;
    const int max_ng = (max_npanels + QG - 1) / QG;
    hipLaunchKernelGGL(k_qbig_prep, dim3(nq, nb), dim3(QA_NT), 0, st, c, qd, (const double *)W, Xf, Dq, B);
    for (int k = 0; k <= max_ng; k++) {
        if (nb >= 3) hipLaunchKernelGGL(k_qbig_step4<4>, dim3(max_nslab, nq, (nb + 3) / 4), dim3(QB_ROWS), 0, st, c, qd, t4off, k, method, Xf, (const int *)Dq, Wq4, T4all, B, nb);
        else if (nb == 2) hipLaunchKernelGGL(k_qbig_step4<2>, dim3(max_nslab, nq, 1), dim3(QB_ROWS), 0, st, c, qd, t4off, k, method, Xf, (const int *)Dq, Wq4, T4all, B, nb);
        else hipLaunchKernelGGL(k_qbig_step4<1>, dim3(max_nslab, nq, 1), dim3(QB_ROWS), 0, st, c, qd, t4off, k, method, Xf, (const int *)Dq, Wq4, T4all, B, nb);
    }
    hipLaunchKernelGGL(k_qbig_finish, dim3((max_fm + 255) / 256, nq, nb), dim3(256), 0, st, c, qd, W, (const double *)Xf, B);
    return (int)hipGetLastError();
}
int stm_launch_rsolve(const DevCtx &c, const int *flist, int nfr, const int *Rj, const double *W, double *X, int lds_bytes,
                      int *err, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nfr <= 0) return 0;
    // (lds_bytes: what ONE right-hand side needs)
    if (nb >= 3 && 4L * lds_bytes <= 65536)
        hipLaunchKernelGGL(k_rsolve<4>, dim3(nfr, (nb + 3) / 4), dim3(RS_NT), (size_t)4 * lds_bytes, st, c, flist, Rj, W, X, err, B, nb);
    else if (nb >= 2 && 2L * lds_bytes <= 131072)
        hipLaunchKernelGGL(k_rsolve<2>, dim3(nfr, (nb + 1) / 2), dim3(RS_NT), (size_t)2 * lds_bytes, st, c, flist, Rj, W, X, err, B, nb);
    else
        hipLaunchKernelGGL(k_rsolve<1>, dim3(nfr, nb), dim3(RS_NT), (size_t)lds_bytes, st, c, flist, Rj, W, X, err, B, nb);
    return (int)hipGetLastError();
}
// back substitution of the split fronts of one level: prep, init, max ceil(fp / 32) steps
int stm_launch_rsolve_big(const DevCtx &c, const QbDesc *qd, int nq, int max_steps, int max_nslab, const int *Rj, const double *W,
                          double *X, double *Acc, int *Lc, int *Rm, int *err, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nq <= 0) return 0;
    hipLaunchKernelGGL(k_rbig_prep, dim3(nq, nb), dim3(RS_NT), 0, st, c, qd, X, Lc, Rm, err, B);
    if (nb >= 3)
        hipLaunchKernelGGL(k_rbig_init<4>, dim3(max_nslab * (STM_QB_ROWS / 64), nq, (nb + 3) / 4), dim3(STM_QB_ROWS), 0, st, c, qd, Rj, W,
                           (const double *)X, Acc, (const int *)Rm, B, nb);
    else if (nb == 2)
        hipLaunchKernelGGL(k_rbig_init<2>, dim3(max_nslab * (STM_QB_ROWS / 64), nq, 1), dim3(STM_QB_ROWS), 0, st, c, qd, Rj, W, (const double *)X, Acc,
                           (const int *)Rm, B, nb);
    else
        hipLaunchKernelGGL(k_rbig_init<1>, dim3(max_nslab * (STM_QB_ROWS / 64), nq, 1), dim3(STM_QB_ROWS), 0, st, c, qd, Rj, W, (const double *)X, Acc,
                           (const int *)Rm, B, nb);
    for (int k = 0; k < max_steps; k++)
        hipLaunchKernelGGL(k_rbig_step, dim3(max_nslab, nq, nb), dim3(STM_QB_ROWS), 0, st, c, qd, k, X, Acc, (const int *)Lc, (const int *)Rm, B);
    return (int)hipGetLastError();
}
int stm_launch_rtsolve(const DevCtx &c, const int *flist, int nfr, const double *Bp, double *U, double *Xr, const int *rowbase,
                       int lds_bytes, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rtsolve, dim3(nfr, nb), dim3(RS_NT), (size_t)lds_bytes, st, c, flist, Bp, U, Xr, rowbase, B);
    return (int)hipGetLastError();
}
int stm_launch_perm(const double *in, const int *perm, double *out, int n, int scatter, hipStream_t st, int nb, long long sin, long long sout)
{
    if (n <= 0 || nb <= 0) return 0;
    hipLaunchKernelGGL(k_perm, dim3((n + 255) / 256, nb), dim3(256), 0, st, in, perm, out, n, scatter, sin, sout);
    return (int)hipGetLastError();
}
int stm_configure_resident(void)
{
    CK(hipFuncSetAttribute((const void *)k_qapply, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_qapply_t<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_qapply_t<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_qapply_t<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_rsolve<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_rsolve<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_rsolve<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_rtsolve, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    return 0;
}
