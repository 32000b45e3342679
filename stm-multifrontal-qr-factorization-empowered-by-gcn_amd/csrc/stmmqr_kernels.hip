// stmmqr_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the multifrontal-QR numeric phase.
//
// One file = the whole device side of the reference's hot loop (STMMQR/src/qr/SparseQR_factorize.c):
//   k_gather_sx      <- qr_stranspose2          (:755-785)   S = A(P,Q) values, pure gather
//   k_setup          <- qr_fsize + the integer half of qr_assemble (:1066-1145, :1239-1248, :1205)
//   k_assemble       <- qr_assemble             (:1151-1285) scatter of S rows and packed child C blocks
//   dev_panel        <- qr_front's column loop  (:1434-1609) + dlarft (T factor of the block reflector)
//   dev_update_block <- qr_larftb / dlarfb      (:1851-1904) C -= V (T' (V' C)) on v_mfma_f64_16x16x4_f64
//   k_front_wg       one workgroup factorizes a whole (small) front; k_panel / k_update: large fronts
//   dev_cpack        <- qr_cpack                (:1639-1685)
//   k_rh_count / k_rh_scan / k_rh_copy <- qr_rhpack (:1691-1784) + the stack compaction of qr_factorize (:597-701)
//
// Design notes (DESIGN.md has the long form):
//  * fronts are column-major with a fixed leading dimension; a front's rows are known only on the device
//    (dead pivot columns change them), so every kernel reads FrontNum for its extents;
//  * the tall-skinny panel is staged in LDS when it fits (<= lds_doubles), Householder norms and v'C dot
//    products use wave64 shuffles + one cross-wave LDS step;
//  * the block reflector is applied with fp64 MFMA tiles (16x16x4): W = V'C over 64-row chunks, W = T'W,
//    C -= V W; V is read in place from F with the unit-diagonal/zero mask applied on the fly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stmmqr_device.h"
#include "stmmqr_kernels.h"

typedef double d4 __attribute__((ext_vector_type(4)));

#define NT 256
#define NW (NT / 64)
#define BN 32                 // trailing-update column block
#define RB 64                 // trailing-update row chunk
#define VS (RB + 2)           // LDS row stride of the V / C chunk images (doubles)
#define WS (BN + 1)

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the whole workgroup; every thread gets the result.  s_red: NW doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    v = wave_sum(v);
    __syncthreads();                       // protect s_red from the previous use
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) r += s_red[w];
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

// ------------------------------------------------------------------------------------------------
// qr_stranspose2: Sx[s] = Ax[smap[s]]   (smap is symbolic: planner, from Ap/Ai/Qfill/PLinv/Sp)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_sx(const double *__restrict__ Ax, const int *__restrict__ smap,
                                                   double *__restrict__ Sx, int anz)
{
    for (int s = blockIdx.x * 256 + threadIdx.x; s < anz; s += gridDim.x * 256) Sx[s] = Ax[smap[s]];
}

// ------------------------------------------------------------------------------------------------
// qr_fsize + row bookkeeping of qr_assemble.  One workgroup per front of the level.
//   Stair[j]  <- one past the last row whose leftmost column is <= j   (the staircase qr_front consumes)
//   Cmap[..]  <- row of the parent that receives row ci of child c      (qr_assemble :1239-1248)
//   Hii[..]   <- S-row ids of the front's rows                          (:1205, :1246)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_setup(DevCtx c, const int *__restrict__ flist)
{
    __shared__ int s_scan[NW];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int tid = threadIdx.x;
    int *St = c.Stair + s.rp;
    int *Cur = c.Cursor + s.rp;
    const int *Sl = c.Sleft + s.col1;

    for (int j = tid; j < s.fn; j += NT) St[j] = (j < s.fp) ? Sl[j + 1] - Sl[j] : 0;
    __syncthreads();
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const int cm = c.fnum[ch].cm;
        const int pc = c.fs[ch].rp + c.fs[ch].fp;
        for (int ci = tid; ci < cm; ci += NT) atomicAdd(&St[c.Rjrel[pc + ci]], 1);
    }
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < s.fn; base += NT) {
        const int j = base + tid;
        // the counts were built with L2 atomics: read them past the (possibly stale) vector L1
        const int v = (j < s.fn) ? __hip_atomic_load(&St[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int tot;
        const int incl = block_incl_scan(v, s_scan, &tot);
        if (j < s.fn) {
            St[j] = carry + incl;
            Cur[j] = carry + incl - v + ((j < s.fp) ? Sl[j + 1] - Sl[j] : 0);
        }
        carry += tot;
    }
    __syncthreads();
    const int fm = carry;
    int *Hi = c.Hii + s.hip;
    for (int r = s.srow0 + tid; r < s.srow1; r += NT) {
        const int k = c.Sj0[r] - s.col1;
        const int i = (k > 0 ? St[k - 1] : 0) + (r - Sl[k]);
        Hi[i] = r;
    }
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const int cm = c.fnum[ch].cm;
        const int pc = c.fs[ch].rp + c.fs[ch].fp;
        const int *Hic = c.Hii + c.fs[ch].hip + c.fnum[ch].rank;
        for (int ci = tid; ci < cm; ci += NT) {
            const int j = c.Rjrel[pc + ci];      // distinct for distinct ci of one child: no race
            const int i = Cur[j];
            Cur[j] = i + 1;
            c.Cmap[pc + ci] = i;
            Hi[i] = Hic[ci];
        }
        __syncthreads();
    }
    if (tid == 0) {
        FrontNum *nm = &c.fnum[f];
        nm->fm = fm; nm->g = 0; nm->rank = min(fm, s.fp); nm->done = 0;
        nm->pg1 = 0; nm->pt = 0; nm->pk1 = 0; nm->pnb = 0; nm->pc0 = 0; nm->cm = 0; nm->rsize = 0; nm->flops = 0; nm->flops_upd = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// qr_assemble: scatter S rows and the children's packed C blocks into the (pre-zeroed) front.
// grid = (max parts, fronts of the level); part p of front f handles every nparts-th 256-element slab.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_assemble(const DevCtx &c, const FrontSym &s, int part, int nparts)
{
    const int tid = threadIdx.x;
    double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int *St = c.Stair + s.rp;
    const int *Sl = c.Sleft + s.col1;
    for (int r = s.srow0 + part * NT + tid; r < s.srow1; r += nparts * NT) {
        const int k = c.Sj0[r] - s.col1;
        const int i = (k > 0 ? St[k - 1] : 0) + (r - Sl[k]);
        for (int p = c.Sp[r]; p < c.Sp[r + 1]; p++) F[i + c.Sjrel[p] * ld] = c.Sx[p];
    }
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const int cm = c.fnum[ch].cm;
        if (cm <= 0) continue;
        const FrontSym cs = c.fs[ch];
        const int cn = cs.fn - cs.fp;
        const int pc = cs.rp + cs.fp;
        const long long tri = (long long)cm * (cm + 1) / 2;
        const long long csize = tri + (long long)cm * (cn - cm);
        const double *C = c.Carena + cs.coff;
        const int *cmap = c.Cmap + pc;
        const int *crel = c.Rjrel + pc;
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
            F[cmap[ci] + crel[cj] * ld] = C[e];
        }
    }
}

__global__ __launch_bounds__(NT) void k_assemble(DevCtx c, const int *__restrict__ flist,
                                                 const int *__restrict__ nparts_list)
{
    const int fi = blockIdx.y;
    const int nparts = nparts_list[fi];
    if ((int)blockIdx.x >= nparts) return;
    const FrontSym s = c.fs[flist[fi]];
    dev_assemble(c, s, blockIdx.x, nparts);
}

// ------------------------------------------------------------------------------------------------
// dlarft('F','C'): T (NB x NB upper triangular, column-major, zero padded) of the reflectors stored in the
// nbp columns of P (column stride pst, row 0 of P = front row g1).  s_diag[j] = front row of the unit
// diagonal of reflector j (BIGROW / tau 0: no reflector); rows >= tlast are structurally zero.
// T(0:b-1,b) = -tau_b T(0:b-1,0:b-1) (V(:,0:b-1)' v_b)   
// ------------------------------------------------------------------------------------------------
__device__ void dev_larft(const double *P, long long pst, int g1, int tlast, int nbp, const int *s_diag,
                          const double *s_tau, double (*s_G)[STM_NB + 1], double (*s_T)[STM_NB + 1], double *Tout)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int npairs = nbp * (nbp - 1) / 2;
    for (int idx = wid; idx < npairs; idx += NW) {
        int b = (int)((1.0 + sqrt(1.0 + 8.0 * idx)) * 0.5);
        while (b * (b - 1) / 2 > idx) b--;
        while ((b + 1) * b / 2 <= idx) b++;
        const int a = idx - b * (b - 1) / 2;
        double sum = 0;
        if (s_tau[a] != 0.0 && s_tau[b] != 0.0) {
            const int db = s_diag[b];                   // > s_diag[a]
            const double *va = P + a * pst - g1, *vb = P + b * pst - g1;
            sum = (lane == 0) ? va[db] : 0.0;           // v_b(db) = 1
            for (int i = db + 1 + lane; i < tlast; i += 64) sum += va[i] * vb[i];
            sum = wave_sum(sum);
        }
        if (lane == 0) s_G[a][b] = sum;
    }
    __syncthreads();
    if (wid == 0) {
        // lane a owns row a of T
        const int a = lane;
        for (int b = 0; b < nbp; b++) {
            if (a < STM_NB) {
                double v = 0;
                const double tb = s_tau[b];
                if (a < b && tb != 0.0) {
                    for (int l = a; l < b; l++) v += s_T[a][l] * s_G[l][b];
                    v *= -tb;
                } else if (a == b) v = tb;
                s_T[a][b] = v;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < STM_NB * STM_NB; e += NT) {
        const int a = e % STM_NB, b = e / STM_NB;
        Tout[e] = (a < nbp && b < nbp) ? s_T[a][b] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// qr_front, one panel of <= STM_NB columns (reference: the column loop :1434-1609 with the panel policy
// fixed to k1 = p*NB, which changes rounding only).  Executed by one whole workgroup.
// The panel F(g1:tmax, k1:k2) is staged in LDS when it fits; otherwise it is worked on in place (L2).
// Produces: R and V in F, Tau, Stair, Rdead, the T factor (Tout, NB x NB, column-major) and the pending
// block-reflector description in FrontNum (pg1, pt, pk1, pnb, pdiag).
// ------------------------------------------------------------------------------------------------
__device__ void dev_panel(const FrontSym &s, FrontNum *num, double *F, int *St, double *Tau, char *Rdead,
                          int p, double tol, int ntol_global, double *Tout, double *lds, int lds_doubles)
{
    __shared__ double s_red[NW];
    __shared__ int s_diag[STM_NB];
    __shared__ double s_tau[STM_NB];
    __shared__ double s_G[STM_NB][STM_NB + 1];
    __shared__ double s_T[STM_NB][STM_NB + 1];

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
        if (tid == 0) num->pnb = 0;
        return;
    }
    const int ntol = min(ntol_global - s.col1, npiv);
    const int g1 = g;
    int tmax = min(m, max(St[k2 - 1], g1 + nbp));
    const int mp = tmax - g1;
    double flops = 0;
    int done = 0, tlast = g1;

    double *P;
    long long pst;
    const bool in_lds = (mp > 0) && ((long long)(mp | 1) * nbp <= (long long)lds_doubles);
    if (in_lds) {
        pst = mp | 1;
        P = lds;
        for (int j = wid; j < nbp; j += NW) {
            const double *src = F + g1 + (k1 + j) * ld;
            for (int i = lane; i < mp; i += 64) P[i + j * pst] = src[i];
        }
    } else {
        pst = ld;
        P = F + g1 + k1 * ld;
    }
    __syncthreads();

    for (int j = 0; j < nbp; j++) {
        const int k = k1 + j;
        if (g >= m) {
            // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
            for (int kk = k + tid; kk < n; kk += NT) {
                if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
                else St[kk] = m;
                Tau[kk] = 0;
            }
            for (int jj = j + tid; jj < nbp; jj += NT) { s_diag[jj] = STM_BIGROW; s_tau[jj] = 0; }
            done = 1;
            break;
        }
        const int t = max(g + 1, St[k]);
        double *col = P + (g - g1) + j * pst;          // col[0] = F(g,k)
        const int len = t - g;                          // >= 1
        // ---- dlarfg ----
        double ss = 0;
        for (int i = 1 + tid; i < len; i += NT) { const double x = col[i]; ss += x * x; }
        ss = block_sum(ss, s_red);
        const double alpha = col[0];
        double tau = 0, beta = alpha, scal = 0;
        if (len > 1 && ss != 0.0) {
            const double xnorm = sqrt(ss);
            beta = -copysign(hypot(alpha, xnorm), alpha);
            tau = (beta - alpha) / beta;
            scal = 1.0 / (alpha - beta);
        }
        const bool dead = (k < ntol) && (fabs(beta) <= tol);
        if (dead) {
            // zero the column from the diagonal down, no reflector, g does not advance (:1495-1544)
            for (int i = tid; i < tmax - g; i += NT) col[i] = 0;
            if (tid == 0) { St[k] = 0; Tau[k] = 0; Rdead[k] = 1; s_diag[j] = STM_BIGROW; s_tau[j] = 0; }
            if (k == npiv - 1) rank = g;                // (:1604-1608) also taken on a dead last pivot
            __syncthreads();
            continue;
        }
        if (tid == 0) { St[k] = t; Tau[k] = tau; col[0] = beta; s_diag[j] = g; s_tau[j] = tau; }
        if (tau != 0.0)
            for (int i = 1 + tid; i < len; i += NT) col[i] *= scal;
        flops += (double)len * (3.0 + 4.0 * (double)(n - k - 1));
        __syncthreads();
        // ---- dlarf on the rest of the panel: one wave per column, shuffles for v'c ----
        if (tau != 0.0) {
            for (int jj = j + 1 + wid; jj < nbp; jj += NW) {
                double *cc = P + (g - g1) + jj * pst;
                double w = (lane == 0) ? cc[0] : 0.0;
                for (int i = 1 + lane; i < len; i += 64) w += col[i] * cc[i];
                w = wave_sum(w) * tau;
                if (lane == 0) cc[0] -= w;
                for (int i = 1 + lane; i < len; i += 64) cc[i] -= w * col[i];
            }
        }
        tlast = t;
        g++;
        if (k == npiv - 1) rank = g;
        __syncthreads();
    }
    __syncthreads();

    dev_larft(P, pst, g1, tlast, nbp, s_diag, s_tau, s_G, s_T, Tout);
    if (in_lds) {
        for (int j = wid; j < nbp; j += NW) {
            double *dst = F + g1 + (k1 + j) * ld;
            for (int i = lane; i < mp; i += 64) dst[i] = P[i + j * pst];
        }
    }
    if (tid < STM_NB) num->pdiag[tid] = (tid < nbp) ? s_diag[tid] : STM_BIGROW;
    if (tid == 0) {
        num->g = g; num->rank = rank; num->done = done;
        num->pg1 = g1; num->pt = tlast; num->pk1 = k1; num->pnb = nbp; num->pc0 = k2;
        num->flops += flops;
        {
            int nlive = 0;
            for (int j = 0; j < nbp; j++) nlive += (s_tau[j] != 0.0);
            num->flops_upd += 4.0 * (double)(tlast - g1) * (double)(n - k2) * (double)nlive;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// qr_larftb(QR_QTX): C <- (I - V T V')' C for one BN-column block of the trailing matrix, on fp64 MFMA.
//   C = F(pg1:pt, c0:c0+BN), V = F(pg1:pt, pk1:pk1+pnb) (unit diagonal at pdiag[], zero above),
//   T = NB x NB upper triangular.  lds: >= 2*BN*VS + STM_NB*WS doubles.
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[i=(l>>4)+4r][j=l&15].
// ------------------------------------------------------------------------------------------------
__device__ void dev_update_block(const FrontSym &s, const FrontNum *num, double *F, const double *T, int cb,
                                 double *lds)
{
    const int g1 = num->pg1, mp = num->pt - num->pg1, k1 = num->pk1, nbp = num->pnb;
    if (nbp <= 0 || mp <= 0) return;
    const int c0 = num->pc0 + cb * BN;
    if (c0 >= s.fn) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    double *Vs = lds;                       // [STM_NB][VS]
    double *Cs = Vs + STM_NB * VS;          // [BN][VS]
    double *Ws = Cs + BN * VS;              // [STM_NB][WS]
    __shared__ int s_pd[STM_NB];

    __syncthreads();                        // previous users of lds are done
    if (tid < STM_NB) s_pd[tid] = num->pdiag[tid];
    __syncthreads();

    const int lrow = tid & 63, lcg = tid >> 6;          // loader mapping: 64 rows x 4 column groups of 8
    const double *Vg = F + g1 + (long long)k1 * ld;
    double *Cg = F + g1 + (long long)c0 * ld;

    // ---- phase 1: W1 = V' C ----
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc = {0, 0, 0, 0};
    for (int r0 = 0; r0 < mp; r0 += RB) {
        const int i = r0 + lrow;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int col = lcg * 8 + q;
            double v = 0, cv = 0;
            if (i < mp) {
                if (col < nbp) {
                    const int d = s_pd[col] - g1;
                    v = (i < d) ? 0.0 : ((i == d) ? 1.0 : Vg[i + col * ld]);
                }
                if (col < nc) cv = Cg[i + col * ld];
            }
            Vs[col * VS + lrow] = v;
            Cs[col * VS + lrow] = cv;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk++) {
            const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
            const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; r++) Ws[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = acc[r];
    __syncthreads();

    // ---- phase 2: W2 = T' W1 (T upper triangular) ----
    {
        const int l = tid & 31, cg = tid >> 5;          // 8 groups x 4 columns
        double w2[4] = {0, 0, 0, 0};
        for (int q = 0; q <= l; q++) {
            const double tq = T[q + l * STM_NB];
#pragma unroll
            for (int x = 0; x < 4; x++) w2[x] += tq * Ws[q * WS + cg * 4 + x];
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 4; x++) Ws[l * WS + cg * 4 + x] = w2[x];
    }
    __syncthreads();

    // ---- phase 3: C -= V W2 ----
    for (int r0 = 0; r0 < mp; r0 += RB) {
        const int i = r0 + lrow;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int col = lcg * 8 + q;
            double v = 0, cv = 0;
            if (i < mp) {
                if (col < nbp) {
                    const int d = s_pd[col] - g1;
                    v = (i < d) ? 0.0 : ((i == d) ? 1.0 : Vg[i + col * ld]);
                }
                if (col < nc) cv = Cg[i + col * ld];
            }
            Vs[col * VS + lrow] = v;
            Cs[col * VS + lrow] = cv;
        }
        __syncthreads();
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

// ------------------------------------------------------------------------------------------------
// small fronts: one workgroup runs the whole front (all panels, all trailing updates, C pack)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_front_wg(DevCtx c, const int *__restrict__ flist, int lds_doubles)
{
    extern __shared__ double dyn_lds[];
    __shared__ double s_Tw[STM_NB * STM_NB];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    FrontNum *num = &c.fnum[f];
    double *F = c.Farena + s.foff;
    for (int p = 0; p < s.npanels; p++) {
        dev_panel(s, num, F, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, s_Tw, dyn_lds,
                  lds_doubles);
        const int k2 = min(s.fn, (p + 1) * STM_NB);
        const int ncb = (s.fn - k2 + BN - 1) / BN;
        for (int cb = 0; cb < ncb; cb++) dev_update_block(s, num, F, s_Tw, cb, dyn_lds);
        __syncthreads();
    }
    dev_cpack(c, s, num, 0, 1);
}

// ------------------------------------------------------------------------------------------------
// large fronts: panel and trailing update are separate launches (many workgroups per update)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_panel(DevCtx c, const int *__restrict__ flist, int p, int lds_doubles)
{
    extern __shared__ double dyn_lds[];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    dev_panel(s, &c.fnum[f], c.Farena + s.foff, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol,
              c.Tws + (long long)c.tslot[f] * STM_NB * STM_NB, dyn_lds, lds_doubles);
}

__global__ __launch_bounds__(NT) void k_update(DevCtx c, const int *__restrict__ flist, int p)
{
    extern __shared__ double dyn_lds[];
    const int f = flist[blockIdx.y];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    dev_update_block(s, &c.fnum[f], c.Farena + s.foff, c.Tws + (long long)c.tslot[f] * STM_NB * STM_NB, blockIdx.x,
                     dyn_lds);
}

// standalone T factor of the pending block reflector described by FrontNum (qr_larftb seam)
__global__ __launch_bounds__(NT) void k_larft(DevCtx c, int f)
{
    __shared__ int s_diag[STM_NB];
    __shared__ double s_tau[STM_NB];
    __shared__ double s_G[STM_NB][STM_NB + 1];
    __shared__ double s_T[STM_NB][STM_NB + 1];
    const FrontSym s = c.fs[f];
    const FrontNum *num = &c.fnum[f];
    const int tid = threadIdx.x;
    if (tid < STM_NB) {
        s_diag[tid] = num->pdiag[tid];
        s_tau[tid] = (tid < num->pnb && num->pdiag[tid] != STM_BIGROW) ? c.Tau[s.rp + num->pk1 + tid] : 0.0;
    }
    __syncthreads();
    const double *P = c.Farena + s.foff + num->pg1 + (long long)num->pk1 * s.ld;
    dev_larft(P, s.ld, num->pg1, num->pt, num->pnb, s_diag, s_tau, s_G, s_T,
              c.Tws + (long long)c.tslot[f] * STM_NB * STM_NB);
}

__global__ __launch_bounds__(NT) void k_cpack(DevCtx c, const int *__restrict__ flist,
                                              const int *__restrict__ nparts_list)
{
    const int fi = blockIdx.y;
    const int nparts = nparts_list[fi];
    if ((int)blockIdx.x >= nparts) return;
    const int f = flist[fi];
    const FrontSym s = c.fs[f];
    dev_cpack(c, s, &c.fnum[f], blockIdx.x, nparts);
}

// ------------------------------------------------------------------------------------------------
// qr_rhpack, split in three: per-column lengths + offsets, offsets of the blocks (Post order = the
// reference's single shrunk stack), coalesced copy.  Layout: SURVEY.md A.6.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_rh_count(DevCtx c, const int *__restrict__ flist)
{
    __shared__ int s_scan[NW];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    FrontNum *num = &c.fnum[f];
    const int tid = threadIdx.x;
    const int fm = num->fm, n = s.fn, fp = s.fp;
    const int *St = c.Stair + s.rp;
    int *off = c.Rhoff + s.rp;
    if (fm <= 0 || n <= 0) {
        for (int k = tid; k < n; k += NT) off[k] = 0;
        if (tid == 0) num->rsize = 0;
        return;
    }
    // pass 1: rm(k) = live pivots among columns 0..k (stored temporarily in off[])
    int carry = 0;
    for (int base = 0; base < fp; base += NT) {
        const int k = base + tid;
        const int live = (k < fp && St[k] != 0) ? 1 : 0;
        int tot;
        const int incl = block_incl_scan(live, s_scan, &tot);
        if (k < fp) off[k] = carry + incl;
        carry += tot;
    }
    __syncthreads();
    const int rm = carry;
    // pass 2: column lengths -> exclusive offsets
    carry = 0;
    for (int base = 0; base < n; base += NT) {
        const int k = base + tid;
        int len = 0;
        if (k < fp) {
            const int t = St[k];
            len = (t == 0) ? off[k] : t;       // dead: rm so far (off[k] excludes k itself since live=0)
        } else if (k < n) {
            const int h = min(rm + (k - fp) + 1, fm);
            len = rm + max(St[k] - h, 0);
        }
        __syncthreads();
        int tot;
        const int incl = block_incl_scan(len, s_scan, &tot);
        if (k < n) off[k] = carry + incl - len;
        carry += tot;
    }
    if (tid == 0) num->rsize = carry;
}

// single workgroup: Rboff[f] = offset of front f's block in Post order; total in *rh_total
__global__ __launch_bounds__(NT) void k_rh_scan(DevCtx c, const int *__restrict__ post, int nf, long long *rh_total)
{
    __shared__ long long s_part[NT];
    const int tid = threadIdx.x;
    const int per = (nf + NT - 1) / NT;
    const int a = tid * per, b = min(nf, a + per);
    long long sum = 0;
    for (int q = a; q < b; q++) sum += c.fnum[post[q]].rsize;
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int t = 0; t < NT; t++) { const long long v = s_part[t]; s_part[t] = run; run += v; }
        *rh_total = run;
    }
    __syncthreads();
    long long run = s_part[tid];
    for (int q = a; q < b; q++) {
        const int f = post[q];
        c.Rboff[f] = run;
        run += c.fnum[f].rsize;
    }
}

__global__ __launch_bounds__(NT) void k_rh_copy(DevCtx c, const int *__restrict__ flist,
                                                const int *__restrict__ nparts_list, double *__restrict__ RH)
{
    const int fi = blockIdx.y;
    const int nparts = nparts_list[fi];
    if ((int)blockIdx.x >= nparts) return;
    const int f = flist[fi];
    const FrontSym s = c.fs[f];
    const FrontNum *num = &c.fnum[f];
    const int fm = num->fm, n = s.fn, fp = s.fp, rm = num->rank;
    if (fm <= 0 || n <= 0) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long ld = s.ld;
    const double *F = c.Farena + s.foff;
    const int *St = c.Stair + s.rp;
    const int *off = c.Rhoff + s.rp;
    double *R = RH + c.Rboff[f];
    // one wave per column
    for (int k = blockIdx.x * NW + wid; k < n; k += nparts * NW) {
        const double *Fk = F + k * ld;
        double *Rk = R + off[k];
        if (k < fp) {
            const int len = ((k + 1 < n) ? off[k + 1] : num->rsize) - off[k];
            for (int i = lane; i < len; i += 64) Rk[i] = Fk[i];
        } else {
            const int h = min(rm + (k - fp) + 1, fm);
            const int t = St[k];
            for (int i = lane; i < rm; i += 64) Rk[i] = Fk[i];
            for (int i = h + lane; i < t; i += 64) Rk[rm + i - h] = Fk[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

int stm_launch_gather_sx(const double *Ax, const int *smap, double *Sx, int anz, hipStream_t st)
{
    if (anz <= 0) return 0;
    int grid = (anz + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_gather_sx, dim3(grid), dim3(256), 0, st, Ax, smap, Sx, anz);
    return (int)hipGetLastError();
}
int stm_launch_setup(const DevCtx &c, const int *flist, int nfr, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_setup, dim3(nfr), dim3(NT), 0, st, c, flist);
    return (int)hipGetLastError();
}
int stm_launch_assemble(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_assemble, dim3(maxparts, nfr), dim3(NT), 0, st, c, flist, nparts);
    return (int)hipGetLastError();
}
int stm_update_lds_bytes(void) { return (int)((2 * BN * VS + STM_NB * WS) * sizeof(double)); }
int stm_launch_front_wg(const DevCtx &c, const int *flist, int nfr, int lds_doubles, hipStream_t st)
{
    if (nfr <= 0) return 0;
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();
    hipLaunchKernelGGL(k_front_wg, dim3(nfr), dim3(NT), bytes, st, c, flist, (int)(bytes / sizeof(double)));
    return (int)hipGetLastError();
}
int stm_launch_panel(const DevCtx &c, const int *flist, int nfr, int p, int lds_doubles, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_panel, dim3(nfr), dim3(NT), (size_t)lds_doubles * sizeof(double), st, c, flist, p,
                       lds_doubles);
    return (int)hipGetLastError();
}
int stm_launch_update(const DevCtx &c, const int *flist, int nfr, int p, int maxcb, hipStream_t st)
{
    if (nfr <= 0 || maxcb <= 0) return 0;
    hipLaunchKernelGGL(k_update, dim3(maxcb, nfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, flist, p);
    return (int)hipGetLastError();
}
int stm_launch_larft(const DevCtx &c, int f, hipStream_t st)
{
    hipLaunchKernelGGL(k_larft, dim3(1), dim3(NT), 0, st, c, f);
    return (int)hipGetLastError();
}
int stm_launch_cpack(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_cpack, dim3(maxparts, nfr), dim3(NT), 0, st, c, flist, nparts);
    return (int)hipGetLastError();
}
int stm_launch_rh_count(const DevCtx &c, const int *flist, int nfr, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rh_count, dim3(nfr), dim3(NT), 0, st, c, flist);
    return (int)hipGetLastError();
}
int stm_launch_rh_scan(const DevCtx &c, const int *post, int nf, long long *rh_total, hipStream_t st)
{
    hipLaunchKernelGGL(k_rh_scan, dim3(1), dim3(NT), 0, st, c, post, nf, rh_total);
    return (int)hipGetLastError();
}
int stm_launch_rh_copy(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, double *RH,
                       hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rh_copy, dim3(maxparts, nfr), dim3(NT), 0, st, c, flist, nparts, RH);
    return (int)hipGetLastError();
}
int stm_configure_kernels(void)
{
    // allow the panel kernels to ask for up to 144 KiB of dynamic LDS (160 KiB per CU on gfx950)
    CK(hipFuncSetAttribute((const void *)k_front_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_panel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_update, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    return 0;
}
