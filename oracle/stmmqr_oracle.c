/* oracle/stmmqr_oracle.c -- TEST INFRASTRUCTURE ONLY (see stmmqr_oracle.h for scope and pinning).
 *
 * Plain-C restatement of the reference's numeric multifrontal QR path.  Every function names the
 * reference lines it follows (paths relative to /root/reference/STMMQR).  Nothing here is used by
 * the product: the HIP library never links or calls this file.
 */
#include "stmmqr_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define IMIN(a, b) ((a) < (b) ? (a) : (b))
#define IMAX(a, b) ((a) > (b) ? (a) : (b))

static double wall(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

/* ------------------------------------------------------------------------------------------------
 * LAPACK-semantics kernels (LAPACK 3.x DLARFG / DLARF / DLARFT / DLARFB; call sites
 * src/qr/SparseQR_factorize.c:1320,1352,1802,1842)
 * ---------------------------------------------------------------------------------------------- */

/* scaled 2-norm, the classic DNRM2 recurrence (overflow-safe) */
static double nrm2(orc_int n, const double *x)
{
    double scale = 0.0, ssq = 1.0;
    for (orc_int i = 0; i < n; i++) {
        if (x[i] != 0.0) {
            double a = fabs(x[i]);
            if (scale < a) { double r = scale / a; ssq = 1.0 + ssq * r * r; scale = a; }
            else           { double r = a / scale; ssq += r * r; }
        }
    }
    return scale * sqrt(ssq);
}

/* DLARFG: H = I - tau [1;v][1;v]', H [alpha;x] = [beta;0].  On return *alpha = beta, x = v. */
double orc_larfg(orc_int n, double *alpha, double *x)
{
    if (n <= 1) return 0.0;
    double xnorm = nrm2(n - 1, x);
    if (xnorm == 0.0) return 0.0;
    double a = *alpha;
    double beta = -copysign(hypot(a, xnorm), a);
    const double safmin = DBL_MIN / (DBL_EPSILON * 0.5);  /* dlamch('S')/dlamch('E') */
    const double rsafmn = 1.0 / safmin;
    int knt = 0;
    if (fabs(beta) < safmin) {
        do {
            knt++;
            for (orc_int i = 0; i < n - 1; i++) x[i] *= rsafmn;
            beta *= rsafmn;
            a *= rsafmn;
        } while (fabs(beta) < safmin && knt < 20);
        xnorm = nrm2(n - 1, x);
        beta = -copysign(hypot(a, xnorm), a);
    }
    double tau = (beta - a) / beta;
    double s = 1.0 / (a - beta);
    for (orc_int i = 0; i < n - 1; i++) x[i] *= s;
    for (int j = 0; j < knt; j++) beta *= safmin;
    *alpha = beta;
    return tau;
}

/* DLARF side='L': C(m x n) <- (I - tau v v') C, v has an explicit first entry */
void orc_larf_left(orc_int m, orc_int n, const double *v, double tau, double *C, orc_int ldc, double *work)
{
    if (tau == 0.0) return;
    for (orc_int j = 0; j < n; j++) {
        const double *c = C + j * ldc;
        double s = 0.0;
        for (orc_int i = 0; i < m; i++) s += c[i] * v[i];
        work[j] = s;
    }
    for (orc_int j = 0; j < n; j++) {
        double *c = C + j * ldc;
        double s = tau * work[j];
        for (orc_int i = 0; i < m; i++) c[i] -= v[i] * s;
    }
}

/* DLARFT direct='F' storev='C': T (k x k upper), V is n x k unit lower trapezoidal */
void orc_larft(orc_int n, orc_int k, const double *V, orc_int ldv, const double *tau, double *T, orc_int ldt)
{
    for (orc_int i = 0; i < k; i++) {
        if (tau[i] == 0.0) {
            for (orc_int j = 0; j <= i; j++) T[j + i * ldt] = 0.0;
            continue;
        }
        /* T(0:i-1,i) = -tau_i * V(i:n-1,0:i-1)' * V(i:n-1,i), V(i,i) = 1 implicit */
        for (orc_int j = 0; j < i; j++) {
            double s = V[i + j * ldv];
            for (orc_int r = i + 1; r < n; r++) s += V[r + j * ldv] * V[r + i * ldv];
            T[j + i * ldt] = -tau[i] * s;
        }
        /* T(0:i-1,i) = T(0:i-1,0:i-1) * T(0:i-1,i)   (upper-triangular matvec, in place) */
        for (orc_int j = 0; j < i; j++) {
            double s = 0.0;
            for (orc_int l = j; l < i; l++) s += T[j + l * ldt] * T[l + i * ldt];
            T[j + i * ldt] = s;
        }
        T[i + i * ldt] = tau[i];
    }
}

/* DLARFB side='L', direct='F', storev='C'.  trans: 'T' -> C <- H' C, 'N' -> C <- H C.
 * work is n x k with leading dimension ldw. */
static void larfb_left(char trans, orc_int m, orc_int n, orc_int k, const double *V, orc_int ldv,
                       const double *T, orc_int ldt, double *C, orc_int ldc, double *W, orc_int ldw)
{
    /* W = C' V  (V unit lower trapezoidal, strict upper part of the V block ignored) */
    for (orc_int l = 0; l < k; l++) {
        for (orc_int j = 0; j < n; j++) {
            const double *c = C + j * ldc;
            double s = c[l];
            for (orc_int r = l + 1; r < m; r++) s += c[r] * V[r + l * ldv];
            W[j + l * ldw] = s;
        }
    }
    /* W = W * T (trans 'T') or W * T' (trans 'N') */
    if (trans == 'T') {
        for (orc_int l = k - 1; l >= 0; l--)
            for (orc_int j = 0; j < n; j++) {
                double s = 0.0;
                for (orc_int q = 0; q <= l; q++) s += W[j + q * ldw] * T[q + l * ldt];
                W[j + l * ldw] = s;
            }
    } else {
        for (orc_int l = 0; l < k; l++)
            for (orc_int j = 0; j < n; j++) {
                double s = 0.0;
                for (orc_int q = l; q < k; q++) s += W[j + q * ldw] * T[l + q * ldt];
                W[j + l * ldw] = s;
            }
    }
    /* C = C - V W' */
    for (orc_int j = 0; j < n; j++) {
        double *c = C + j * ldc;
        for (orc_int l = 0; l < k; l++) {
            double w = W[j + l * ldw];
            if (w == 0.0) continue;
            c[l] -= w;
            for (orc_int r = l + 1; r < m; r++) c[r] -= V[r + l * ldv] * w;
        }
    }
}

/* qr_larftb (SparseQR_factorize.c:1851-1904): T at W, larfb workspace at W + k*k, ldwork = n */
void orc_larftb(int method, orc_int m, orc_int n, orc_int k, orc_int ldc, orc_int ldv,
                const double *V, const double *Tau, double *C, double *W)
{
    if (m <= 0 || n <= 0 || k <= 0) return;
    double *T = W, *Work = W + k * k;
    orc_larft(m, k, V, ldv, Tau, T, k);
    larfb_left(method == 0 ? 'T' : 'N', m, n, k, V, ldv, T, k, C, ldc, Work, n);
}

/* ------------------------------------------------------------------------------------------------
 * qr_stranspose2 (SparseQR_factorize.c:755-785): numeric values of S = A(P,Q) in row form
 * ---------------------------------------------------------------------------------------------- */
void orc_stranspose2(orc_int m, orc_int n, const orc_int *Ap, const orc_int *Ai, const double *Ax,
                     const orc_int *Qfill, const orc_int *Sp, const orc_int *PLinv, double *Sx, orc_int *W)
{
    memcpy(W, Sp, sizeof(orc_int) * (size_t)m);
    for (orc_int col = 0; col < n; col++) {
        orc_int j = Qfill ? Qfill[col] : col;
        for (orc_int p = Ap[j]; p < Ap[j + 1]; p++) Sx[W[PLinv[Ai[p]]]++] = Ax[p];
    }
}

/* ------------------------------------------------------------------------------------------------
 * qr_fsize (SparseQR_factorize.c:1066-1145)
 * ---------------------------------------------------------------------------------------------- */
orc_int orc_fsize(orc_int f, const orc_int *Super, const orc_int *Rp, const orc_int *Rj, const orc_int *Sleft,
                  const orc_int *Child, const orc_int *Childp, const orc_int *Cm, orc_int *Fmap, orc_int *Stair)
{
    orc_int col1 = Super[f], fp = Super[f + 1] - col1;
    orc_int p1 = Rp[f], fn = Rp[f + 1] - p1;
    for (orc_int j = 0; j < fn; j++) Fmap[Rj[p1 + j]] = j;
    for (orc_int j = 0; j < fn; j++) Stair[j] = (j < fp) ? Sleft[col1 + j + 1] - Sleft[col1 + j] : 0;
    for (orc_int p = Childp[f]; p < Childp[f + 1]; p++) {
        orc_int c = Child[p];
        orc_int pc = Rp[c] + (Super[c + 1] - Super[c]);
        for (orc_int ci = 0; ci < Cm[c]; ci++) Stair[Fmap[Rj[pc + ci]]]++;
    }
    orc_int fm = 0;
    for (orc_int j = 0; j < fn; j++) { orc_int t = fm; fm += Stair[j]; Stair[j] = t; }
    return fm;
}

/* ------------------------------------------------------------------------------------------------
 * qr_assemble (SparseQR_factorize.c:1151-1285)
 * ---------------------------------------------------------------------------------------------- */
void orc_assemble(orc_int f, orc_int fm, const orc_int *Super, const orc_int *Rp, const orc_int *Rj,
                  const orc_int *Sp, const orc_int *Sj, const orc_int *Sleft, const orc_int *Child,
                  const orc_int *Childp, const double *Sx, const orc_int *Fmap, const orc_int *Cm,
                  double *const *Cblock, const orc_int *Hr, orc_int *Stair, orc_int *Hii,
                  const orc_int *Hip, double *F, orc_int *Cmap)
{
    orc_int col1 = Super[f], fp = Super[f + 1] - col1;
    orc_int fn = Rp[f + 1] - Rp[f];
    memset(F, 0, sizeof(double) * (size_t)(fm * fn));
    orc_int *Hi = Hii + Hip[f];

    /* rows of S whose leftmost column is a pivot column of f */
    for (orc_int k = 0; k < fp; k++) {
        for (orc_int row = Sleft[col1 + k]; row < Sleft[col1 + k + 1]; row++) {
            orc_int i = Stair[k]++;
            for (orc_int p = Sp[row]; p < Sp[row + 1]; p++) F[i + Fmap[Sj[p]] * fm] = Sx[p];
            Hi[i] = row;
        }
    }
    /* children: packed upper-trapezoidal C blocks */
    for (orc_int p = Childp[f]; p < Childp[f + 1]; p++) {
        orc_int c = Child[p];
        orc_int fpc = Super[c + 1] - Super[c];
        orc_int pc = Rp[c] + fpc;
        orc_int cn = (Rp[c + 1] - Rp[c]) - fpc;
        orc_int cm = Cm[c];
        const double *C = Cblock[c];
        const orc_int *Hichild = Hii + Hip[c] + Hr[c];
        for (orc_int ci = 0; ci < cm; ci++) {
            orc_int i = Stair[Fmap[Rj[pc + ci]]]++;
            Cmap[ci] = i;
            Hi[i] = Hichild[ci];
        }
        for (orc_int cj = 0; cj < cn; cj++) {
            double *Fj = F + fm * Fmap[Rj[pc + cj]];
            orc_int len = (cj < cm) ? cj + 1 : cm;
            for (orc_int ci = 0; ci < len; ci++) Fj[Cmap[ci]] = *C++;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * qr_front (SparseQR_factorize.c:1383-1618)
 * ---------------------------------------------------------------------------------------------- */
orc_int orc_front(orc_int m, orc_int n, orc_int npiv, double tol, orc_int ntol, const orc_chunk *ch,
                  double *F, orc_int *Stair, char *Rdead, double *Tau, double *W, double *flops)
{
    orc_int fchunk = IMAX(ch->fchunk, 1);
    orc_int minchunk = IMAX(ch->minchunk, fchunk / ch->minchunk_ratio);
    npiv = IMIN(n, IMAX(0, npiv));
    ntol = IMIN(ntol, npiv);
    orc_int rank = IMIN(m, npiv);
    orc_int g = 0, g1 = 0, k1 = 0, k2 = 0, nv = 0, vzeros = 0, t = 0;
    double *V = F;

    for (orc_int k = 0; k < n; k++) {
        orc_int t0 = t;
        t = Stair[k];
        if (g >= m) {
            /* no rows left: remaining pivots are dead, remaining columns are empty */
            for (; k < npiv; k++) { Rdead[k] = 1; Stair[k] = 0; Tau[k] = 0; }
            for (; k < n; k++)    { Stair[k] = m; Tau[k] = 0; }
            return rank;
        }
        t = IMAX(g + 1, t);
        Stair[k] = t;

        /* flush early when the pending block reflector is mostly structural zeros (:1467-1483) */
        vzeros += nv * (t - t0);
        if (nv >= minchunk) {
            orc_int vsize = (nv * (nv + 1)) / 2 + nv * (t - g1 - nv);
            if (vzeros > IMAX(16, vsize / 2)) {
                orc_larftb(0, t0 - g1, n - k2, nv, m, m, V, Tau + k1, F + g1 + k2 * m, W);
                nv = 0; vzeros = 0;
            }
        }

        double *col = F + g + k * m;
        double tau = orc_larfg(t - g, col, col + 1);
        double wk;
        if (k < ntol && (wk = fabs(*col)) <= tol) {
            /* dead pivot column (:1495-1544) */
            for (orc_int i = g; i < m; i++) F[i + k * m] = 0;
            Stair[k] = 0; Tau[k] = 0; Rdead[k] = 1;
            if (nv > 0) {
                orc_larftb(0, t0 - g1, n - k2, nv, m, m, V, Tau + k1, F + g1 + k2 * m, W);
                nv = 0; vzeros = 0;
            }
        } else {
            Tau[k] = tau;
            if (nv == 0) {
                g1 = g; k1 = k; k2 = IMIN(n, k + fchunk);
                V = F + g1 + k1 * m;
                orc_int mleft = m - g1, nleft = n - k1;
                if (mleft * (nleft - (fchunk + 4)) < ch->small || mleft <= fchunk / 2 || fchunk <= 1) k2 = n;
            }
            nv++;
            if (flops) *flops += (double)((t - g) * (3 + 4 * (n - k - 1)));
            /* apply H_k to the rest of the panel (:1577, qr_private_apply1 :1359-1381) */
            if (t - g > 0 && k2 - k - 1 > 0) {
                double save = *col;
                *col = 1;
                orc_larf_left(t - g, k2 - k - 1, col, tau, col + m, m, W);
                *col = save;
            }
            g++;
            if (k == k2 - 1 || g == m) {
                orc_larftb(0, t - g1, n - k2, nv, m, m, V, Tau + k1, F + g1 + k2 * m, W);
                nv = 0; vzeros = 0;
            }
        }
        if (k == npiv - 1) rank = g;
    }
    return rank;
}

/* qr_fcsize (:1623-1634) */
orc_int orc_fcsize(orc_int m, orc_int n, orc_int npiv, orc_int rank)
{
    orc_int cn = n - npiv, cm = IMIN(m - rank, cn);
    return (cm * (cm + 1)) / 2 + cm * (cn - cm);
}

/* qr_cpack (:1639-1685) */
orc_int orc_cpack(orc_int m, orc_int n, orc_int npiv, orc_int rank, const double *F, double *C)
{
    orc_int cn = n - npiv, cm = IMIN(m - rank, cn);
    if (cm <= 0 || cn <= 0) return 0;
    F += rank + npiv * m;
    for (orc_int k = 0; k < cn; k++, F += m) {
        orc_int len = (k < cm) ? k + 1 : cm;
        memcpy(C, F, sizeof(double) * (size_t)len);
        C += len;
    }
    return cm;
}

/* qr_rhpack, keepH = TRUE (:1691-1784).  F and R may alias (in-place compaction at the start of F). */
orc_int orc_rhpack(orc_int m, orc_int n, orc_int npiv, const orc_int *Stair, const double *F, double *R, orc_int *p_rm)
{
    double *R0 = R;
    if (m <= 0 || n <= 0) { *p_rm = 0; return 0; }
    orc_int rm = 0, k;
    for (k = 0; k < npiv; k++, F += m) {
        orc_int t = Stair[k];
        if (t == 0) t = rm;
        else if (rm < m) rm++;
        for (orc_int i = 0; i < t; i++) *R++ = F[i];
    }
    orc_int h = rm;
    for (; k < n; k++, F += m) {
        for (orc_int i = 0; i < rm; i++) *R++ = F[i];
        orc_int t = Stair[k];
        h = IMIN(h + 1, m);
        for (orc_int i = h; i < t; i++) *R++ = F[i];
    }
    *p_rm = rm;
    return (orc_int)(R - R0);
}

/* qr_hpinv (:991-1060) */
void orc_hpinv(const orc_symbolic *S, orc_numeric *N, orc_int *W)
{
    orc_int nf = S->nf, m = S->m, n = S->n, row1 = 0, row2 = m, maxfm = 0;
    for (orc_int i = S->Sleft[n]; i < m; i++) W[i] = --row2;
    for (orc_int f = 0; f < nf; f++) {
        orc_int *Hi = N->Hii + S->Hip[f];
        orc_int rm = N->Hr[f], fm = N->Hm[f];
        for (orc_int i = 0; i < rm; i++) W[Hi[i]] = row1++;
        orc_int cn = (S->Rp[f + 1] - S->Rp[f]) - (S->Super[f + 1] - S->Super[f]);
        orc_int cm = IMIN(fm - rm, cn);
        maxfm = IMAX(maxfm, fm);
        for (orc_int i = fm - 1; i >= rm + cm; i--) W[Hi[i]] = --row2;
    }
    N->maxfm = maxfm;
    for (orc_int i = 0; i < m; i++) N->HPinv[i] = W[S->PLinv[i]];
    for (orc_int f = 0; f < nf; f++) {
        orc_int *Hi = N->Hii + S->Hip[f];
        for (orc_int i = 0; i < N->Hm[f]; i++) Hi[i] = W[Hi[i]];
    }
}

/* ------------------------------------------------------------------------------------------------
 * qr_factorize + qr_kernel, serial path (SparseQR_factorize.c:222-749, 791-985)
 * ---------------------------------------------------------------------------------------------- */
int orc_factorize(const orc_symbolic *S, const orc_int *Ap, const orc_int *Ai, const double *Ax,
                  double tol, orc_int ntol, const orc_chunk *ch_in, orc_numeric *N)
{
    orc_int m = S->m, n = S->n, nf = S->nf, maxfn = S->maxfn;
    if (!S->do_rank_detection) tol = -1;
    orc_chunk ch = *ch_in;
    ch.fchunk = IMIN(m, ch.fchunk);
    orc_int wtsize = IMAX(ch.fchunk, 1) * IMAX(maxfn, 1);

    orc_int *Wi = malloc(sizeof(orc_int) * (size_t)(IMAX(m, nf) + 1));
    double **Cblock = malloc(sizeof(double *) * (size_t)(nf + 1));
    double *Sx = malloc(sizeof(double) * (size_t)IMAX(S->anz, 1));
    orc_int *Fmap = malloc(sizeof(orc_int) * (size_t)IMAX(n, 1));
    orc_int *Cmap = malloc(sizeof(orc_int) * (size_t)IMAX(maxfn, 1));
    double *W = malloc(sizeof(double) * (size_t)wtsize);
    if (!Wi || !Cblock || !Sx || !Fmap || !Cmap || !W) return -2;

    orc_stranspose2(m, n, Ap, Ai, Ax, S->Qfill, S->Sp, S->PLinv, Sx, Wi);
    memset(N->Rdead, 0, (size_t)n);
    orc_int *Cm = N->Cm;

    double *Stack = N->Stack;
    double *head = Stack, *top = Stack + S->maxstack;
    orc_int sumfrank = 0, maxfrank = 1, csave = 0;   /* maxfrank starts at 1 (:555) */
    N->flopcount = 0; N->t_assemble = N->t_front = N->t_pack = 0;

    for (orc_int kf = 0; kf < nf; kf++) {
        orc_int f = S->Post[kf];
        orc_int *Stair = N->HStair + S->Rp[f];
        double *Tau = N->HTau + S->Rp[f];
        orc_int fm = orc_fsize(f, S->Super, S->Rp, S->Rj, S->Sleft, S->Child, S->Childp, Cm, Fmap, Stair);
        orc_int fn = S->Rp[f + 1] - S->Rp[f];
        orc_int col1 = S->Super[f], fp = S->Super[f + 1] - col1;
        orc_int fsize = fm * fn;
        N->Hm[f] = fm;
        double *F = head;
        N->Rblock_off[f] = F - Stack;
        head += fsize;

        double t0 = wall();
        orc_assemble(f, fm, S->Super, S->Rp, S->Rj, S->Sp, S->Sj, S->Sleft, S->Child, S->Childp, Sx, Fmap,
                     Cm, Cblock, N->Hr, Stair, N->Hii, S->Hip, F, Cmap);
        double t1 = wall();
        /* pop the children's C blocks (:925-933) */
        for (orc_int p = S->Childp[f]; p < S->Childp[f + 1]; p++) {
            orc_int c = S->Child[p];
            orc_int fpc = S->Super[c + 1] - S->Super[c], cn = (S->Rp[c + 1] - S->Rp[c]) - fpc, cm = Cm[c];
            double *end = Cblock[c] + (cm * (cm + 1)) / 2 + cm * (cn - cm);
            if (end > top) top = end;
        }
        orc_int frank = orc_front(fm, fn, fp, tol, ntol - col1, &ch, F, Stair, N->Rdead + col1, Tau, W, &N->flopcount);
        double t2 = wall();
        sumfrank += frank;
        maxfrank = IMAX(maxfrank, frank);

        orc_int csize = orc_fcsize(fm, fn, fp, frank);
        top -= csize;
        Cblock[f] = top;
        Cm[f] = orc_cpack(fm, fn, fp, frank, F, top);
        if (N->Csave) {
            N->Csave_off[f] = csave;
            memcpy(N->Csave + csave, top, sizeof(double) * (size_t)csize);
            csave += csize;
        }
        orc_int rm;
        orc_int rsize = orc_rhpack(fm, fn, fp, Stair, F, F, &rm);
        N->Hr[f] = rm;
        head = F + rsize;
        double t3 = wall();
        N->t_assemble += t1 - t0; N->t_front += t2 - t1; N->t_pack += t3 - t2;
    }
    N->rank = sumfrank;
    N->maxfrank = maxfrank;
    N->rh_total = head - Stack;
    orc_hpinv(S, N, Wi);
    if (ntol >= n) N->rank1 = N->rank;
    else { orc_int r = 0; for (orc_int j = 0; j < ntol; j++) r += !N->Rdead[j]; N->rank1 = r; }

    free(Wi); free(Cblock); free(Sx); free(Fmap); free(Cmap); free(W);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Checkers: apply Q / Q', multiply / solve with R straight from the packed R+H blocks
 * (format: qr_rhpack :1691-1784; decoders in the reference: SparseQR.c:1455-1545, 2218-2517)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { orc_int start, len, row; double tau; } hvec;

/* enumerate the Householder vectors of front f: start/len inside the packed block, first front row */
static orc_int front_hvecs(const orc_symbolic *S, const orc_numeric *N, orc_int f, hvec *H)
{
    orc_int fp = S->Super[f + 1] - S->Super[f], pr = S->Rp[f], fn = S->Rp[f + 1] - pr, fm = N->Hm[f];
    const orc_int *Stair = N->HStair + pr;
    const double *Tau = N->HTau + pr;
    orc_int p = 0, rm = 0, h = 0, nh = 0;
    for (orc_int k = 0; k < fn && nh < fm; k++) {
        orc_int t = Stair[k];
        if (k < fp) {
            if (t == 0) { p += rm; continue; }
            if (rm < fm) rm++;
            h = rm;
        } else {
            h = IMIN(h + 1, fm);
        }
        p += rm;
        H[nh].tau = Tau[k]; H[nh].start = p; H[nh].len = IMAX(t - h, 0); H[nh].row = nh;
        p += IMAX(t - h, 0);
        nh++;
        if (h == fm) break;
    }
    return nh;
}

void orc_qmult(int method, const orc_symbolic *S, const orc_numeric *N, double *x, double *work)
{
    orc_int m = S->m, nf = S->nf;
    hvec *H = malloc(sizeof(hvec) * (size_t)(S->maxfn + 1));
    if (method == 0) {
        for (orc_int i = 0; i < m; i++) work[N->HPinv[i]] = x[i];
        for (orc_int f = 0; f < nf; f++) {
            orc_int nh = front_hvecs(S, N, f, H);
            const double *R = N->Stack + N->Rblock_off[f];
            const orc_int *Hi = N->Hii + S->Hip[f];
            for (orc_int q = 0; q < nh; q++) {
                if (H[q].tau == 0) continue;
                const double *v = R + H[q].start;
                const orc_int *rows = Hi + H[q].row;
                double s = work[rows[0]];
                for (orc_int i = 0; i < H[q].len; i++) s += v[i] * work[rows[i + 1]];
                s *= H[q].tau;
                work[rows[0]] -= s;
                for (orc_int i = 0; i < H[q].len; i++) work[rows[i + 1]] -= s * v[i];
            }
        }
        memcpy(x, work, sizeof(double) * (size_t)m);
    } else {
        memcpy(work, x, sizeof(double) * (size_t)m);
        for (orc_int f = nf - 1; f >= 0; f--) {
            orc_int nh = front_hvecs(S, N, f, H);
            const double *R = N->Stack + N->Rblock_off[f];
            const orc_int *Hi = N->Hii + S->Hip[f];
            for (orc_int q = nh - 1; q >= 0; q--) {
                if (H[q].tau == 0) continue;
                const double *v = R + H[q].start;
                const orc_int *rows = Hi + H[q].row;
                double s = work[rows[0]];
                for (orc_int i = 0; i < H[q].len; i++) s += v[i] * work[rows[i + 1]];
                s *= H[q].tau;
                work[rows[0]] -= s;
                for (orc_int i = 0; i < H[q].len; i++) work[rows[i + 1]] -= s * v[i];
            }
        }
        for (orc_int i = 0; i < m; i++) x[i] = work[N->HPinv[i]];
    }
    free(H);
}

/* walk the R part of front f: calls visit(row_in_front, global_col, value_ptr) semantics inline */
void orc_rmult(const orc_symbolic *S, const orc_numeric *N, const double *x, double *y)
{
    orc_int m = S->m, nf = S->nf, row0 = 0;
    for (orc_int i = 0; i < m; i++) y[i] = 0;
    for (orc_int f = 0; f < nf; f++) {
        orc_int fp = S->Super[f + 1] - S->Super[f], pr = S->Rp[f], fn = S->Rp[f + 1] - pr, fm = N->Hm[f];
        const orc_int *Stair = N->HStair + pr;
        const double *R = N->Stack + N->Rblock_off[f];
        orc_int rm = 0, h = 0;
        for (orc_int k = 0; k < fn; k++) {
            orc_int t = Stair[k], nr, skip;
            if (k < fp) {
                if (t == 0) { nr = rm; skip = rm; }
                else { if (rm < fm) rm++; nr = rm; skip = t; h = rm; }
            } else {
                nr = rm; h = IMIN(h + 1, fm); skip = rm + IMAX(t - h, 0);
            }
            double xk = x[S->Rj[pr + k]];
            for (orc_int i = 0; i < nr; i++) y[row0 + i] += R[i] * xk;
            R += skip;
        }
        row0 += N->Hr[f];
    }
}

int orc_rsolve(const orc_symbolic *S, const orc_numeric *N, const double *y, double *x)
{
    /* qr_rsolve, STMMQR/src/qr/SparseQR.c:2218-2470 (multifrontal rows; no singletons here): back substitution over the
     * fronts in reverse order; a dead pivot column gets x = 0 (basic solution), the live pivot columns of a front form
     * an rm x rm upper triangle whose row i is the i-th live column; y is indexed by the row order of R */
    orc_int n = S->n, nf = S->nf;
    orc_int *row0 = malloc(sizeof(orc_int) * (size_t)(nf + 1));
    row0[0] = 0;
    for (orc_int f = 0; f < nf; f++) row0[f + 1] = row0[f] + N->Hr[f];
    orc_int *coff = malloc(sizeof(orc_int) * (size_t)(S->maxfn + 1));
    orc_int *live = malloc(sizeof(orc_int) * (size_t)(S->maxfn + 1));
    double *acc = malloc(sizeof(double) * (size_t)(S->maxfn + 1));
    for (orc_int j = 0; j < n; j++) x[j] = 0;
    for (orc_int f = nf - 1; f >= 0; f--) {
        orc_int fp = S->Super[f + 1] - S->Super[f], pr = S->Rp[f], fn = S->Rp[f + 1] - pr, fm = N->Hm[f];
        const orc_int *Stair = N->HStair + pr;
        const double *R = N->Stack + N->Rblock_off[f];
        orc_int rm = 0, h = 0, p = 0;
        /* column offsets inside the packed block (qr_rhpack order) and the list of live pivot columns */
        for (orc_int k = 0; k < fn; k++) {
            coff[k] = p;
            orc_int t = Stair[k];
            if (k < fp) {
                if (t == 0) { p += rm; }                       /* dead: rm R entries, no H */
                else { if (rm < fm) { live[rm] = k; rm++; } h = rm; p += t; }
            } else { h = IMIN(h + 1, fm); p += rm + IMAX(t - h, 0); }
        }
        /* (rm at the non-pivotal columns is the final rm: every pivot column comes first) */
        for (orc_int i = 0; i < rm; i++) acc[i] = y[row0[f] + i];
        for (orc_int k = fp; k < fn; k++) {
            double xk = x[S->Rj[pr + k]];
            if (xk != 0) for (orc_int i = 0; i < rm; i++) acc[i] -= R[coff[k] + i] * xk;
        }
        for (orc_int q = rm - 1; q >= 0; q--) {
            orc_int k = live[q];
            double xk = acc[q] / R[coff[k] + q];
            x[S->Super[f] + k] = xk;
            for (orc_int i = 0; i < q; i++) acc[i] -= R[coff[k] + i] * xk;
        }
    }
    free(row0); free(coff); free(live); free(acc);
    return 0;
}
