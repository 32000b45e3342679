/* oracle/stmmqr_oracle.h -- TEST INFRASTRUCTURE ONLY, never part of the shipped product.
 *
 * CPU restatement ("port") of the reference's numeric multifrontal-QR path
 *   STMMQR/src/qr/SparseQR_factorize.c  (qr_factorize ... qr_larftb)
 * in plain C, with own LAPACK-semantics kernels (dlarfg/dlarf/dlarft/dlarfb are NOT under
 * /root/reference: the reference links OpenBLAS-0.3.9 + LAPACK, STMMQR/README.md:10,
 * Makefile.option:108; restated from the published LAPACK 3.x algorithms).
 *
 * Pinning: validated against the REAL reference built by oracle/Makefile (`make ref`,
 * oracle/_ref/libstmmqr_ref.so) -- integer outputs bit-exact, floating-point outputs to a
 * normwise 1e-11 -- by tests/test_oracle_vs_ref.py (runs where /root/reference exists) and
 * against the committed golden vectors tests/golden/*.npz generated from that reference by
 * tests/golden/make_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use this library.
 */
#ifndef STMMQR_ORACLE_H
#define STMMQR_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef long orc_int;   /* the reference's Sparse_long (SparseBase_config.h:29) */

/* symbolic object, same arrays as qr_symbolic (SparseQR_struct.h:26-137) */
typedef struct {
    orc_int m, n, anz, nf, maxfn, rjsize, hisize, maxstack, do_rank_detection;
    const orc_int *Sp, *Sj, *Qfill, *PLinv, *Sleft;
    const orc_int *Child, *Childp, *Super, *Rp, *Rj, *Post, *Hip;
} orc_symbolic;

/* numeric object; all arrays caller-allocated */
typedef struct {
    double  *Stack;        /* maxstack doubles: packed R+H at the bottom when done      */
    orc_int *Rblock_off;   /* nf: offset of front f's packed R+H inside Stack           */
    char    *Rdead;        /* n                                                         */
    orc_int *HStair;       /* rjsize                                                    */
    double  *HTau;         /* rjsize                                                    */
    orc_int *Hii;          /* hisize                                                    */
    orc_int *HPinv;        /* m                                                         */
    orc_int *Hm, *Hr;      /* nf                                                        */
    orc_int *Cm;           /* nf: numeric rows of each front's contribution block       */
    orc_int rank, rank1, maxfrank, maxfm, rh_total;
    double  flopcount;     /* the reference's FLOP_COUNT total (SparseQR_factorize.c:1571) */
    /* optional debugging capture (may be NULL) */
    double  *Csave;        /* all packed C blocks, concatenated in postorder            */
    orc_int *Csave_off;    /* nf                                                         */
    double  t_assemble, t_front, t_pack;   /* seconds                                   */
} orc_numeric;

typedef struct { orc_int fchunk, small, minchunk, minchunk_ratio; } orc_chunk;

/* SparseQR_factorize.c:755-785 */
void orc_stranspose2(orc_int m, orc_int n, const orc_int *Ap, const orc_int *Ai, const double *Ax,
                     const orc_int *Qfill, const orc_int *Sp, const orc_int *PLinv, double *Sx, orc_int *W);
/* :1066-1145 */
orc_int orc_fsize(orc_int f, const orc_int *Super, const orc_int *Rp, const orc_int *Rj, const orc_int *Sleft,
                  const orc_int *Child, const orc_int *Childp, const orc_int *Cm, orc_int *Fmap, orc_int *Stair);
/* :1151-1285 ; Cblock_off indexes into Cbase */
void orc_assemble(orc_int f, orc_int fm, const orc_int *Super, const orc_int *Rp, const orc_int *Rj,
                  const orc_int *Sp, const orc_int *Sj, const orc_int *Sleft, const orc_int *Child,
                  const orc_int *Childp, const double *Sx, const orc_int *Fmap, const orc_int *Cm,
                  double *const *Cblock, const orc_int *Hr, orc_int *Stair, orc_int *Hii,
                  const orc_int *Hip, double *F, orc_int *Cmap);
/* :1383-1618 */
orc_int orc_front(orc_int m, orc_int n, orc_int npiv, double tol, orc_int ntol, const orc_chunk *ch,
                  double *F, orc_int *Stair, char *Rdead, double *Tau, double *W, double *flops);
/* :1851-1904, method QR_QTX only (0) plus QR_QX (1) for the Q-apply checker */
void orc_larftb(int method, orc_int m, orc_int n, orc_int k, orc_int ldc, orc_int ldv,
                const double *V, const double *Tau, double *C, double *W);
/* :1623-1634, :1639-1685, :1691-1784 */
orc_int orc_fcsize(orc_int m, orc_int n, orc_int npiv, orc_int rank);
orc_int orc_cpack(orc_int m, orc_int n, orc_int npiv, orc_int rank, const double *F, double *C);
orc_int orc_rhpack(orc_int m, orc_int n, orc_int npiv, const orc_int *Stair, const double *F, double *R, orc_int *p_rm);
/* :991-1060 */
void orc_hpinv(const orc_symbolic *S, orc_numeric *N, orc_int *W);
/* :222-749 (serial path: qr_kernel(0) over Post), returns 0 on success */
int orc_factorize(const orc_symbolic *S, const orc_int *Ap, const orc_int *Ai, const double *Ax,
                  double tol, orc_int ntol, const orc_chunk *ch, orc_numeric *N);

/* LAPACK-semantics building blocks, exposed for unit checks */
double orc_larfg(orc_int n, double *alpha, double *x);
void   orc_larf_left(orc_int m, orc_int n, const double *v, double tau, double *C, orc_int ldc, double *work);
void   orc_larft(orc_int n, orc_int k, const double *V, orc_int ldv, const double *tau, double *T, orc_int ldt);

/* consumers used as checkers (restating SparseQR.c:1455-1545,1706-1836,2218-2517 in unblocked form):
 *  x (length m) is overwritten.  method 0: x <- Q' x, 1: x <- Q x ; both include the HPinv row permutation */
void orc_qmult(int method, const orc_symbolic *S, const orc_numeric *N, double *x, double *work);
/* y(0:m) <- R x, R is rank-by-n upper trapezoidal (squeezed form) */
void orc_rmult(const orc_symbolic *S, const orc_numeric *N, const double *x, double *y);
/* solve R x = y (qr_rsolve): dead pivot columns get x = 0 (basic solution); returns 0 */
int  orc_rsolve(const orc_symbolic *S, const orc_numeric *N, const double *y, double *x);

#ifdef __cplusplus
}
#endif
#endif
