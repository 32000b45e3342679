/* oracle/refapi.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's PUBLIC API run end to end on one matrix, with nothing interposed:
 *     SparseQR -> QR_qmult x {QR_QTX, QR_QX, QR_XQT, QR_XQ} -> QR_solve x {RX_EQUALS_B, RETX_EQUALS_B, RTX_EQUALS_B,
 *     RTX_EQUALS_ETB} -> SparseQR_free      (STMMQR/include/SparseQR.h:25-36,403-417; SparseQR.c:66,1838,2118)
 * on seeded dense operands, every result written in refdump's tagged binary format.
 *
 * oracle/Makefile links this file twice:
 *   _ref/refapi           against the compiled reference (libstmmqr_ref.so)            -> golden outputs
 *   _ref/refapi_relinked  against the reference's objects MINUS SparseQR_factorize.o and SparseQR_multithreads.o, plus
 *                         the binding stub of INTEGRATION.md 2 and -lstmmqr_hip -- exactly the link recipe of
 *                         INTEGRATION.md 1.  There the reference's SparseQR() calls this repository's qr_factorize and its
 *                         qr_panel (SparseQR.c:1659,1663) calls this repository's qr_larftb with all four methods.
 * tests/test_relinked_reference.py compares the two.
 *
 * usage: refapi <matrix.mtx> <ordering -1|0..3> <out.bin>
 */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "SparseQR.h"

#ifndef Long
#define Long Sparse_long
#endif

static FILE *g_out = NULL;
static void put(const char *name, char ty, long count, const void *data)
{
    char tag[32];
    memset(tag, 0, sizeof tag);
    strncpy(tag, name, 31);
    fwrite(tag, 1, 32, g_out);
    fwrite(&ty, 1, 1, g_out);
    fwrite(&count, sizeof(long), 1, g_out);
    if (count > 0) fwrite(data, 8, (size_t)count, g_out);
}
static void put_l(const char *name, long v) { put(name, 'q', 1, &v); }

/* seeded operand entry (i, j): smooth, no symmetry, O(1) */
static double entry(Long i, Long j) { return cos(0.37 * (double)i + 1.3 * (double)j) + 0.25 * sin(0.011 * (double)i * (double)(j + 1)) + 0.01 * (double)j; }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s <matrix.mtx> <ordering> <out.bin>\n", argv[0]); return 1; }
    sparse_common Common, *cc = &Common;
    SparseCore_start(cc);
    FILE *fp = fopen(argv[1], "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    FILE *dn1 = fopen("/dev/null", "a+"), *dn2 = fopen("/dev/null", "a+");
    int mtype;
    sparse_csc *A = (sparse_csc *)SparseCore_read_matrix(fp, 1, &mtype, cc, dn1, dn2, 0);
    fclose(fp); fclose(dn1); fclose(dn2);
    if (!A || mtype != SPARSE_CSC) { fprintf(stderr, "input must be sparse\n"); return 1; }
    const Long m = A->nrow, n = A->ncol;
    double mx = qr_maxcolnorm(A, cc);
    if (mx == 0) mx = 1;
    const double tol = 20 * ((double)m + (double)n) * DBL_EPSILON * mx;          /* qrtest.c:135-142 */
    cc->SPQR_grain = 1;
    cc->status = SPARSE_OK;
    Relaxfactor_setting(n, SparseCore_nnz(A, cc), RELAX_FOR_QR, cc);
    long ordering;
    switch (atoi(argv[2])) {
        case 0: ordering = QR_ORDERING_AMD; break;
        case 1: ordering = QR_ORDERING_COLAMD; break;
        case 2: ordering = QR_ORDERING_ONLYMETIS; break;
        case 3: ordering = QR_ORDERING_NESDIS; break;
        default: ordering = QR_ORDERING_DEFAULT;
    }
    chunk_getSettings(32, 5000, 4, 4);
    char name[64] = "refapi";
    SparseQR_factorization *QR = SparseQR((int)ordering, tol, A, cc, name);
    if (!QR) { fprintf(stderr, "SparseQR failed, status %d\n", cc->status); return 3; }
    g_out = fopen(argv[3], "wb");
    if (!g_out) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    put_l("m", m); put_l("n", n); put_l("rank", QR->rank); put_l("n1rows", QR->n1rows); put_l("n1cols", QR->n1cols);
    put_l("status_after_factorize", cc->status);

    const Long nr = 3;
    /* ---- QR_qmult, all four methods ---- */
    for (int method = QR_QTX; method <= QR_XQ; method++) {
        const int left = (method == QR_QTX || method == QR_QX);
        dense_array *X = left ? SparseCore_zeros(m, nr, SPARSE_REAL, cc) : SparseCore_zeros(nr, m, SPARSE_REAL, cc);
        double *x = (double *)X->x;
        if (left) { for (Long j = 0; j < nr; j++) for (Long i = 0; i < m; i++) x[i + j * m] = entry(i, j); }
        else      { for (Long i = 0; i < m; i++) for (Long k = 0; k < nr; k++) x[k + i * nr] = entry(i, k); }
        dense_array *Y = QR_qmult(method, QR, X, cc);
        char tag[32]; snprintf(tag, sizeof tag, "qmult_%d", method);
        if (!Y) { fprintf(stderr, "QR_qmult(%d) failed, status %d\n", method, cc->status); return 4; }
        put(tag, 'd', (long)(Y->nrow * Y->ncol), Y->x);
        snprintf(tag, sizeof tag, "qmult_%d_status", method);
        put_l(tag, cc->status);
        SparseCore_free_dense(&Y, cc);
        SparseCore_free_dense(&X, cc);
    }
    /* ---- Q (Q'X): must give X back whatever the signs of the reflectors ---- */
    {
        dense_array *X = SparseCore_zeros(m, nr, SPARSE_REAL, cc);
        double *x = (double *)X->x;
        for (Long j = 0; j < nr; j++) for (Long i = 0; i < m; i++) x[i + j * m] = entry(i, j);
        dense_array *Y = QR_qmult(QR_QTX, QR, X, cc);
        dense_array *Z = Y ? QR_qmult(QR_QX, QR, Y, cc) : NULL;
        if (!Z) { fprintf(stderr, "QR_qmult round trip failed, status %d\n", cc->status); return 4; }
        put("qmult_x", 'd', (long)(m * nr), X->x);
        put("qmult_10", 'd', (long)(m * nr), Z->x);
        SparseCore_free_dense(&Z, cc); SparseCore_free_dense(&Y, cc); SparseCore_free_dense(&X, cc);
    }
    /* ---- QR_solve, all four systems (B: m x nr for the R systems, n x nr for the R' systems) ---- */
    for (int system = QR_RX_EQUALS_B; system <= QR_RTX_EQUALS_ETB; system++) {
        const Long rows = (system <= QR_RETX_EQUALS_B) ? m : n;
        dense_array *B = SparseCore_zeros(rows, nr, SPARSE_REAL, cc);
        double *b = (double *)B->x;
        for (Long j = 0; j < nr; j++) for (Long i = 0; i < rows; i++) b[i + j * rows] = entry(i + 5, j + 2);
        dense_array *X = QR_solve(system, QR, B, cc);
        char tag[32]; snprintf(tag, sizeof tag, "solve_%d", system);
        if (!X) { fprintf(stderr, "QR_solve(%d) failed, status %d\n", system, cc->status); return 5; }
        put(tag, 'd', (long)(X->nrow * X->ncol), X->x);
        SparseCore_free_dense(&X, cc);
        SparseCore_free_dense(&B, cc);
    }
    /* ---- R and H in sparse form (SparseLQ.c: qr_rcount :102, qr_rconvert :299, qr_trapezoidal :519) on the multifrontal
     *      part; small inputs only (the arrays go into a committed fixture) ---- */
    if (m <= 500) {
        qr_symbolic *S = QR->QRsym; qr_numeric *N = QR->QRnum;
        const Long n2s = S->n, econ = S->m;
        Long *Ra = calloc((size_t)n2s + 1, sizeof(Long)), *H2p = calloc((size_t)S->rjsize + 2, sizeof(Long)), nh = 0;
        qr_rcount(S, N, 0, econ, n2s, 0, Ra, NULL, H2p, &nh);
        Long *Rpp = malloc(((size_t)n2s + 1) * sizeof(Long)), tot = 0;
        for (Long j = 0; j < n2s; j++) { Rpp[j] = tot; tot += Ra[j]; }
        Rpp[n2s] = tot;
        Long *fill = malloc(((size_t)n2s + 1) * sizeof(Long));
        memcpy(fill, Rpp, ((size_t)n2s + 1) * sizeof(Long));
        Long *Rai = malloc((size_t)(tot > 0 ? tot : 1) * sizeof(Long)); double *Rax = malloc((size_t)(tot > 0 ? tot : 1) * sizeof(double));
        const Long hnz = H2p[nh];
        Long *H2i = malloc((size_t)(hnz > 0 ? hnz : 1) * sizeof(Long)); double *H2x = malloc((size_t)(hnz > 0 ? hnz : 1) * sizeof(double));
        double *H2Tau = malloc((size_t)(nh > 0 ? nh : 1) * sizeof(double));
        qr_rconvert(S, N, 0, econ, n2s, 0, fill, Rai, Rax, NULL, NULL, NULL, H2p, H2i, H2x, H2Tau);
        put("rc_Rp", 'q', (long)n2s + 1, Rpp); put("rc_Ri", 'q', (long)tot, Rai); put("rc_Rx", 'd', (long)tot, Rax);
        put_l("rc_nh", nh); put("rc_Hp", 'q', (long)nh + 1, H2p); put("rc_Hi", 'q', (long)hnz, H2i); put("rc_Hx", 'd', (long)hnz, H2x);
        put("rc_HTau", 'd', (long)nh, H2Tau);
        Long *Tp, *Ti, *Qtrap; double *Tx;
        Long trank = qr_trapezoidal(n2s, Rpp, Rai, Rax, 0, S->Qfill, 0, &Tp, &Ti, &Tx, &Qtrap, cc);
        put_l("rc_trap_rank", trank);
        if (Tp) {
            put("rc_Tp", 'q', (long)n2s + 1, Tp); put("rc_Ti", 'q', (long)tot, Ti); put("rc_Tx", 'd', (long)tot, Tx); put("rc_Qtrap", 'q', (long)n2s, Qtrap);
            SparseCore_free(n2s + 1, sizeof(Long), Tp, cc); SparseCore_free(tot, sizeof(Long), Ti, cc);
            SparseCore_free(tot, sizeof(double), Tx, cc); SparseCore_free(n2s, sizeof(Long), Qtrap, cc);
        }
        free(Ra); free(H2p); free(Rpp); free(fill); free(Rai); free(Rax); free(H2i); free(H2x); free(H2Tau);
    }
    /* ---- the driver's own acceptance flow (qrtest.c:11-53): b = A [0..n-1], x = E R \ (Q'b) ---- */
    {
        double one[2] = {1, 0}, zero[2] = {0, 0};
        dense_array *X0 = SparseCore_zeros(n, 1, SPARSE_REAL, cc), *B = SparseCore_zeros(m, 1, SPARSE_REAL, cc);
        for (Long i = 0; i < n; i++) ((double *)X0->x)[i] = (double)i;
        SparseCore_sdmult(A, 0, one, zero, X0, B, cc);
        dense_array *Y = QR_qmult(QR_QTX, QR, B, cc);
        dense_array *Xs = Y ? QR_solve(QR_RETX_EQUALS_B, QR, Y, cc) : NULL;
        if (!Xs) { fprintf(stderr, "driver flow failed, status %d\n", cc->status); return 6; }
        put("driver_x", 'd', (long)n, Xs->x);
        double d = 0;
        for (Long j = 0; j < n; j++) { const double e = ((double *)Xs->x)[j] - (double)j; d += e * e; }
        printf("res = %8.1e\n", sqrt(d) / (double)n);
        SparseCore_free_dense(&Y, cc); SparseCore_free_dense(&Xs, cc);
        SparseCore_free_dense(&X0, cc); SparseCore_free_dense(&B, cc);
    }
    put_l("status_end", cc->status);
    SparseQR_free(&QR, cc);                       /* (qr_freenum releases what this repository's qr_factorize allocated) */
    SparseCore_free_sparse(&A, cc);
    /* what is still allocated now is the common workspace only: the same count and bytes in both links */
    put_l("malloc_count_exit", (long)cc->malloc_count);
    put_l("memory_inuse_exit", (long)cc->memory_inuse);
    fclose(g_out);
    printf("malloc_count at exit = %ld\n", (long)cc->malloc_count);
    SparseCore_finish(cc);
    return 0;
}
