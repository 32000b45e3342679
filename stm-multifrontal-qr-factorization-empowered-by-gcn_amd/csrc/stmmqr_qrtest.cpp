// stmmqr_qrtest.cpp -- the reference's Matrix Market test driver on this library alone.
//
//     stmmqr_qrtest <matrix.mtx> <graph_id> [ordering]        (argument-compatible with STMMQR/test/qrtest.c:65-217)
//       ordering: 0 AMD, 1 COLAMD, 2 METIS, 3 NESDIS, absent = default = COLAMD (qrtest.c:155-169)
//
// Prints the reference driver's lines ("Matrix %6ld-by-%-6ld nnz: %6ld", "QR use COLAMD", "Analyze time:", "Factorize time:",
// "SparseQR TOTAL time:", "res = %8.1e") and appends "graph_id  Ana_time  Fac_time  total  res" to ./Results/QR_Time.txt exactly
// as qrtest.c:125-128,189-201 does.
//
// What runs where.  Everything is this library: the reader (stmmqr_read_matrix_market), singletons + COLAMD + symbolic
// analysis on the host and the numeric factorization on the device (stmmqr_sparseqr), and the residual check of
// qrtest.c:11-53 -- b = A [0..n-1], y = Q'b, x = E (R \ y) -- on the factors that stay resident in HBM
// (stmmqr_sparseqr_qmult / stmmqr_sparseqr_solve).  No reference code is needed or loaded.
//
// AMD, METIS and NESDIS are third-party ordering packages in the reference (component #8) and are not rebuilt
// here: for `ordering` 0, 2, 3 the driver needs a library built from the reference, named by --reflib=<path> or the
// environment variable STMMQR_REFERENCE_LIB; then the reference's SparseQR() runs around this library's qr_factorize
// (the drop-in seam of INTEGRATION.md: this library is loaded first with RTLD_GLOBAL, so the reference's internal call of
// qr_factorize, SparseQR.c:349,371, and qr_panel's calls of qr_larftb bind to the symbols exported here).  Without one the
// driver says so and stops with exit code 2.
//
// The graph-feature side files of the reference driver (Results/QR_Node.txt, QR_Edge.txt: input of the GCN ordering
// classifier, qrtest.c:105-109) are opened for append like the reference does and left untouched: out of scope.
#include <dlfcn.h>
#include <sys/time.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/stmmqr_hip.h"

namespace {

// dense_array (STMMQR/include/SparseCore.h:885-897)
struct ref_dense {
    size_t nrow, ncol, nzmax, d;
    void *x, *z;
    int xtype, dtype;
};
// SparseQR_factorization (STMMQR/include/SparseQR_struct.h:218-255)
struct ref_qr {
    double tol;
    void *QRsym, *QRnum;
    stm_long *R1p, *R1j;
    double *R1x;
    stm_long r1nz;
    stm_long *Q1fill, *P1inv, *HP1inv, *Rmap, *RmapInv;
    stm_long n1rows, n1cols, narows, nacols, rank;
    double Ana_time, Fac_time;
    int allow_tol;
};

template <class F> bool sym(void *h, const char *name, F &fn)
{
    fn = (F)dlsym(h, name);
    if (!fn) fprintf(stderr, "stmmqr_qrtest: the reference library does not export %s\n", name);
    return fn != nullptr;
}

double now()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec / 1000000.0;
}

}  // namespace

int main(int argc, char **argv)
{
    std::vector<char *> pos;
    const char *reflib = getenv("STMMQR_REFERENCE_LIB");
    for (int i = 1; i < argc; i++) {
        if (strncmp(argv[i], "--reflib=", 9) == 0) reflib = argv[i] + 9;
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 2) {
        fprintf(stderr, "usage: %s <matrix.mtx> <graph_id> [ordering 0 AMD | 1 COLAMD | 2 METIS | 3 NESDIS] [--reflib=<reference .so>]\n", argv[0]);
        return 1;
    }
    const char *fmatrix = pos[0];
    const int graph_id = atoi(pos[1]);

    // ---- the matrix: this library's reader (qrtest.c:112 SparseCore_read_matrix, prefer = 1) ----
    stm_long m = 0, n = 0, nnz = 0, *Ap = nullptr, *Ai = nullptr;
    double *Ax = nullptr;
    {
        FILE *fp = fopen(fmatrix, "r");
        if (!fp) { printf("%s file is not exist!\n", fmatrix); return 0; }        // (qrtest.c:85-88)
        fclose(fp);
    }
    if (stmmqr_read_matrix_market(fmatrix, &m, &n, &nnz, &Ap, &Ai, &Ax) != 0) {
        printf("input matrix must be sparse\n");                                   // (qrtest.c:114-118)
        fprintf(stderr, "stmmqr_qrtest: %s\n", stmmqr_mm_last_error());
        return 1;
    }
    { FILE *a = fopen("./Results/QR_Node.txt", "a+"), *b = fopen("./Results/QR_Edge.txt", "a+"); if (a) fclose(a); if (b) fclose(b); }
    FILE *fresult = fopen("./Results/QR_Time.txt", "a+");
    if (fresult) fprintf(fresult, "%d\t", graph_id);
    printf("Matrix %6ld-by-%-6ld nnz: %6ld\n", (long)m, (long)n, (long)nnz);

    // tol = 20 (m + n) eps max_j ||A(:,j)||_2   (qrtest.c:135-142, qr_maxcolnorm)
    double maxnorm = 0;
    for (stm_long j = 0; j < n; j++) {
        double s = 0;
        for (stm_long p = Ap[j]; p < Ap[j + 1]; p++) s += Ax[p] * Ax[p];
        maxnorm = std::max(maxnorm, std::sqrt(s));
    }
    if (maxnorm == 0) maxnorm = 1;
    const double tol = 20 * ((double)m + (double)n) * DBL_EPSILON * maxnorm;

    long ordarg = -1;
    if (pos.size() >= 3) ordarg = atoi(pos[2]);
    const bool need_reference = (ordarg == 0 || ordarg == 2 || ordarg == 3);        // AMD / METIS / NESDIS
    if (!need_reference) {
        // ---- everything on this library (qrtest.c:144-204) ----
        stmmqr_relax relax;
        stmmqr_relax_for_qr(n, nnz, &relax);                                        // Relaxfactor_setting (qrtest.c:153)
        chunk_getSettings(32, 5000, 4, 4);                                          // (qrtest.c:152)
        stmmqr_qr *QR = nullptr;
        const double t0 = now();
        printf("QR use COLAMD \n");                                                 // (SparseQR.c:935)
        const int e = stmmqr_sparseqr(7 /* QR_ORDERING_DEFAULT */, tol, m, n, Ap, Ai, Ax, nullptr, &relax, -1, &QR);
        const double t1 = now();
        if (e) { fprintf(stderr, "stmmqr_qrtest: SparseQR failed (%d): %s\n", e, stmmqr_last_error()); return 3; }
        double info[12];
        (void)stmmqr_sparseqr_info(QR, info);
        printf("\nAnalyze time: %lf\n", info[4]);                                   // (SparseQR.c:342, -DPRINT_TIME)
        printf("Factorize time: %lf\n", info[5]);                                   // (:354)
        printf("SparseQR TOTAL time: %f\n\n", t1 - t0);
        if (fresult) fprintf(fresult, "%lf\t%lf\t%lf\t", info[4], info[5], t1 - t0);
        // check_error (qrtest.c:11-53)
        std::vector<double> x0((size_t)std::max<stm_long>(n, 1)), b((size_t)std::max<stm_long>(m, 1), 0.0), y((size_t)std::max<stm_long>(m, 1)),
            xs((size_t)std::max<stm_long>(n, 1));
        for (stm_long j = 0; j < n; j++) {
            x0[(size_t)j] = (double)j;
            for (stm_long p = Ap[j]; p < Ap[j + 1]; p++) b[(size_t)Ai[p]] += Ax[p] * (double)j;
        }
        double res = NAN;
        int e2 = stmmqr_sparseqr_qmult(QR, 0 /* QR_QTX */, b.data(), std::max<stm_long>(m, 1), m, 1, y.data(), std::max<stm_long>(m, 1));
        if (!e2) e2 = stmmqr_sparseqr_solve(QR, 1 /* QR_RETX_EQUALS_B */, y.data(), std::max<stm_long>(m, 1), 1, xs.data(), std::max<stm_long>(n, 1));
        if (e2) fprintf(stderr, "stmmqr_qrtest: solve failed (%d): %s\n", e2, stmmqr_last_error());
        else {
            double d = 0;
            for (stm_long j = 0; j < n; j++) { const double q = xs[(size_t)j] - (double)j; d += q * q; }
            res = std::sqrt(d) / (double)n;
        }
        printf("res = %8.1e\n", res);
        if (fresult) { fprintf(fresult, "%8.1e\n", res); fclose(fresult); }
        if (getenv("STMMQR_QRTEST_VERBOSE"))
            printf("rank = %ld  n1rows = %ld  n1cols = %ld  nf = %ld  flops = %.6g  device ms = %.3f  GFLOP/s = %.1f\n", (long)info[0], (long)info[1],
                   (long)info[2], (long)info[3], info[6], info[8], info[8] > 0 ? info[6] / info[8] * 1e-6 : 0.0);
        stmmqr_sparseqr_free(QR);
        stmmqr_free(Ap); stmmqr_free(Ai); stmmqr_free(Ax);
        stmmqr_shutdown();
        return e2 ? 3 : 0;
    }
    // ---- AMD / METIS / NESDIS: the reference's SparseQR() around this library's seam ----
    if (!reflib) {
        fprintf(stderr, "stmmqr_qrtest: ordering %ld (0 AMD, 2 METIS, 3 NESDIS) is a third-party package of the reference and not built here;\n"
                        "  name a library built from the reference (--reflib=<path> or STMMQR_REFERENCE_LIB), or use COLAMD (1 / default).\n"
                        "  Matrix read: %ld x %ld, %ld entries, tol %.3e\n", ordarg, (long)m, (long)n, (long)nnz, tol);
        return 2;
    }
    // this library first and global, so that the reference's call of qr_factorize binds here
    Dl_info self;
    void *me = nullptr;
    if (dladdr((void *)&stmmqr_read_matrix_market, &self) && self.dli_fname) me = dlopen(self.dli_fname, RTLD_NOW | RTLD_GLOBAL);
    void *h = dlopen(reflib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "stmmqr_qrtest: cannot load %s: %s\n", reflib, dlerror()); return 2; }
    (void)me;
    int (*r_start)(void *) = nullptr, (*r_finish)(void *) = nullptr;
    stm_sparse_csc *(*r_alloc)(size_t, size_t, size_t, int, int, int, int, void *) = nullptr;
    int (*r_free_sparse)(stm_sparse_csc **, void *) = nullptr, (*r_free_dense)(ref_dense **, void *) = nullptr;
    void (*r_relax)(size_t, size_t, int, void *) = nullptr;
    ref_qr *(*r_sparseqr)(int, double, stm_sparse_csc *, void *, char *) = nullptr;
    int (*r_qrfree)(ref_qr **, void *) = nullptr;
    ref_dense *(*r_qmult)(int, ref_qr *, ref_dense *, void *) = nullptr, *(*r_solve)(int, ref_qr *, ref_dense *, void *) = nullptr;
    ref_dense *(*r_zeros)(size_t, size_t, int, void *) = nullptr;
    int (*r_sdmult)(stm_sparse_csc *, int, double *, double *, ref_dense *, ref_dense *, void *) = nullptr;
    if (!(sym(h, "SparseCore_start", r_start) && sym(h, "SparseCore_finish", r_finish) &&
          sym(h, "SparseCore_allocate_sparse", r_alloc) && sym(h, "SparseCore_free_sparse", r_free_sparse) &&
          sym(h, "SparseCore_free_dense", r_free_dense) && sym(h, "Relaxfactor_setting", r_relax) &&
          sym(h, "SparseQR", r_sparseqr) && sym(h, "SparseQR_free", r_qrfree) && sym(h, "QR_qmult", r_qmult) &&
          sym(h, "QR_solve", r_solve) && sym(h, "SparseCore_zeros", r_zeros) && sym(h, "SparseCore_sdmult", r_sdmult)))
        return 2;

    std::vector<double> ccbuf(4096, 0.0);                     // sparse_common (1248 bytes in the stock build), opaque here
    void *cc = ccbuf.data();
    r_start(cc);
    stm_common_layout lay;
    stmmqr_get_common_layout(&lay);
    // the matrix in the reference's allocator (it frees it with its own accounting): SPARSE_REAL = 1, sorted, packed
    stm_sparse_csc *A = r_alloc((size_t)m, (size_t)n, (size_t)std::max<stm_long>(nnz, 1), 1, 1, 0, 1, cc);
    if (!A) { fprintf(stderr, "stmmqr_qrtest: the reference could not allocate the matrix\n"); return 3; }
    memcpy(A->p, Ap, sizeof(stm_long) * (size_t)(n + 1));
    memcpy(A->i, Ai, sizeof(stm_long) * (size_t)nnz);
    memcpy(A->x, Ax, sizeof(double) * (size_t)nnz);
    stmmqr_free(Ap); stmmqr_free(Ai); stmmqr_free(Ax);

    // cc->SPQR_grain = 1: one task, no TPSM pool (the device path has its own scheduler; qrtest.c:145-150 sizes the pool
    // for the CPU path).  chunk_getSettings (qrtest.c:152) is this library's; Relaxfactor_setting as the driver (:153).
    *(double *)((char *)cc + lay.SPQR_grain) = 1.0;
    *(int *)((char *)cc + lay.status) = 0;
    chunk_getSettings(32, 5000, 4, 4);
    r_relax((size_t)n, (size_t)nnz, 1 /* RELAX_FOR_QR, SparseCore.h:1285 */, cc);
    long ordering = 7;                                        // QR_ORDERING_DEFAULT (SparseQR_definitions.h:6-21)
    if (pos.size() >= 3) {
        switch (atoi(pos[2])) {
            case 0: ordering = 5; break;                      // AMD
            case 1: ordering = 2; break;                      // COLAMD
            case 2: ordering = 11; break;                     // ONLYMETIS
            case 3: ordering = 6; break;                      // NESDIS
            default: ordering = 7;
        }
    }
    char name[64] = "stmmqr_qrtest";
    const double t0 = now();
    ref_qr *QR = r_sparseqr((int)ordering, tol, A, cc, name);
    const double t1 = now();
    if (!QR) { fprintf(stderr, "stmmqr_qrtest: SparseQR failed (status %d): %s\n", *(int *)((char *)cc + lay.status), stmmqr_last_error()); return 3; }
    printf("SparseQR TOTAL time: %f\n\n", t1 - t0);
    if (fresult) fprintf(fresult, "%lf\t%lf\t%lf\t", QR->Ana_time, QR->Fac_time, t1 - t0);

    // ---- check_error (qrtest.c:11-53): x = 0..n-1, b = A x, y = Q'b, x_sol = E (R \\ y) ----
    double one[2] = {1, 0}, zero[2] = {0, 0};
    ref_dense *X = r_zeros((size_t)n, 1, 1, cc), *B = r_zeros((size_t)m, 1, 1, cc);
    for (stm_long i = 0; i < n; i++) ((double *)X->x)[i] = (double)i;
    r_sdmult(A, 0, one, zero, X, B, cc);
    ref_dense *Y = r_qmult(0 /* QR_QTX */, QR, B, cc);
    ref_dense *Xs = Y ? r_solve(1 /* QR_RETX_EQUALS_B */, QR, Y, cc) : nullptr;
    double res = NAN;
    if (Xs) {
        double d = 0;
        for (stm_long j = 0; j < n; j++) { const double e = ((double *)Xs->x)[j] - (double)j; d += e * e; }
        res = std::sqrt(d) / (double)n;
    }
    printf("res = %8.1e\n", res);
    if (fresult) { fprintf(fresult, "%8.1e\n", res); fclose(fresult); }
    if (Y) r_free_dense(&Y, cc);
    if (Xs) r_free_dense(&Xs, cc);
    r_free_dense(&X, cc); r_free_dense(&B, cc);
    r_qrfree(&QR, cc);
    r_free_sparse(&A, cc);
    r_finish(cc);
    stmmqr_shutdown();
    return 0;
}
