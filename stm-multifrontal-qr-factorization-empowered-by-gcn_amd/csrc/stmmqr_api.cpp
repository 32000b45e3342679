// stmmqr_api.cpp -- libstmmqr_hip_api.so: the reference's OUTER entry points under the reference's own names.
//
//     SparseQR_factorization *SparseQR (int ordering, double tol, sparse_csc *A, sparse_common *cc, char *result_name)
//     int SparseQR_free (SparseQR_factorization **QR, sparse_common *cc)
//     dense_array *QR_qmult (int method, SparseQR_factorization *QR, dense_array *X, sparse_common *cc)
//     dense_array *QR_solve (int system, SparseQR_factorization *QR, dense_array *B, sparse_common *cc)
//     double qr_maxcolnorm (sparse_csc *A, sparse_common *cc)
//     int TPSM_init (int, int, int, int) / int TPSM_destroy (int)
// (prototypes STMMQR/include/SparseQR.h:25-36,403-417,433; structs SparseQR_struct.h:218-255, SparseCore.h:885-897; the pool's
// entry points include/tpsm/tpsm_main.h).  A program written against the reference's headers -- its own driver test/qrtest.c,
// unmodified -- links this library and libstmmqr_hip.so in place of the reference's QR module (src/qr/*) and thread pool; what
// it still takes from the reference is the sparse-matrix toolbox it calls itself (SparseCore_read_matrix, SparseCore_sdmult, ...).
// The test infrastructure builds exactly that program and tests/test_outer_api.py runs it.
//
// An OPT-IN second shared object: libstmmqr_hip.so itself does not export these names, so the other integration -- the
// reference's own SparseQR.o around this library's qr_factorize -- gets no duplicate symbols.
//
// What the returned SparseQR_factorization holds: the fields a caller of the public API reads (tol, the sizes, rank, Ana_time,
// Fac_time, Q1fill, allow_tol) and QRsym (borrowed view of this library's analysis, layout of qr_symbolic).  QRnum is NULL:
// the numeric factors stay in HBM, owned by the handle behind the struct; QR_qmult / QR_solve run there.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/stmmqr_hip.h"

extern "C" {

typedef struct dense_array_struct {          // SparseCore.h:885-897
    size_t nrow, ncol, nzmax, d;
    void *x, *z;
    int xtype, dtype;
} dense_array;

typedef struct SparseQR_factorization_struct {   // SparseQR_struct.h:218-255
    double tol;
    stm_qr_symbolic *QRsym;
    stm_qr_numeric *QRnum;
    stm_long *R1p, *R1j;
    double *R1x;
    stm_long r1nz;
    stm_long *Q1fill, *P1inv, *HP1inv, *Rmap, *RmapInv;
    stm_long n1rows, n1cols, narows, nacols, rank;
    double Ana_time, Fac_time;
    int allow_tol;
} SparseQR_factorization;

#define API_MAGIC 0x53544d4d51524150ULL
struct ApiQR {                                   // what SparseQR() really allocates: the public struct first
    SparseQR_factorization pub;
    unsigned long long magic;
    stmmqr_qr *impl;
};

static ApiQR *own(SparseQR_factorization *QR)
{
    ApiQR *a = reinterpret_cast<ApiQR *>(QR);
    return (a && a->magic == API_MAGIC) ? a : nullptr;
}

double qr_maxcolnorm(stm_sparse_csc *A, stm_sparse_common *cc)
{
    (void)cc;
    if (!A || !A->p || !A->x) return 0;
    const stm_long *Ap = (const stm_long *)A->p;
    const double *Ax = (const double *)A->x;
    double mx = 0;
    for (size_t j = 0; j < A->ncol; j++) {
        double s = 0;
        for (stm_long p = Ap[j]; p < Ap[j + 1]; p++) s += Ax[p] * Ax[p];
        mx = std::fmax(mx, std::sqrt(s));
    }
    return mx;
}

SparseQR_factorization *SparseQR(int ordering, double tol, stm_sparse_csc *A, stm_sparse_common *cc, char *result_name)
{
    (void)result_name;                           // (the reference writes graph files for its GCN classifier under that name)
    if (!A || !A->p || !A->i || !A->x) { stmmqr_cc_set_status(cc, STMMQR_ERR_INVALID); return nullptr; }
    if (A->stype != 0 || A->xtype != 1) {        // SparseQR.c:98-110: unsymmetric real input
        stmmqr_cc_set_status(cc, STMMQR_ERR_INVALID);
        return nullptr;
    }
    const stm_long m = (stm_long)A->nrow, n = (stm_long)A->ncol;
    const stm_long *Ap = (const stm_long *)A->p;
    // the driver sets the amalgamation knobs in cc right before the call (Relaxfactor_setting (n, nnz, RELAX_FOR_QR, cc),
    // qrtest.c:153); sparse_common is opaque here, so the same rule is applied directly
    stmmqr_relax rx;
    stmmqr_relax_for_qr(n, Ap[n], &rx);
    stmmqr_qr *impl = nullptr;
    const int e = stmmqr_sparseqr(ordering, tol, m, n, Ap, (const stm_long *)A->i, (const double *)A->x, nullptr, &rx, -1, &impl);
    if (e || !impl) {
        if (impl) stmmqr_sparseqr_free(impl);
        stmmqr_cc_set_status(cc, e ? e : STMMQR_ERR_DEVICE);
        fprintf(stderr, "SparseQR (libstmmqr_hip_api): %s\n", stmmqr_last_error());
        return nullptr;
    }
    ApiQR *a = (ApiQR *)stmmqr_cc_malloc(1, sizeof(ApiQR), cc);
    if (!a) { stmmqr_sparseqr_free(impl); return nullptr; }
    memset(a, 0, sizeof *a);
    a->magic = API_MAGIC;
    a->impl = impl;
    double info[12] = {0};
    (void)stmmqr_sparseqr_info(impl, info);
    SparseQR_factorization &Q = a->pub;
    Q.tol = stmmqr_sparseqr_tol(impl);              // the EFFECTIVE tolerance, as the reference stores it (SparseQR.c:126-139)
    Q.QRsym = const_cast<stm_qr_symbolic *>(stmmqr_sparseqr_symbolic_view(impl));
    Q.QRnum = nullptr;
    Q.Q1fill = const_cast<stm_long *>(stmmqr_sparseqr_q1fill(impl));
    Q.rank = (stm_long)info[0]; Q.n1rows = (stm_long)info[1]; Q.n1cols = (stm_long)info[2];
    Q.narows = m; Q.nacols = n;
    Q.Ana_time = info[4]; Q.Fac_time = info[5];
    Q.allow_tol = Q.tol >= 0;
    return &a->pub;
}

int SparseQR_free(SparseQR_factorization **QR, stm_sparse_common *cc)
{
    if (!QR || !*QR) return 1;
    ApiQR *a = own(*QR);
    if (!a) return 0;                            // not ours (a factorization made by the reference's SparseQR)
    if (a->impl) stmmqr_sparseqr_free(a->impl);
    a->magic = 0;
    stmmqr_cc_free(1, sizeof(ApiQR), a, cc);
    *QR = nullptr;
    return 1;
}

static dense_array *new_dense(size_t nrow, size_t ncol, stm_sparse_common *cc)       // SparseCore_allocate_dense, d = nrow
{
    dense_array *X = (dense_array *)stmmqr_cc_malloc(1, sizeof(dense_array), cc);
    if (!X) return nullptr;
    memset(X, 0, sizeof *X);
    X->nrow = nrow; X->ncol = ncol; X->d = nrow;
    X->nzmax = nrow * ncol > 0 ? nrow * ncol : 1;
    X->xtype = 1; X->dtype = 0;                  // SPARSE_REAL, SPARSE_DOUBLE
    X->x = stmmqr_cc_malloc(X->nzmax, sizeof(double), cc);
    if (!X->x) { stmmqr_cc_free(1, sizeof(dense_array), X, cc); return nullptr; }
    memset(X->x, 0, X->nzmax * sizeof(double));
    return X;
}
static void free_dense(dense_array *X, stm_sparse_common *cc)
{
    if (!X) return;
    stmmqr_cc_free(X->nzmax, sizeof(double), X->x, cc);
    stmmqr_cc_free(1, sizeof(dense_array), X, cc);
}

dense_array *QR_qmult(int method, SparseQR_factorization *QR, dense_array *X, stm_sparse_common *cc)
{
    ApiQR *a = own(QR);
    if (!a || !X || !X->x || X->xtype != 1 || method < 0 || method > 3) { stmmqr_cc_set_status(cc, STMMQR_ERR_INVALID); return nullptr; }
    const size_t m = (size_t)QR->narows;
    if ((method <= 1 && X->nrow != m) || (method >= 2 && X->ncol != m)) {             // SparseQR.c:1838-1852
        stmmqr_cc_set_status(cc, STMMQR_ERR_INVALID);
        return nullptr;
    }
    dense_array *Y = new_dense(X->nrow, X->ncol, cc);
    if (!Y) return nullptr;
    const int e = stmmqr_sparseqr_qmult(a->impl, method, (const double *)X->x, (stm_long)X->d, (stm_long)X->nrow, (stm_long)X->ncol,
                                        (double *)Y->x, (stm_long)Y->d);
    if (e) { free_dense(Y, cc); stmmqr_cc_set_status(cc, e); return nullptr; }
    return Y;
}

dense_array *QR_solve(int system, SparseQR_factorization *QR, dense_array *B, stm_sparse_common *cc)
{
    ApiQR *a = own(QR);
    if (!a || !B || !B->x || B->xtype != 1 || system < 0 || system > 3) { stmmqr_cc_set_status(cc, STMMQR_ERR_INVALID); return nullptr; }
    const size_t m = (size_t)QR->narows, n = (size_t)QR->nacols;
    if (B->nrow != (system <= 1 ? m : n)) { stmmqr_cc_set_status(cc, STMMQR_ERR_INVALID); return nullptr; }   // SparseQR.c:2140-2150
    dense_array *X = new_dense(system <= 1 ? n : m, B->ncol, cc);
    if (!X) return nullptr;
    const int e = stmmqr_sparseqr_solve(a->impl, system, (const double *)B->x, (stm_long)B->d, (stm_long)B->ncol, (double *)X->x, (stm_long)X->d);
    if (e) { free_dense(X, cc); stmmqr_cc_set_status(cc, e); fprintf(stderr, "QR_solve (libstmmqr_hip_api): %s\n", stmmqr_last_error()); return nullptr; }
    return X;
}

// The reference's thread pool (TPSM: SURVEY.md 8 rows a13 / a14) has no counterpart to start: tree parallelism is the step
// timeline on HIP streams inside the library.  The driver's calls succeed and do nothing.
int TPSM_init(int pool_size, int buffer_size, int sync_size, int affinity_mode) { (void)pool_size; (void)buffer_size; (void)sync_size; (void)affinity_mode; return 0; }
int TPSM_destroy(int mode) { (void)mode; return 0; }

}  // extern "C"
