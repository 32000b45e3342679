/* oracle/refdump.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Golden-vector generator: drives the REAL reference (oracle/_ref/libstmmqr_ref.so, built by
 * oracle/Makefile from the sources under /root/reference/STMMQR) exactly like its own driver
 * does (STMMQR/test/qrtest.c:65-217) and records everything that crosses the hot-path seam
 *
 *     qr_numeric *qr_factorize (sparse_csc **Ahandle, Long freeA, double tol, Long ntol,
 *                               qr_symbolic *QRsym, sparse_common *cc)
 *                               (STMMQR/include/SparseQR.h:127-135, call sites SparseQR.c:349,371)
 *
 * The seam is captured by symbol interposition: this executable defines qr_factorize itself, so the
 * call inside the reference's SparseQR() binds here; we copy the inputs, forward to the real
 * function (dlsym RTLD_NEXT) and serialise inputs + outputs into a tagged binary file that
 * tests/golden/make_golden.py turns into .npz fixtures.
 *
 * usage: refdump <matrix.mtx> <ordering -1|0..3> <grain> <tolmode d|n> <out.bin|-> [reps]
 *   ordering: -1 default(COLAMD) 0 AMD 1 COLAMD 2 METIS 3 NESDIS   (qrtest.c:155-169)
 *   grain   : cc->SPQR_grain; 1 = serial qr_kernel(0) (STMMQR/README.md:71-72)
 *   tolmode : d = driver default tol = 20(m+n)eps*maxcolnorm (qrtest.c:135-142), n = no rank detection (tol=-1)
 *   threads : env REFDUMP_POOL (pool size for TPSM_init when grain>1; default 64)
 *   env REFDUMP_HIPLIB=<path to libstmmqr_hip.so>: the interposed seam calls THAT library's qr_factorize
 *   (drop-in check: the reference's SparseQR / QR_qmult / QR_solve / SparseQR_free run on its result)
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>
#include "SparseQR.h"
#include "tpsm.h"
#include "tpsm_sysinfo.h"

#ifndef Long
#define Long Sparse_long
#endif

static FILE *g_out = NULL;
static void *g_hiplib = NULL;
static double g_fac_seconds = 0;
static double g_flops = 0;

static void put(const char *name, char ty, long count, const void *data)
{
    if (!g_out) return;
    char tag[32];
    memset(tag, 0, sizeof tag);
    strncpy(tag, name, 31);
    fwrite(tag, 1, 32, g_out);
    fwrite(&ty, 1, 1, g_out);
    fwrite(&count, sizeof(long), 1, g_out);
    size_t es = (ty == 'b') ? 1 : 8;
    if (count > 0 && data) fwrite(data, es, (size_t)count, g_out);
    else if (count > 0) { /* NULL array: emit zeros so the reader stays in sync */
        void *z = calloc((size_t)count, es); fwrite(z, es, (size_t)count, g_out); free(z);
    }
}
static void put_l(const char *name, long v) { put(name, 'q', 1, &v); }
static void put_d(const char *name, double v) { put(name, 'd', 1, &v); }

static double now(void)
{
    struct timeval tv; gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec / 1e6;
}

typedef qr_numeric *(*factorize_fn)(sparse_csc **, Long, double, Long, qr_symbolic *, sparse_common *);

/* Interposed seam. */
qr_numeric *qr_factorize(sparse_csc **Ahandle, Long freeA, double tol, Long ntol,
                         qr_symbolic *QRsym, sparse_common *cc)
{
    static factorize_fn real = NULL;
    if (!real && getenv("REFDUMP_HIPLIB")) {
        /* drop-in check: route the seam to libstmmqr_hip.so's qr_factorize; everything around it
         * (SparseQR, QR_qmult, QR_solve, SparseQR_free) stays the compiled reference */
        void *h = dlopen(getenv("REFDUMP_HIPLIB"), RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "refdump: cannot load %s: %s\n", getenv("REFDUMP_HIPLIB"), dlerror()); exit(2); }
        g_hiplib = h;
        real = (factorize_fn)dlsym(h, "qr_factorize");
        printf("seam routed to %s\n", getenv("REFDUMP_HIPLIB"));
    }
    if (!real) real = (factorize_fn)dlsym(RTLD_NEXT, "qr_factorize");
    if (!real) { fprintf(stderr, "refdump: cannot find the reference qr_factorize\n"); exit(2); }

    sparse_csc *A = *Ahandle;
    Long m = (Long)A->nrow, n = (Long)A->ncol;
    Long *Ap = (Long *)A->p;
    Long anz = Ap[n];
    /* inputs (A may be freed by the callee when freeA) */
    put_l("in_m", m); put_l("in_n", n); put_l("in_freeA", freeA); put_l("in_ntol", ntol);
    put_d("in_tol", tol);
    put("in_Ap", 'q', n + 1, Ap);
    put("in_Ai", 'q', anz, A->i);
    put("in_Ax", 'd', anz, A->x);
    put_l("FCHUNK", (long)FCHUNK); put_l("SMALL", (long)SMALL);
    put_l("MINCHUNK", (long)MINCHUNK); put_l("MINCHUNK_RATIO", (long)MINCHUNK_RATIO);
    put_d("SPQR_grain", cc->SPQR_grain); put_d("SPQR_small", cc->SPQR_small);
    put_l("SPQR_shrink", cc->SPQR_shrink);

    /* symbolic object (borrowed, never modified by the callee) */
    qr_symbolic *S = QRsym;
    Long nf = S->nf;
    put_l("sym_m", S->m); put_l("sym_n", S->n); put_l("sym_anz", S->anz); put_l("sym_nf", nf);
    put_l("sym_maxfn", S->maxfn); put_l("sym_rjsize", S->rjsize);
    put_l("sym_do_rank_detection", S->do_rank_detection); put_l("sym_maxstack", S->maxstack);
    put_l("sym_hisize", S->hisize); put_l("sym_keepH", S->keepH);
    put_l("sym_ntasks", S->ntasks); put_l("sym_ns", S->ns);
    put("sym_Sp", 'q', S->m + 1, S->Sp);
    put("sym_Sj", 'q', S->anz, S->Sj);
    put("sym_Qfill", 'q', S->Qfill ? S->n : 0, S->Qfill);
    put("sym_PLinv", 'q', S->m, S->PLinv);
    put("sym_Sleft", 'q', S->n + 2, S->Sleft);
    put("sym_Parent", 'q', nf + 1, S->Parent);
    put("sym_Child", 'q', nf + 1, S->Child);
    put("sym_Childp", 'q', nf + 2, S->Childp);
    put("sym_Super", 'q', nf + 1, S->Super);
    put("sym_Rp", 'q', nf + 1, S->Rp);
    put("sym_Rj", 'q', S->rjsize, S->Rj);
    put("sym_Post", 'q', nf + 1, S->Post);
    put("sym_Hip", 'q', nf + 1, S->Hip);
    put("sym_Fm", 'q', nf + 1, S->Fm);
    put("sym_Cm", 'q', nf + 1, S->Cm);
    if (S->ntasks > 1) {
        Long nt = S->ntasks;
        put("sym_TaskChildp", 'q', nt + 2, S->TaskChildp);
        put("sym_TaskChild", 'q', nt + 1, S->TaskChild);
        put("sym_TaskStack", 'q', nt + 1, S->TaskStack);
        put("sym_TaskFront", 'q', nf + 1, S->TaskFront);
        put("sym_TaskFrontp", 'q', nt + 2, S->TaskFrontp);
        put("sym_On_stack", 'q', nf + 1, S->On_stack);
        put("sym_Stack_maxstack", 'q', S->ns, S->Stack_maxstack);
    }
    put_d("flopcount_bound", cc->SPQR_flopcount_bound);

    double t0 = now();
    qr_numeric *N = real(Ahandle, freeA, tol, ntol, QRsym, cc);
    g_fac_seconds = now() - t0;
    g_flops = cc->SPQR_flopcount;
    put_d("fac_seconds", g_fac_seconds);
    put_d("flopcount", cc->SPQR_flopcount);
    put_l("status", cc->status);
    if (!N) { put_l("num_null", 1); return N; }
    put_l("num_null", 0);
    put_l("num_rank", N->rank); put_l("num_rank1", N->rank1); put_l("num_maxfrank", N->maxfrank);
    put_l("num_maxfm", N->maxfm); put_l("num_ns", N->ns); put_l("num_ntasks", N->ntasks);
    put("num_Rdead", 'b', S->n, N->Rdead);
    put("num_HStair", 'q', N->rjsize, N->HStair);
    put("num_HTau", 'd', N->rjsize, N->HTau);
    put("num_Hii", 'q', N->hisize, N->Hii);
    put("num_HPinv", 'q', S->m, N->HPinv);
    put("num_Hm", 'q', nf, N->Hm);
    put("num_Hr", 'q', nf, N->Hr);
    put("num_Stack_size", 'q', N->ns, N->Stack_size);
    /* Rblock as (stack id, offset) pairs */
    {
        Long *rs = malloc(sizeof(Long) * (nf > 0 ? nf : 1)), *ro = malloc(sizeof(Long) * (nf > 0 ? nf : 1));
        for (Long f = 0; f < nf; f++) {
            rs[f] = -1; ro[f] = -1;
            for (Long s = 0; s < N->ns; s++) {
                double *b = N->Stacks[s];
                if (N->Rblock[f] >= b && N->Rblock[f] <= b + N->Stack_size[s]) { rs[f] = s; ro[f] = N->Rblock[f] - b; break; }
            }
        }
        put("num_Rblock_stack", 'q', nf, rs);
        put("num_Rblock_off", 'q', nf, ro);
        free(rs); free(ro);
    }
    for (Long s = 0; s < N->ns; s++) {
        char nm[32]; snprintf(nm, sizeof nm, "num_Stack_%ld", (long)s);
        put(nm, 'd', N->Stack_size[s], N->Stacks[s]);
    }
    return N;
}

/* the reference driver's acceptance check (qrtest.c:11-53), restated */
static double solve_residual(sparse_csc *A, SparseQR_factorization *QR, sparse_common *cc, double *backward)
{
    double one[2] = {1, 0}, zero[2] = {0, 0}, minusone[2] = {-1, 0};
    Long n = A->ncol;
    dense_array *X = SparseCore_zeros(n, 1, A->xtype, cc);
    dense_array *B = SparseCore_zeros(A->nrow, 1, A->xtype, cc);
    double *x = (double *)X->x;
    for (Long i = 0; i < n; i++) x[i] = (double)i;
    SparseCore_sdmult(A, 0, one, zero, X, B, cc);
    double tq0 = now();
    dense_array *Y = QR_qmult(QR_QTX, QR, B, cc);
    double tq1 = now();
    dense_array *Xs = QR_solve(QR_RETX_EQUALS_B, QR, Y, cc);
    double tq2 = now();
    printf("REF qmult(QTX) seconds: %.6f   solve(RETX_EQUALS_B) seconds: %.6f\n", tq1 - tq0, tq2 - tq1);
    double *xs = (double *)Xs->x, d = 0;
    for (Long j = 0; j < n; j++) { double e = xs[j] - (double)j; d += e * e; }
    put("solve_x", 'd', n, xs);
    /* backward error ||A xs - b|| / (||A||_F ||xs|| + ||b||) */
    {
        dense_array *Rr = SparseCore_zeros(A->nrow, 1, A->xtype, cc);
        double *r = (double *)Rr->x, *b = (double *)B->x;
        SparseCore_sdmult(A, 0, one, zero, Xs, Rr, cc);
        double rn = 0, bn = 0, xn = 0, an = 0;
        for (Long i = 0; i < (Long)A->nrow; i++) { rn += (r[i] - b[i]) * (r[i] - b[i]); bn += b[i] * b[i]; }
        for (Long j = 0; j < n; j++) xn += xs[j] * xs[j];
        Long *Ap = (Long *)A->p; double *Ax = (double *)A->x;
        for (Long p = 0; p < Ap[n]; p++) an += Ax[p] * Ax[p];
        *backward = sqrt(rn) / (sqrt(an) * sqrt(xn) + sqrt(bn));
        SparseCore_free_dense(&Rr, cc);
    }
    (void)minusone;
    SparseCore_free_dense(&Y, cc); SparseCore_free_dense(&X, cc);
    SparseCore_free_dense(&B, cc); SparseCore_free_dense(&Xs, cc);
    return sqrt(d) / (double)n;
}

#include <execinfo.h>
#include <signal.h>
static void on_segv(int sig)
{
    /* say where: the TPSM mode (grain > 1) of the reference as built here dies inside the pool set-up on hosts that do not look
     * like the machine its checked-in include/tpsm/Numainfo.h describes (2 sockets, 4 NUMA nodes, 128 cores) */
    void *bt[48];
    int n = backtrace(bt, 48);
    fprintf(stderr, "refdump: signal %d\n", sig);
    backtrace_symbols_fd(bt, n, 2);
    _exit(128 + sig);
}

int main(int argc, char **argv)
{
    signal(SIGSEGV, on_segv);
    if (argc < 6) {
        fprintf(stderr, "usage: %s <matrix.mtx> <ordering> <grain> <tolmode d|n> <out.bin|-> [reps]\n", argv[0]);
        return 1;
    }
    const char *path = argv[1];
    int ordsel = atoi(argv[2]);
    double grain = atof(argv[3]);
    char tolmode = argv[4][0];
    const char *outp = argv[5];
    int reps = argc > 6 ? atoi(argv[6]) : 1;

    sparse_common Common, *cc = &Common;
    SparseCore_start(cc);
    FILE *fp = fopen(path, "r");
    if (!fp) { fprintf(stderr, "cannot open %s\n", path); return 1; }
    FILE *devnull1 = fopen("/dev/null", "a+"), *devnull2 = fopen("/dev/null", "a+");
    int mtype;
    sparse_csc *A = (sparse_csc *)SparseCore_read_matrix(fp, 1, &mtype, cc, devnull1, devnull2, 0);
    fclose(fp); fclose(devnull1); fclose(devnull2);
    if (!A || mtype != SPARSE_CSC) { fprintf(stderr, "input must be sparse\n"); return 1; }
    Long m = A->nrow, n = A->ncol;
    printf("Matrix %6ld-by-%-6ld nnz: %6ld\n", (long)m, (long)n, (long)SparseCore_nnz(A, cc));

    double tol;
    if (tolmode == 'n') tol = -1;
    else {
        double mx = qr_maxcolnorm(A, cc);
        if (mx == 0) mx = 1;
        tol = 20 * ((double)m + (double)n) * DBL_EPSILON * mx;
    }
    cc->SPQR_grain = grain;
    cc->status = SPARSE_OK;
    int pool = getenv("REFDUMP_POOL") ? atoi(getenv("REFDUMP_POOL")) : 64;
    if (grain > 1) TPSM_init(pool, 2000, 3000, TPSM_NODE_AFFINITY);
    Relaxfactor_setting(n, SparseCore_nnz(A, cc), RELAX_FOR_QR, cc);

    long ordering;
    switch (ordsel) {
        case 0: ordering = QR_ORDERING_AMD; break;
        case 1: ordering = QR_ORDERING_COLAMD; break;
        case 2: ordering = QR_ORDERING_ONLYMETIS; break;
        case 3: ordering = QR_ORDERING_NESDIS; break;
        default: ordering = QR_ORDERING_DEFAULT;
    }

    double best = 1e300, ana = 0;
    SparseQR_factorization *QR = NULL;
    for (int r = 0; r < reps; r++) {
        int last = (r == reps - 1);
        if (last && strcmp(outp, "-") != 0) {
            g_out = fopen(outp, "wb");
            if (!g_out) { fprintf(stderr, "cannot write %s\n", outp); return 1; }
            put_l("A_m", m); put_l("A_n", n);
            put("A_p", 'q', n + 1, A->p);
            put("A_i", 'q', ((Long *)A->p)[n], A->i);
            put("A_x", 'd', ((Long *)A->p)[n], A->x);
            put_l("ordering", ordering);
        }
        chunk_getSettings(32, 5000, 4, 4);      /* qrtest.c:152; qr_analyze may raise to 80/8000 */
        char name[64] = "refdump";
        QR = SparseQR((int)ordering, tol, A, cc, name);
        if (!QR) { fprintf(stderr, "SparseQR failed, status %d\n", cc->status); return 3; }
        if (g_fac_seconds < best) best = g_fac_seconds;
        ana = QR->Ana_time;
        if (!last) SparseQR_free(&QR, cc);
    }
    if (grain > 1) TPSM_destroy(TPSM_SHUTDOWN_GENTLY);

    put_l("n1rows", QR->n1rows); put_l("n1cols", QR->n1cols); put_l("QR_rank", QR->rank);
    put_d("QR_tol", QR->tol);
    put_d("ana_seconds", ana);
    put_d("best_fac_seconds", best);
    double bwd = 0;
    double res = solve_residual(A, QR, cc, &bwd);
    put_d("res", res); put_d("backward_err", bwd);
    printf("nf = %ld ntasks = %ld rank = %ld flops = %.17g (bound %.17g)\n", (long)QR->QRsym->nf,
           (long)QR->QRnum->ntasks, (long)QR->rank, g_flops, cc->SPQR_flopcount_bound);
    printf("REF factorize seconds (best of %d): %.6f   GFLOP/s: %.3f\n", reps, best,
           g_flops > 0 ? g_flops / best * 1e-9 : 0.0);
    printf("res = %8.1e  backward = %8.1e\n", res, bwd);
    if (g_out) fclose(g_out);
    SparseQR_free(&QR, cc);
    SparseCore_free_sparse(&A, cc);
    SparseCore_finish(cc);
    if (getenv("REFDUMP_HIPLIB") && g_hiplib) {
        /* the drop-in library offers an explicit end-of-use call (include/stmmqr_hip.h: stmmqr_shutdown); a host program
         * that dlopen()ed it calls that before it returns from main, then exits normally */
        void (*shut)(void) = (void (*)(void))dlsym(g_hiplib, "stmmqr_shutdown");
        if (shut && !getenv("REFDUMP_NO_SHUTDOWN")) shut();
    }
    return 0;
}
