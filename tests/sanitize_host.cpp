// Host-only pieces of the library under AddressSanitizer / UBSan (CPU build only: GPU sanitizers are not available on the pool).
// Built and run by tests/test_sanitize_host.py:
//   g++ -fsanitize=address,undefined csrc/{stmmqr_mmio,stmmqr_symbolic,stmmqr_colamd,stmmqr_sparseqr}.cpp tests/sanitize_host.cpp
// For every Matrix-Market file on the command line: the reader (stmmqr_read_matrix_market: hostile files included), the symbolic
// analysis in natural order (stmmqr_analyze) and the host half of SparseQR() (singletons, COLAMD, analysis:
// stmmqr_sparseqr_symbolic).  The device entry points stmmqr_sparseqr.cpp refers to are never called here; they are stubbed below.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/stmmqr_hip.h"

extern "C" int stm_fail(int code, const char *msg) { fprintf(stderr, "[sanitize_host] library error %d: %s\n", code, msg ? msg : ""); return code; }
// device half: not part of this build
extern "C" {
stmmqr_plan *stmmqr_plan_create(const stmmqr_symbolic_view *, int, int *st) { if (st) *st = STMMQR_ERR_DEVICE; return nullptr; }
void stmmqr_plan_destroy(stmmqr_plan *) {}
int stmmqr_factorize_device(stmmqr_plan *, const stm_long *, const stm_long *, const double *, int, double, stm_long, stmmqr_stats *) { return STMMQR_ERR_DEVICE; }
int stmmqr_plan_download(stmmqr_plan *, double *, stm_long *, char *, stm_long *, double *, stm_long *, stm_long *, stm_long *, stm_long *, stm_long *, stmmqr_stats *) { return STMMQR_ERR_DEVICE; }
int stmmqr_plan_qmult(stmmqr_plan *, int, double *, stm_long, stm_long) { return STMMQR_ERR_DEVICE; }
int stmmqr_plan_rsolve(stmmqr_plan *, int, const double *, stm_long, double *, stm_long, stm_long) { return STMMQR_ERR_DEVICE; }
}

int main(int argc, char **argv)
{
    int nread = 0, nrefused = 0, nanalyzed = 0;
    for (int a = 1; a < argc; a++) {
        stm_long m = 0, n = 0, nnz = 0, *Ap = nullptr, *Ai = nullptr;
        double *Ax = nullptr;
        const int rc = stmmqr_read_matrix_market(argv[a], &m, &n, &nnz, &Ap, &Ai, &Ax);
        if (rc != 0) { nrefused++; printf("%s: refused (%d) %s\n", argv[a], rc, stmmqr_mm_last_error()); continue; }
        nread++;
        // natural order
        stmmqr_analysis *An = nullptr;
        int e = stmmqr_analyze(m, n, Ap, Ai, nullptr, 1, nullptr, &An);
        double info[8] = {0};
        if (!e && An) { stmmqr_analysis_info(An, info); nanalyzed++; }
        stmmqr_analysis_free(An);
        // SparseQR()'s host half with the driver's knobs: singletons, COLAMD, analysis
        stmmqr_relax rx;
        stmmqr_relax_for_qr(n, nnz, &rx);
        stmmqr_qr *qr = nullptr;
        e = stmmqr_sparseqr_symbolic(7, -2.0, m, n, Ap, Ai, Ax, nullptr, &rx, &qr);
        double qi[12] = {0};
        if (!e && qr) stmmqr_sparseqr_info(qr, qi);
        stmmqr_sparseqr_free(qr);
        printf("%s: %ld x %ld, %ld entries, nf %g / %g, flop bound %g, rc %d\n", argv[a], (long)m, (long)n, (long)nnz, info[7], qi[3], info[0], e);
        stmmqr_free(Ap); stmmqr_free(Ai); stmmqr_free(Ax);
    }
    printf("read %d, refused %d, analyzed %d\n", nread, nrefused, nanalyzed);
    return 0;
}
