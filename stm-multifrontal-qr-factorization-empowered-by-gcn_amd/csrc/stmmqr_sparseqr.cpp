// stmmqr_sparseqr.cpp -- SparseQR() end to end on this library alone (SURVEY.md 8 f2 stage 2 + f4).
//
// Reference counterparts (paths relative to /root/reference/STMMQR):
//   stmmqr_sparseqr         SparseQR                src/qr/SparseQR.c:66-450   (singletons -> ordering -> analysis -> numeric)
//     find_singletons       qr_1colamd              :515-1128  (column-singleton breadth-first search, row permutation P1inv,
//                                                               pruned matrix, fill-reducing ordering of the pruned matrix)
//     order_pruned          SparseChol_colamd       src/chol/SparseChol_analyze.c:924-1080 (COLAMD, then the column
//                                                               elimination tree's post-order folded into the permutation)
//     split_r1_y            SparseQR.c:216-329      (R1 = singleton rows in row form, Y = the rest, columns in Q1fill order)
//   stmmqr_sparseqr_qmult   QR_qmult                :1838-2020 (row permutation HP1inv around the multifrontal reflectors)
//   stmmqr_sparseqr_solve   QR_solve / qr_rsolve    :2118-2517 (multifrontal rows on the device, singleton rows last :2490-2515)
// The symbolic analysis is stmmqr_analyze (stmmqr_symbolic.cpp), the numeric factorization and the Q / R operations on
// the resident factors are the device path (stmmqr_host.cpp).  Orderings built here: COLAMD (the driver's default),
// natural / fixed, and a permutation given by the caller; AMD / METIS / NESDIS (third-party packages in the reference) are
// refused with a message -- a caller that has such a permutation passes it as QR_ORDERING_GIVEN.
#include <sys/time.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/stmmqr_hip.h"
#include "stmmqr_internal.h"

bool stm_colamd_order(stm_long n_row, stm_long n_col, const stm_long *Ap, const stm_long *Ai, std::vector<stm_long> &perm);

namespace {

typedef stm_long Long;
const Long NONE = -1;

double wall()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec / 1e6;
}

// column elimination tree of A(:, perm) and its (unweighted) post-order, folded into perm  (SparseChol_colamd :1045-1075)
void postorder_columns(Long m, Long n, const Long *Ap, const Long *Ai, std::vector<Long> &perm)
{
    std::vector<Long> parent((size_t)std::max<Long>(n, 1), NONE), anc((size_t)std::max<Long>(n, 1), NONE), prevc((size_t)std::max<Long>(m, 1), NONE);
    for (Long k = 0; k < n; k++) {
        const Long j = perm[(size_t)k];
        for (Long p = Ap[j]; p < Ap[j + 1]; p++) {
            const Long i = Ai[p];
            Long c = prevc[(size_t)i];
            prevc[(size_t)i] = k;
            while (c != NONE && c != k) {
                const Long a = anc[(size_t)c];
                if (a == k) break;
                anc[(size_t)c] = k;
                if (a == NONE) { parent[(size_t)c] = k; break; }
                c = a;
            }
        }
    }
    // children in increasing index, depth first from every root in index order
    std::vector<Long> head((size_t)std::max<Long>(n, 1), NONE), next((size_t)std::max<Long>(n, 1), NONE), stack((size_t)std::max<Long>(n, 1)), post;
    for (Long j = n - 1; j >= 0; j--)
        if (parent[(size_t)j] != NONE) { next[(size_t)j] = head[(size_t)parent[(size_t)j]]; head[(size_t)parent[(size_t)j]] = j; }
    post.reserve((size_t)n);
    for (Long r = 0; r < n; r++) {
        if (parent[(size_t)r] != NONE) continue;
        Long top = 0;
        stack[0] = r;
        while (top >= 0) {
            const Long p = stack[(size_t)top], c = head[(size_t)p];
            if (c == NONE) { top--; post.push_back(p); }
            else { head[(size_t)p] = next[(size_t)c]; stack[(size_t)++top] = c; }
        }
    }
    std::vector<Long> np((size_t)std::max<Long>(n, 1));
    for (Long k = 0; k < n; k++) np[(size_t)k] = perm[(size_t)post[(size_t)k]];
    perm.swap(np);
}

}  // namespace

struct stmmqr_qr {
    Long m = 0, n = 0, n1rows = 0, n1cols = 0, rank = 0;
    double tol = 0;
    int ordering_used = 0;
    std::vector<Long> Q1fill, P1inv, HP1inv, R1p, R1j;        // P1inv / HP1inv / R1*: only when n1cols > 0
    std::vector<double> R1x;
    std::vector<Long> Yp, Yi;                                 // the matrix handed to the numeric phase when singletons exist
    std::vector<double> Yx;
    stmmqr_analysis *sym = nullptr;
    stmmqr_plan *plan = nullptr;
    stmmqr_stats stats = {};
    std::vector<char> Rdead;                                  // of the multifrontal part (n - n1cols)
    double ana_seconds = 0, fac_seconds = 0, sym_info[8] = {};
    const Long *Fp = nullptr, *Fi = nullptr;                  // what the numeric phase factorizes (A or Y)
    const double *Fx = nullptr;
    ~stmmqr_qr()
    {
        if (plan) stmmqr_plan_destroy(plan);
        if (sym) stmmqr_analysis_free(sym);
    }
};

extern "C" {

static int numeric_impl(stmmqr_qr *QR, int device);

static int sparseqr_impl(int ordering, double tol, Long m, Long n, const Long *Ap, const Long *Ai, const double *Ax, const Long *Quser,
                         const stmmqr_relax *relax, int device, int numeric, stmmqr_qr **out)
{
    if (!out) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: null output");
    *out = nullptr;
    if (m < 0 || n < 0 || !Ap || (Ap[n] > 0 && (!Ai || !Ax))) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: bad matrix");
    // orderings (SparseQR_definitions.h:6-21)
    enum { FIXED = 0, NATURAL = 1, COLAMD = 2, GIVEN = 3, CHOL = 4, AMD = 5, NESDIS = 6, DEFAULT = 7, BEST = 8, BESTAMD = 9, METIS = 10, ONLYMETIS = 11 };
    if (ordering == DEFAULT) ordering = COLAMD;                                                   // (SparseQR.c:869-890)
    if (ordering == GIVEN && !Quser) ordering = FIXED;
    if (!(ordering == FIXED || ordering == NATURAL || ordering == COLAMD || ordering == GIVEN))
        return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: this ordering (AMD / METIS / NESDIS / best-of) is not built here; compute the "
                                            "permutation outside and pass it as QR_ORDERING_GIVEN (3)");
    const bool fill_reducing = (ordering == COLAMD);
    std::unique_ptr<stmmqr_qr> QR(new stmmqr_qr());
    QR->m = m; QR->n = n;
    if (tol <= -2) {
        // QR_DEFAULT_TOL: 20 (m + n) eps max_j |A(:,j)|_2  (qr_tol / qr_maxcolnorm, SparseQR.c:126-130,1134-1144,1376-1420)
        double mx = 0;
        for (Long j = 0; j < n; j++) {
            // (dnrm2's scaled form: the largest entry carries the magnitude)
            double scale = 0, ssq = 1;
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) {
                const double a = std::fabs(Ax[p]);
                if (a == 0) continue;
                if (scale < a) { ssq = 1 + ssq * (scale / a) * (scale / a); scale = a; }
                else ssq += (a / scale) * (a / scale);
            }
            mx = std::max(mx, scale * std::sqrt(ssq));
        }
        tol = 20.0 * ((double)m + (double)n) * DBL_EPSILON * mx;
        tol = std::min(tol, DBL_MAX);
    }
    if (tol < 0) tol = -1;                                                                        // QR_NO_TOL (SparseQR.c:131-135)
    QR->tol = tol;
    const double t_start = wall();

    // ---- column singletons (qr_1colamd :565-700): a column with one entry above tol in a live row is a singleton; removing
    //      its row may create new ones (breadth-first) ----
    std::vector<Long> &Q1 = QR->Q1fill;
    Q1.assign((size_t)std::max<Long>(n, 1), 0);
    std::vector<Long> degree((size_t)std::max<Long>(n, 1), 0), qrow((size_t)std::max<Long>(n, 1), NONE);
    Long n1cols = 0, n1rows = 0;
    const bool given = (ordering == GIVEN);
    if (!given) {
        for (Long j = 0; j < n; j++) {
            const Long p = Ap[j], d = Ap[j + 1] - p;
            if (d == 0) { Q1[(size_t)n1cols] = j; qrow[(size_t)n1cols++] = NONE; degree[(size_t)j] = NONE; }
            else if (d == 1 && std::fabs(Ax[p]) > tol) { Q1[(size_t)n1cols] = j; qrow[(size_t)n1cols++] = Ai[p]; degree[(size_t)j] = NONE; }
            else degree[(size_t)j] = d;
        }
    } else {
        // (a permutation given by the caller is used as it is: SparseQR.c keeps the singleton search for every ordering, but a
        //  GIVEN permutation there only passes through qr_analyze -- SparseQR.c:338; the driver never uses it)
        for (Long j = 0; j < n; j++) degree[(size_t)j] = std::max<Long>(Ap[j + 1] - Ap[j], 1);
    }
    // A' in row form (columns of every row in increasing order)
    std::vector<Long> ATp((size_t)m + 2, 0), ATj((size_t)std::max<Long>(Ap[n], 1));
    for (Long p = 0; p < Ap[n]; p++) {
        if (Ai[p] < 0 || Ai[p] >= m) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: row index out of range");
        ATp[(size_t)Ai[p] + 1]++;
    }
    for (Long i = 0; i < m; i++) ATp[(size_t)i + 1] += ATp[(size_t)i];
    {
        std::vector<Long> w(ATp.begin(), ATp.end() - 1);
        for (Long j = 0; j < n; j++)
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) ATj[(size_t)w[(size_t)Ai[p]]++] = j;
    }
    std::vector<char> row_dead((size_t)std::max<Long>(m, 1), 0);
    for (Long k = 0; k < n1cols; k++) {
        Long row = qrow[(size_t)k];
        if (row == NONE || row_dead[(size_t)row]) { qrow[(size_t)k] = NONE; continue; }          // a dead singleton
        n1rows++;
        row_dead[(size_t)row] = 1;
        for (Long p = ATp[(size_t)row]; p < ATp[(size_t)row + 1]; p++) {
            const Long j = ATj[(size_t)p];
            Long d = degree[(size_t)j];
            if (d == NONE) continue;
            degree[(size_t)j] = --d;
            if (d == 0) { Q1[(size_t)n1cols] = j; qrow[(size_t)n1cols++] = NONE; degree[(size_t)j] = NONE; }
            else if (d == 1) {
                for (Long p2 = Ap[j]; p2 < Ap[j + 1]; p2++) {
                    const Long i = Ai[p2];
                    if (!row_dead[(size_t)i] && std::fabs(Ax[p2]) > tol) {
                        Q1[(size_t)n1cols] = j; qrow[(size_t)n1cols++] = i; degree[(size_t)j] = NONE;
                        break;
                    }
                }
            }
        }
    }
    // ---- row permutation: singleton rows first, in the order they were found (:720-766) ----
    if (n1cols > 0) {
        QR->P1inv.assign((size_t)std::max<Long>(m, 1), 0);
        QR->R1p.assign((size_t)n1rows + 1, 0);
        Long kk = 0;
        for (Long k = 0; k < n1cols; k++) {
            const Long i = qrow[(size_t)k];
            if (i == NONE) continue;
            QR->P1inv[(size_t)i] = kk;
            QR->R1p[(size_t)kk] = ATp[(size_t)i + 1] - ATp[(size_t)i];
            kk++;
        }
        for (Long i = 0; i < m; i++)
            if (!row_dead[(size_t)i]) QR->P1inv[(size_t)i] = kk++;
    }
    // ---- the rest of the column order (:770-1050) ----
    if (given) {
        for (Long k = 0; k < n; k++) Q1[(size_t)k] = Quser[k];
    } else if (!fill_reducing) {
        if (n1cols == 0) { for (Long k = 0; k < n; k++) Q1[(size_t)k] = k; }
        else { Long k = n1cols; for (Long j = 0; j < n; j++) if (degree[(size_t)j] > 0) Q1[(size_t)k++] = j; }
    } else {
        // pruned matrix: live rows and non-singleton columns, both in their original order
        std::vector<Long> cmap((size_t)std::max<Long>(n, 1), NONE), cinv, rmap((size_t)std::max<Long>(m, 1), NONE);
        Long n2 = 0, m2 = 0;
        for (Long j = 0; j < n; j++)
            if (n1cols == 0 || degree[(size_t)j] > 0) { cmap[(size_t)j] = n2++; cinv.push_back(j); }
        for (Long i = 0; i < m; i++)
            if (!row_dead[(size_t)i]) rmap[(size_t)i] = m2++;
        std::vector<Long> Pp((size_t)n2 + 1, 0), Pi;
        Pi.reserve((size_t)Ap[n]);
        for (Long c = 0; c < n2; c++) {
            const Long j = cinv[(size_t)c];
            const size_t at = Pi.size();
            for (Long p = Ap[j]; p < Ap[j + 1]; p++)
                if (rmap[(size_t)Ai[p]] != NONE) Pi.push_back(rmap[(size_t)Ai[p]]);
            std::sort(Pi.begin() + (long)at, Pi.end());                  // (the reference's transposes deliver sorted columns)
            Pp[(size_t)c + 1] = (Long)Pi.size();
        }
        if (Pi.empty()) Pi.push_back(0);
        std::vector<Long> perm;
        if (!stm_colamd_order(m2, n2, Pp.data(), Pi.data(), perm)) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: duplicate or unsorted entries in a column");
        postorder_columns(m2, n2, Pp.data(), Pi.data(), perm);
        for (Long k = 0; k < n2; k++) Q1[(size_t)(n1cols + k)] = cinv[(size_t)perm[(size_t)k]];
    }
    QR->n1cols = n1cols; QR->n1rows = n1rows;
    QR->ordering_used = ordering;

    // ---- R1 (singleton rows, row form) and Y (what is left; SparseQR.c:216-329) ----
    const Long n2 = n - n1cols, m2 = m - n1rows;
    const Long *Fp = Ap, *Fi = Ai;
    const double *Fx = Ax;
    const Long *Fq = Q1.data();
    if (n1cols > 0) {
        std::vector<Long> &R1p = QR->R1p;
        {   // counts -> pointers (qr_cumsum)
            Long tot = 0;
            for (Long i = 0; i < n1rows; i++) { const Long c = R1p[(size_t)i]; R1p[(size_t)i] = tot; tot += c; }
            R1p[(size_t)n1rows] = tot;
            QR->R1j.assign((size_t)std::max<Long>(tot, 1), 0);
            QR->R1x.assign((size_t)std::max<Long>(tot, 1), 0.0);
        }
        std::vector<Long> fill(R1p.begin(), R1p.end());
        QR->Yp.assign((size_t)n2 + 1, 0);
        for (Long k = 0; k < n1cols; k++) {
            const Long j = Q1[(size_t)k];
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) {
                const Long inew = QR->P1inv[(size_t)Ai[p]];
                if (inew >= n1rows) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: internal (singleton column reaches a live row)");
                const Long q = fill[(size_t)inew]++;
                QR->R1j[(size_t)q] = k; QR->R1x[(size_t)q] = Ax[p];
            }
        }
        for (Long k = n1cols; k < n; k++) {
            const Long j = Q1[(size_t)k];
            QR->Yp[(size_t)(k - n1cols)] = (Long)QR->Yi.size();
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) {
                const Long inew = QR->P1inv[(size_t)Ai[p]];
                if (inew < n1rows) { const Long q = fill[(size_t)inew]++; QR->R1j[(size_t)q] = k; QR->R1x[(size_t)q] = Ax[p]; }
                else { QR->Yi.push_back(inew - n1rows); QR->Yx.push_back(Ax[p]); }
            }
        }
        QR->Yp[(size_t)n2] = (Long)QR->Yi.size();
        if (QR->Yi.empty()) { QR->Yi.push_back(0); QR->Yx.push_back(0.0); }
        Fp = QR->Yp.data(); Fi = QR->Yi.data(); Fx = QR->Yx.data(); Fq = nullptr;          // analysis of Y: ordering FIXED (:360)
    }

    // ---- symbolic analysis (qr_analyze) ----
    int e = stmmqr_analyze(n1cols > 0 ? m2 : m, n1cols > 0 ? n2 : n, Fp, Fi, Fq, tol >= 0, relax, &QR->sym);
    if (e) return e;
    (void)stmmqr_analysis_info(QR->sym, QR->sym_info);
    QR->ana_seconds = wall() - t_start;

    // the matrix the numeric phase factorizes: A itself (borrowed until stmmqr_sparseqr_numeric returns) or Y (owned)
    QR->Fp = Fp; QR->Fi = Fi; QR->Fx = Fx;
    if (numeric) {
        e = numeric_impl(QR.get(), device);
        if (e) return e;
    }
    *out = QR.release();
    return 0;
}

static int numeric_impl(stmmqr_qr *QR, int device)
{
    if (QR->plan) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_numeric: already factorized");
    const Long m = QR->m, n = QR->n, n1cols = QR->n1cols, n1rows = QR->n1rows, n2 = n - n1cols;
    const Long *Fp = QR->Fp, *Fi = QR->Fi;
    const double *Fx = QR->Fx;
    const double tol = QR->tol;
    int e = 0;
    // ---- numeric factorization on the device (qr_factorize: the Fac_time interval of SparseQR.c:346-377) ----
    const double t_fac = wall();
    const stm_qr_symbolic *S = stmmqr_analysis_symbolic(QR->sym);
    stmmqr_symbolic_view V;
    memset(&V, 0, sizeof V);
    V.m = S->m; V.n = S->n; V.anz = S->anz; V.nf = S->nf; V.maxfn = S->maxfn; V.rjsize = S->rjsize; V.hisize = S->hisize;
    V.do_rank_detection = S->do_rank_detection;
    V.Sp = S->Sp; V.Sj = S->Sj; V.Qfill = S->Qfill; V.PLinv = S->PLinv; V.Sleft = S->Sleft; V.Child = S->Child; V.Childp = S->Childp;
    V.Super = S->Super; V.Rp = S->Rp; V.Rj = S->Rj; V.Post = S->Post; V.Hip = S->Hip; V.Fm = S->Fm; V.maxstack = S->maxstack;
    int st = 0;
    QR->plan = stmmqr_plan_create(&V, device, &st);
    if (!QR->plan) return st ? st : STMMQR_ERR_DEVICE;
    const Long ntol = (n1cols > 0) ? n2 : n;                                              // (SparseQR.c:349 / :371)
    e = stmmqr_factorize_device(QR->plan, Fp, Fi, Fx, 0, tol, ntol, &QR->stats);
    if (e) return e;
    Long scal[4] = {0, 0, 0, 0};
    QR->Rdead.assign((size_t)std::max<Long>(S->n, 1), 0);
    std::vector<Long> HPinv((size_t)std::max<Long>(S->m, 1));
    e = stmmqr_plan_download(QR->plan, nullptr, nullptr, QR->Rdead.data(), nullptr, nullptr, nullptr, HPinv.data(), nullptr, nullptr, scal, nullptr);
    if (e) return e;
    QR->fac_seconds = wall() - t_fac;
    QR->rank = n1rows + scal[1];                                                            // n1rows + rank1 (SparseQR.c:393)
    if (n1cols > 0) {
        QR->HP1inv.assign((size_t)std::max<Long>(m, 1), 0);
        for (Long i = 0; i < m; i++) {
            const Long k = QR->P1inv[(size_t)i];
            QR->HP1inv[(size_t)i] = (k < n1rows) ? k : HPinv[(size_t)(k - n1rows)] + n1rows;
        }
    }
    return 0;
}

int stmmqr_sparseqr(int ordering, double tol, stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                    const stm_long *Quser, const stmmqr_relax *relax, int device, stmmqr_qr **out)
{
    try {
        return sparseqr_impl(ordering, tol, m, n, Ap, Ai, Ax, Quser, relax, device, 1, out);
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_sparseqr: out of memory");
    } catch (...) {
        return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr: internal error");
    }
}

/* the host half alone (singletons, ordering, R1 / Y, symbolic analysis): no device is touched.  A's arrays must stay valid
 * until stmmqr_sparseqr_numeric has run (when no singletons were found the numeric phase reads A itself). */
int stmmqr_sparseqr_symbolic(int ordering, double tol, stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                             const stm_long *Quser, const stmmqr_relax *relax, stmmqr_qr **out)
{
    try {
        return sparseqr_impl(ordering, tol, m, n, Ap, Ai, Ax, Quser, relax, -1, 0, out);
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_sparseqr_symbolic: out of memory");
    } catch (...) {
        return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_symbolic: internal error");
    }
}

int stmmqr_sparseqr_numeric(stmmqr_qr *qr, int device)
{
    if (!qr) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_numeric: null argument");
    try {
        return numeric_impl(qr, device);
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_sparseqr_numeric: out of memory");
    }
}

void stmmqr_sparseqr_free(stmmqr_qr *qr) { delete qr; }

int stmmqr_sparseqr_info(const stmmqr_qr *qr, double *info)
{
    if (!qr || !info) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_info: null argument");
    info[0] = (double)qr->rank; info[1] = (double)qr->n1rows; info[2] = (double)qr->n1cols; info[3] = qr->sym_info[7];
    info[4] = qr->ana_seconds; info[5] = qr->fac_seconds; info[6] = qr->stats.flops; info[7] = qr->sym_info[0];
    info[8] = qr->stats.ms_total; info[9] = (double)qr->ordering_used; info[10] = qr->sym_info[3]; info[11] = (double)qr->stats.retries;
    return 0;
}

/* the tolerance the factorization really used: qr_tol(A) for tol <= -2 (QR_DEFAULT_TOL), EMPTY (-1) for any other negative one
 * (SparseQR.c:126-139: what the reference stores in SparseQR_factorization::tol) */
double stmmqr_sparseqr_tol(const stmmqr_qr *qr) { return qr ? qr->tol : -1.0; }
const stm_long *stmmqr_sparseqr_q1fill(const stmmqr_qr *qr) { return qr ? qr->Q1fill.data() : nullptr; }
const stm_qr_symbolic *stmmqr_sparseqr_symbolic_view(const stmmqr_qr *qr) { return qr && qr->sym ? stmmqr_analysis_symbolic(qr->sym) : nullptr; }
stmmqr_plan *stmmqr_sparseqr_plan(stmmqr_qr *qr) { return qr ? qr->plan : nullptr; }
/* the matrix handed to the numeric phase when singletons were removed (NULL pointers otherwise: it was A itself) */
int stmmqr_sparseqr_y(const stmmqr_qr *qr, const stm_long **Yp, const stm_long **Yi, const double **Yx)
{
    if (!qr) return stm_fail(STMMQR_ERR_INVALID, "null argument");
    const bool has = qr->n1cols > 0;
    if (Yp) *Yp = has ? qr->Yp.data() : nullptr;
    if (Yi) *Yi = has ? qr->Yi.data() : nullptr;
    if (Yx) *Yx = has ? qr->Yx.data() : nullptr;
    return 0;
}

// ---- QR_qmult (SparseQR.c:1838-2020): X is nrow x ncol (ldx), Y gets the same shape (ldy) ----
int stmmqr_sparseqr_qmult(stmmqr_qr *qr, int method, const double *X, stm_long ldx, stm_long nrow, stm_long ncol, double *Y, stm_long ldy)
{
    if (!qr || !X || !Y || method < 0 || method > 3) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_qmult: bad arguments");
    const Long m = qr->m, n1 = qr->n1rows;
    const bool left = method <= 1;
    if ((left ? nrow : ncol) != m) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_qmult: X does not match the rows of A");
    if (ldx < nrow || ldy < nrow) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_qmult: bad leading dimension");
    try {
        const Long m2 = m - n1;
        const bool sing = qr->n1cols > 0;
        auto pinv = [&](Long i) -> Long { return sing ? qr->P1inv[(size_t)i] : i; };         // row of [singleton rows; Y]
        if (left) {
            // the rows of Y's part as one m2 x ncol block: Q'X takes them in Y's row order and returns R's row order, Q X the reverse
            std::vector<double> Z((size_t)std::max<Long>(m2, 1) * (size_t)std::max<Long>(ncol, 1));
            if (method == 0) {
                for (Long j = 0; j < ncol; j++)
                    for (Long i = 0; i < m; i++) {
                        const Long k = pinv(i);
                        if (k < n1) Y[k + j * ldy] = X[i + j * ldx]; else Z[(size_t)(k - n1) + (size_t)j * (size_t)m2] = X[i + j * ldx];
                    }
                if (m2 > 0 && ncol > 0) { const int e = stmmqr_plan_qmult(qr->plan, 0, Z.data(), m2, ncol); if (e) return e; }
                for (Long j = 0; j < ncol; j++)
                    for (Long r = 0; r < m2; r++) Y[n1 + r + j * ldy] = Z[(size_t)r + (size_t)j * (size_t)m2];
            } else {
                for (Long j = 0; j < ncol; j++)
                    for (Long r = 0; r < m2; r++) Z[(size_t)r + (size_t)j * (size_t)m2] = X[n1 + r + j * ldx];
                if (m2 > 0 && ncol > 0) { const int e = stmmqr_plan_qmult(qr->plan, 1, Z.data(), m2, ncol); if (e) return e; }
                for (Long j = 0; j < ncol; j++)
                    for (Long i = 0; i < m; i++) {
                        const Long k = pinv(i);
                        Y[i + j * ldy] = (k < n1) ? X[k + j * ldx] : Z[(size_t)(k - n1) + (size_t)j * (size_t)m2];
                    }
            }
        } else {
            // X Q' = (Q X')', X Q = (Q' X')': the same with rows and columns exchanged (X is nrow x m)
            std::vector<double> Z((size_t)std::max<Long>(nrow, 1) * (size_t)std::max<Long>(m2, 1));
            if (method == 3) {
                for (Long i = 0; i < m; i++) {
                    const Long k = pinv(i);
                    for (Long r = 0; r < nrow; r++) {
                        if (k < n1) Y[r + k * ldy] = X[r + i * ldx]; else Z[(size_t)r + (size_t)(k - n1) * (size_t)nrow] = X[r + i * ldx];
                    }
                }
                if (m2 > 0 && nrow > 0) { const int e = stmmqr_plan_qmult(qr->plan, 3, Z.data(), nrow, nrow); if (e) return e; }
                for (Long c = 0; c < m2; c++)
                    for (Long r = 0; r < nrow; r++) Y[r + (n1 + c) * ldy] = Z[(size_t)r + (size_t)c * (size_t)nrow];
            } else {
                for (Long c = 0; c < m2; c++)
                    for (Long r = 0; r < nrow; r++) Z[(size_t)r + (size_t)c * (size_t)nrow] = X[r + (n1 + c) * ldx];
                if (m2 > 0 && nrow > 0) { const int e = stmmqr_plan_qmult(qr->plan, 2, Z.data(), nrow, nrow); if (e) return e; }
                for (Long i = 0; i < m; i++) {
                    const Long k = pinv(i);
                    for (Long r = 0; r < nrow; r++) Y[r + i * ldy] = (k < n1) ? X[r + k * ldx] : Z[(size_t)r + (size_t)(k - n1) * (size_t)nrow];
                }
            }
        }
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_sparseqr_qmult: out of memory");
    }
    return 0;
}

// ---- QR_solve (SparseQR.c:2118-2216): systems 0 R X = B, 1 R E' X = B (B: m x nrhs in R's row order, X: n x nrhs),
//      2 R' X = B, 3 R' X = E' B (B: n x nrhs, X: m x nrhs) ----
int stmmqr_sparseqr_solve(stmmqr_qr *qr, int system, const double *B, stm_long ldb, stm_long nrhs, double *X, stm_long ldx)
{
    if (!qr || !B || !X || system < 0 || system > 3 || nrhs < 0) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_solve: bad arguments");
    const Long m = qr->m, n = qr->n, n1r = qr->n1rows, n1c = qr->n1cols, n2 = n - n1c, m2 = m - n1r;
    const Long brows = system <= 1 ? m : n, xrows = system <= 1 ? n : m;
    if (ldb < brows || ldx < xrows) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparseqr_solve: bad leading dimension");
    if (nrhs == 0) return 0;
    try {
        if (n1c == 0) return stmmqr_plan_rsolve(qr->plan, system, B, ldb, X, ldx, nrhs);   // (the plan's Qfill is Q1fill)
        const bool useE = (system == 1 || system == 3);
        auto colmap = [&](Long k) -> Long { return useE ? qr->Q1fill[(size_t)k] : k; };
        if (system <= 1) {
            // multifrontal rows first (qr_rsolve :2290-2470), singleton rows last (:2490-2515)
            std::vector<double> Z((size_t)std::max<Long>(m2, 1) * (size_t)nrhs), X2((size_t)std::max<Long>(n2, 1) * (size_t)nrhs, 0.0);
            for (Long j = 0; j < nrhs; j++)
                for (Long r = 0; r < m2; r++) Z[(size_t)r + (size_t)j * (size_t)m2] = B[n1r + r + j * ldb];
            if (n2 > 0) { const int e = stmmqr_plan_rsolve(qr->plan, 0, Z.data(), std::max<Long>(m2, 1), X2.data(), std::max<Long>(n2, 1), nrhs); if (e) return e; }
            for (Long j = 0; j < nrhs; j++) {
                double *x = X + j * ldx;
                for (Long k = 0; k < n; k++) x[k] = 0;
                for (Long k = 0; k < n2; k++) x[colmap(n1c + k)] = X2[(size_t)k + (size_t)j * (size_t)std::max<Long>(n2, 1)];
                for (Long i = n1r - 1; i >= 0; i--) {
                    double v = B[i + j * ldb];
                    for (Long p = qr->R1p[(size_t)i] + 1; p < qr->R1p[(size_t)i + 1]; p++) v -= qr->R1x[(size_t)p] * x[colmap(qr->R1j[(size_t)p])];
                    const Long p = qr->R1p[(size_t)i];
                    x[colmap(qr->R1j[(size_t)p])] = v / qr->R1x[(size_t)p];
                }
            }
            return 0;
        }
        // R' systems with singleton rows: forward through the singleton rows, then the multifrontal part.  Rank-deficient
        // factorizations (qr_private_rtsolve's "squeezed" R, SparseQR.c:2522-2700: Rmap / RmapInv): a dead column has no equation.
        // Dead columns only exist in the multifrontal part (a singleton's diagonal passed the tolerance test to become one), so
        // the singleton stage is the same -- what it subtracts from a dead column's entry of b is never read -- and the plan's
        // R' solve skips the dead columns of its own part; the solution's rows are the singleton rows, then the live
        // multifrontal rows in order, then zeros: the reference's layout.
        std::vector<double> b((size_t)n), B2((size_t)std::max<Long>(n2, 1) * (size_t)nrhs), X2((size_t)std::max<Long>(m2, 1) * (size_t)nrhs, 0.0);
        for (Long j = 0; j < nrhs; j++) {
            for (Long k = 0; k < n; k++) b[(size_t)k] = B[colmap(k) + j * ldb];             // b in R's column order
            double *x = X + j * ldx;
            for (Long i = 0; i < m; i++) x[i] = 0;
            for (Long i = 0; i < n1r; i++) {
                const Long p0 = qr->R1p[(size_t)i];
                const double xi = b[(size_t)qr->R1j[(size_t)p0]] / qr->R1x[(size_t)p0];
                x[i] = xi;
                for (Long p = p0 + 1; p < qr->R1p[(size_t)i + 1]; p++) b[(size_t)qr->R1j[(size_t)p]] -= qr->R1x[(size_t)p] * xi;
            }
            for (Long k = 0; k < n2; k++) B2[(size_t)k + (size_t)j * (size_t)std::max<Long>(n2, 1)] = b[(size_t)(n1c + k)];
        }
        if (n2 > 0 && m2 > 0) { const int e = stmmqr_plan_rsolve(qr->plan, 2, B2.data(), std::max<Long>(n2, 1), X2.data(), std::max<Long>(m2, 1), nrhs); if (e) return e; }
        for (Long j = 0; j < nrhs; j++)
            for (Long r = 0; r < m2; r++) X[n1r + r + j * ldx] = X2[(size_t)r + (size_t)j * (size_t)std::max<Long>(m2, 1)];
        return 0;
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_sparseqr_solve: out of memory");
    }
}

}  // extern "C"
