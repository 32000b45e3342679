// stmmqr_export.cpp -- R / H out in the reference's sparse format, and LQ.  Host code: the functions turn
// the packed R+H blocks a numeric factorization returned (qr_numeric: this library's qr_factorize or the reference's own)
// into compressed sparse columns; they move data and count it -- no arithmetic -- so "results identical" means bit for bit
// (tests/test_export.py: the reference's own qr_rcount / qr_rconvert / qr_trapezoidal outputs on the same objects).
//
// Reference counterparts (STMMQR/src/qr/SparseLQ.c, prototypes STMMQR/include/SparseQR.h:282-340), exported under the
// reference's names so that a relinked reference can drop SparseLQ.o as well:
//   qr_rcount      :102-297   entries per column of R (split at column n2 into Ra | Rb, Rb optionally by row = transposed)
//                             and of H (one column per live reflector), from the packed blocks
//   qr_rconvert    :299-517   the same walk, filling row indices and values; H rows are the permuted row ids (Hii) + n1rows
//   qr_trapezoidal :519-689   column permutation that puts the columns whose last entry is on the diagonal first
//                             (R -> [R1 R2], R1 upper triangular), with the permutation composed into Qfill
//   stmmqr_sparselq  SparseLQ :691-734   LQ of A = QR of A' (the reference returns exactly that object)
// The walk over a front's packed columns is the one qr_rhpack wrote (SparseQR_factorize.c:1691-1784).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/stmmqr_hip.h"
#include "stmmqr_internal.h"

namespace {

typedef stm_long Long;

// Visits every stored entry of every front in packed order: r(f-row-in-R, column j, value) for the R part, and for each
// column with a Householder vector h_begin(column slot) then h(row-in-front, value) for its entries below the diagonal.
template <class FR, class FHB, class FH>
void walk_packed(const stm_qr_symbolic *S, const stm_qr_numeric *N, Long n1rows, FR &&on_r, FHB &&on_h_begin, FH &&on_h)
{
    const Long nf = S->nf;
    const bool keepH = N->keepH != 0;
    Long row1 = n1rows;
    for (Long f = 0; f < nf; f++) {
        const double *R = N->Rblock[f];
        const Long col1 = S->Super[f], fp = S->Super[f + 1] - col1, pr = S->Rp[f], fn = S->Rp[f + 1] - pr;
        const Long *Stair = keepH ? N->HStair + pr : nullptr;
        const double *Tau = keepH ? N->HTau + pr : nullptr;
        const Long fm = keepH ? N->Hm[f] : 0;
        Long rm = 0, h = 0, t = 0;
        for (Long k = 0; k < fn; k++) {
            Long j;
            if (k < fp) {
                j = col1 + k;
                if (keepH) {
                    t = Stair[k];
                    if (t == 0) t = rm;                       // dead pivot column: R part only
                    else if (rm < fm) rm++;                   // live pivot: one more row of R
                    h = rm;
                } else if (!N->Rdead[j]) rm++;
            } else {
                j = S->Rj[pr + k];
                if (keepH) { t = Stair[k]; h = std::min(h + 1, fm); }
            }
            for (Long i = 0; i < rm; i++) on_r(f, row1 + i, j, *R++);
            if (keepH && t >= h) {
                if (Tau[k] != 0.0 && on_h_begin(f, k, h, Tau[k])) {
                    for (Long i = h; i < t; i++) on_h(f, i, *R++);
                } else R += (t - h);
            }
        }
        row1 += rm;
    }
}

}  // namespace

extern "C" {

void qr_rcount(stm_qr_symbolic *S, stm_qr_numeric *N, stm_long n1rows, stm_long econ, stm_long n2, int getT, stm_long *Ra,
               stm_long *Rb, stm_long *H2p, stm_long *p_nh)
{
    if (!S || !N) return;
    const bool getRa = Ra != nullptr, getRb = Rb != nullptr, getH = H2p && p_nh && N->keepH;
    if (!(getRa || getRb || getH)) return;
    Long nh = 0, hnz = 0;
    walk_packed(S, N, n1rows,
                [&](Long, Long row, Long j, double v) {
                    if (v == 0.0 || row >= econ) return;
                    if (j < n2) { if (getRa) Ra[j]++; }
                    else if (getRb) { if (getT) Rb[row]++; else Rb[j - n2]++; }
                },
                [&](Long, Long, Long, double) -> bool {
                    if (!getH) return false;
                    H2p[nh++] = hnz++;                        // (the unit diagonal is an entry of H)
                    return true;
                },
                [&](Long, Long, double v) { if (v != 0.0) hnz++; });
    if (getH) { H2p[nh] = hnz; *p_nh = nh; }
}

void qr_rconvert(stm_qr_symbolic *S, stm_qr_numeric *N, stm_long n1rows, stm_long econ, stm_long n2, int getT, stm_long *Rap,
                 stm_long *Rai, double *Rax, stm_long *Rbp, stm_long *Rbi, double *Rbx, stm_long *H2p, stm_long *H2i, double *H2x,
                 double *H2Tau)
{
    (void)getT;                                               // (the reference fills Rb by column whatever getT says, :428-433)
    if (!S || !N) return;
    const bool getRa = Rap && Rai && Rax, getRb = Rbp && Rbi && Rbx, getH = H2p && H2i && H2x && H2Tau && N->keepH;
    if (!(getRa || getRb || getH)) return;
    Long nh = 0, ph = 0;
    const Long *Hi = nullptr;
    walk_packed(S, N, n1rows,
                [&](Long, Long row, Long j, double v) {
                    if (v == 0.0 || row >= econ) return;
                    if (j < n2) { if (getRa) { const Long p = Rap[j]++; Rai[p] = row; Rax[p] = v; } }
                    else if (getRb) { const Long p = Rbp[j - n2]++; Rbi[p] = row; Rbx[p] = v; }
                },
                [&](Long f, Long, Long h, double tau) -> bool {
                    if (!getH) return false;
                    Hi = N->Hii + S->Hip[f];
                    H2Tau[nh++] = tau;
                    H2i[ph] = Hi[h - 1] + n1rows;
                    H2x[ph] = 1.0;
                    ph++;
                    return true;
                },
                [&](Long, Long i, double v) {
                    if (v != 0.0) { H2i[ph] = Hi[i] + n1rows; H2x[ph] = v; ph++; }
                });
}

stm_long qr_trapezoidal(stm_long n, stm_long *Rp, stm_long *Ri, double *Rx, stm_long bncols, stm_long *Qfill, int skip_if_trapezoidal,
                        stm_long **p_Tp, stm_long **p_Ti, double **p_Tx, stm_long **p_Qtrap, stm_sparse_common *cc)
{
    if (!p_Tp || !p_Ti || !p_Tx || !p_Qtrap || !Rp) return -1;
    *p_Tp = nullptr; *p_Ti = nullptr; *p_Tx = nullptr; *p_Qtrap = nullptr;
    // a column "lives" when its last entry sits on the next diagonal position
    Long rank = 0, t1nz = 0;
    bool found_dead = false, trapezoidal = true;
    for (Long k = 0; k < n; k++) {
        const Long len = Rp[k + 1] - Rp[k];
        const Long i = len > 0 ? Ri[Rp[k + 1] - 1] : -1;
        if (i > rank) return -1;                              // not upper triangular with a leading staircase
        if (i == rank) { rank++; t1nz += len; if (found_dead) trapezoidal = false; }
        else found_dead = true;
    }
    if (trapezoidal && skip_if_trapezoidal) return rank;
    const Long rnz = Rp[n];
    Long *Tp = (Long *)stm_cc_malloc((size_t)n + 1, sizeof(Long), cc), *Ti = (Long *)stm_cc_malloc((size_t)rnz, sizeof(Long), cc);
    double *Tx = (double *)stm_cc_malloc((size_t)rnz, sizeof(double), cc);
    Long *Qt = (Long *)stm_cc_malloc((size_t)(n + bncols), sizeof(Long), cc);
    if (!Tp || !Ti || !Tx || !Qt) {
        stm_cc_free((size_t)n + 1, sizeof(Long), Tp, cc); stm_cc_free((size_t)rnz, sizeof(Long), Ti, cc);
        stm_cc_free((size_t)rnz, sizeof(double), Tx, cc); stm_cc_free((size_t)(n + bncols), sizeof(Long), Qt, cc);
        return -1;
    }
    Long k1 = 0, k2 = rank, p1 = 0, p2 = t1nz;
    rank = 0;
    for (Long k = 0; k < n; k++) {
        const Long len = Rp[k + 1] - Rp[k];
        const Long i = len > 0 ? Ri[Rp[k + 1] - 1] : -1;
        Long &kd = (i == rank) ? k1 : k2, &pd = (i == rank) ? p1 : p2;
        if (i == rank) rank++;
        Tp[kd] = pd;
        Qt[kd] = Qfill ? Qfill[k] : k;
        kd++;
        for (Long p = Rp[k]; p < Rp[k + 1]; p++) { Ti[pd] = Ri[p]; Tx[pd] = Rx[p]; pd++; }
    }
    for (Long k = n; k < n + bncols; k++) Qt[k] = Qfill ? Qfill[k] : k;
    Tp[n] = rnz;
    *p_Tp = Tp; *p_Ti = Ti; *p_Tx = Tx; *p_Qtrap = Qt;
    return rank;
}

// R (and optionally H) of the factorization a plan holds, as compressed sparse columns: the device results are downloaded
// once and converted.  Rp (n + 1), Ri / Rx (nnz) and Hp (nh + 1), Hi / Hx (nnz), HTau (nh) are malloc'ed: stmmqr_free.  Any of
// the H outputs may be NULL.  R's rows are numbered as the reference numbers them (row1 counts the live pivots front by
// front), its columns are the columns of the factorized matrix in its own order (apply Qfill for A's columns).
int stmmqr_plan_export_r(stmmqr_plan *plan, const stm_qr_symbolic *S, stm_long econ, stm_long **Rp_out, stm_long **Ri_out,
                         double **Rx_out, stm_long *nh_out, stm_long **Hp_out, stm_long **Hi_out, double **Hx_out, double **HTau_out)
{
    if (!plan || !S || !Rp_out || !Ri_out || !Rx_out) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_plan_export_r: null argument");
    try {
        stm_long rh_total = 0, rank = 0;
        int e = stmmqr_plan_result_sizes(plan, &rh_total, &rank);
        if (e) return e;
        const Long nf = S->nf, n = S->n, m = S->m;
        std::vector<double> Stack((size_t)std::max<Long>(rh_total, 1)), HTau((size_t)std::max<Long>(S->rjsize, 1));
        std::vector<Long> off((size_t)std::max<Long>(nf, 1)), HStair((size_t)std::max<Long>(S->rjsize, 1)), Hii((size_t)std::max<Long>(S->hisize, 1)),
            HPinv((size_t)std::max<Long>(m, 1)), Hm((size_t)std::max<Long>(nf, 1)), Hr((size_t)std::max<Long>(nf, 1));
        std::vector<char> Rdead((size_t)std::max<Long>(n, 1));
        Long scal[4];
        e = stmmqr_plan_download(plan, Stack.data(), off.data(), Rdead.data(), HStair.data(), HTau.data(), Hii.data(), HPinv.data(), Hm.data(),
                                 Hr.data(), scal, nullptr);
        if (e) return e;
        std::vector<double *> Rblock((size_t)std::max<Long>(nf, 1));
        for (Long f = 0; f < nf; f++) Rblock[(size_t)f] = Stack.data() + off[(size_t)f];
        stm_qr_numeric N;
        memset(&N, 0, sizeof N);
        N.Rblock = Rblock.data(); N.keepH = 1; N.Rdead = Rdead.data(); N.HStair = HStair.data(); N.HTau = HTau.data(); N.Hii = Hii.data();
        N.Hm = Hm.data(); N.Hr = Hr.data(); N.nf = nf; N.n = n; N.m = m;
        const bool wantH = nh_out && Hp_out && Hi_out && Hx_out && HTau_out;
        Long *Rp = (Long *)calloc((size_t)n + 1, sizeof(Long));
        Long *Hp = wantH ? (Long *)calloc((size_t)std::max<Long>(S->rjsize, 1) + 1, sizeof(Long)) : nullptr;
        if (!Rp || (wantH && !Hp)) { free(Rp); free(Hp); return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_plan_export_r: out of memory"); }
        Long nh = 0;
        qr_rcount(const_cast<stm_qr_symbolic *>(S), &N, 0, econ, n, 0, Rp, nullptr, Hp, wantH ? &nh : nullptr);
        Long tot = 0;
        for (Long j = 0; j < n; j++) { const Long c = Rp[j]; Rp[j] = tot; tot += c; }
        Rp[n] = tot;
        Long *Ri = (Long *)malloc(sizeof(Long) * (size_t)std::max<Long>(tot, 1));
        double *Rx = (double *)malloc(sizeof(double) * (size_t)std::max<Long>(tot, 1));
        const Long hnz = wantH ? Hp[nh] : 0;
        Long *Hi = wantH ? (Long *)malloc(sizeof(Long) * (size_t)std::max<Long>(hnz, 1)) : nullptr;
        double *Hx = wantH ? (double *)malloc(sizeof(double) * (size_t)std::max<Long>(hnz, 1)) : nullptr;
        double *Ht = wantH ? (double *)malloc(sizeof(double) * (size_t)std::max<Long>(nh, 1)) : nullptr;
        if (!Ri || !Rx || (wantH && (!Hi || !Hx || !Ht))) {
            free(Rp); free(Hp); free(Ri); free(Rx); free(Hi); free(Hx); free(Ht);
            return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_plan_export_r: out of memory");
        }
        std::vector<Long> fill(Rp, Rp + n);
        qr_rconvert(const_cast<stm_qr_symbolic *>(S), &N, 0, econ, n, 0, fill.data(), Ri, Rx, nullptr, nullptr, nullptr, wantH ? Hp : nullptr, Hi, Hx, Ht);
        *Rp_out = Rp; *Ri_out = Ri; *Rx_out = Rx;
        if (wantH) { *nh_out = nh; *Hp_out = Hp; *Hi_out = Hi; *Hx_out = Hx; *HTau_out = Ht; }
        return 0;
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_plan_export_r: out of memory");
    }
}

// SparseLQ (SparseLQ.c:691-734): the LQ factorization of A is the QR factorization of A' (L = R'); the reference returns the
// SparseQR object of the transposed matrix, and so does this.
int stmmqr_sparselq(int ordering, double tol, stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                    const stmmqr_relax *relax, int device, stmmqr_qr **out)
{
    if (!Ap || m < 0 || n < 0 || !out) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparselq: bad arguments");
    try {
        const Long nz = Ap[n];
        std::vector<Long> Tp((size_t)m + 1, 0), Ti((size_t)std::max<Long>(nz, 1));
        std::vector<double> Tx((size_t)std::max<Long>(nz, 1));
        for (Long p = 0; p < nz; p++) {
            if (Ai[p] < 0 || Ai[p] >= m) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_sparselq: row index out of range");
            Tp[(size_t)Ai[p] + 1]++;
        }
        for (Long i = 0; i < m; i++) Tp[(size_t)i + 1] += Tp[(size_t)i];
        std::vector<Long> w(Tp.begin(), Tp.end() - 1);
        for (Long j = 0; j < n; j++)
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) { const Long q = w[(size_t)Ai[p]]++; Ti[(size_t)q] = j; Tx[(size_t)q] = Ax[p]; }
        (void)ordering;                                       // (the reference passes QR_ORDERING_DEFAULT whatever it is given, :729)
        return stmmqr_sparseqr(7, tol, n, m, Tp.data(), Ti.data(), Tx.data(), nullptr, relax, device, out);
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_sparselq: out of memory");
    }
}

}  // extern "C"
