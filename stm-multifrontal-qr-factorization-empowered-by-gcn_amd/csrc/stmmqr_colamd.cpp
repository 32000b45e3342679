// stmmqr_colamd.cpp -- column approximate minimum degree ordering (stage 2), host-only integer code.
//
// The driver's default fill-reducing ordering (qrtest.c:155-169 -> SparseQR.c:930-961 -> SparseChol_colamd,
// src/chol/SparseChol_analyze.c:924-1080 -> colamd, src/base/colamd.c:738-902): the COLAMD algorithm of Davis, Gilbert,
// Larimore and Ng (ACM TOMS 30, 2004), i.e. a symbolic LU of A with row merging in which the next pivot column is the one
// of least approximate external degree, with mass elimination, aggressive row absorption and supercolumn detection.
//
// A fill-reducing ordering is a heuristic: ANY permutation gives a valid factorization, but a drop-in symbolic phase has
// to reproduce the reference's fronts, so the rules that decide ties are restated here exactly as the reference applies
// them (file:line = src/base/colamd.c):
//   * dense / empty columns go last in natural order, dense columns are those with more than
//     max(16, 10 sqrt(min(n_row, n_col))) entries; only completely dense rows are removed (knobs of SparseChol_colamd:
//     prune_dense = 10, prune_dense2 = -1, aggressive = TRUE, SparseCore_common.c:184-188)                    (:1161-1290)
//   * initial score of a column = sum over its rows of (row degree - 1), capped at n_col                       (:1293-1337)
//   * degree lists are LIFO: columns are pushed at the head, initially in decreasing index, so that among equal scores the
//     lowest index / the most recently rescored column is taken first                                           (:1356-1388, 1862-1877)
//   * pivot row pattern = the live columns of the pivot column's rows in the order they are met                (:1516-1550)
//   * set differences |row \ pivot row| by tagged marks; a row whose difference is empty is absorbed           (:1600-1640)
//   * approximate degree of a column = sum of the set differences of its rows (capped) + |pivot row| - thickness, capped
//     at n_col - k - thickness                                                                                  (:1650-1690, 1843-1860)
//   * supercolumns: columns of the pivot row with equal hash (sum of row indices mod n_col + 1), equal length, equal score
//     and identical row lists are merged into the FIRST of them in bucket order, buckets are LIFO              (:2023-2170)
//   * non-principal columns are numbered right after their principal column, in index order                    (:1922-2020)
// Storage is a pool of index lists that is compacted when it runs full; compaction keeps the order inside every list, so
// -- as in the reference -- when it happens has no influence on the ordering.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/stmmqr_hip.h"
#include "stmmqr_internal.h"

namespace {

typedef stm_long Long;
const Long NONE = -1;

struct Colamd {
    Long n_row = 0, n_col = 0;
    std::vector<Long> pool;                  // column lists (row indices), then row lists (column indices), then free space
    Long pfree = 0;
    // columns.  start < 0: dead (-1 principal: ordered, `order` valid; -2 absorbed into `parent`)
    std::vector<Long> cstart, clen, thick, parent, score, order, prev, next, hash, hnext, headhash;
    // rows.  mark < 0: dead
    std::vector<Long> rstart, rlen, rdeg, mark, rsave;
    std::vector<Long> head;                  // per score: first column of the degree list; (< -1: -(c + 2) = first column of the
                                             // hash bucket while the degree list of that value is empty)
    bool col_alive(Long c) const { return cstart[(size_t)c] >= 0; }
    bool row_alive(Long r) const { return mark[(size_t)r] >= 0; }

    // put every live list back to back at the start of the pool, dropping dead members, order preserved
    void compact()
    {
        Long dst = 0;
        for (Long c = 0; c < n_col; c++) {
            if (!col_alive(c)) continue;
            const Long src = cstart[(size_t)c], len = clen[(size_t)c];
            cstart[(size_t)c] = dst;
            for (Long q = 0; q < len; q++) {
                const Long r = pool[(size_t)(src + q)];
                if (row_alive(r)) pool[(size_t)dst++] = r;
            }
            clen[(size_t)c] = dst - cstart[(size_t)c];
        }
        // rows are not stored in index order: tag the first slot of every live row, then sweep the pool once
        for (Long r = 0; r < n_row; r++) {
            if (!row_alive(r) || rlen[(size_t)r] == 0) { mark[(size_t)r] = -1; continue; }
            rsave[(size_t)r] = pool[(size_t)rstart[(size_t)r]];
            pool[(size_t)rstart[(size_t)r]] = -r - 1;
        }
        Long src = dst;
        while (src < pfree) {
            if (pool[(size_t)src] >= 0) { src++; continue; }
            const Long r = -pool[(size_t)src] - 1;
            pool[(size_t)src] = rsave[(size_t)r];
            const Long len = rlen[(size_t)r];
            rstart[(size_t)r] = dst;
            for (Long q = 0; q < len; q++) {
                const Long c = pool[(size_t)src++];
                if (col_alive(c)) pool[(size_t)dst++] = c;
            }
            rlen[(size_t)r] = dst - rstart[(size_t)r];
        }
        pfree = dst;
    }

    Long reset_marks(Long tag, Long max_mark)
    {
        if (tag <= 0 || tag >= max_mark) {
            for (Long r = 0; r < n_row; r++)
                if (row_alive(r)) mark[(size_t)r] = 0;
            tag = 1;
        }
        return tag;
    }

    // A: n_row x n_col in column form (Ap, Ai; sorted, no duplicates).  perm[k] = the k-th column of the ordering.
    bool run(Long nr, Long nc, const Long *Ap, const Long *Ai, std::vector<Long> &perm)
    {
        n_row = nr; n_col = nc;
        const Long nnz = Ap[nc];
        const size_t cap = (size_t)(2 * nnz + nc + nnz / 5 + 64);
        pool.assign(cap, 0);
        auto col_vec = [&](std::vector<Long> &v, Long val) { v.assign((size_t)nc + 1, val); };
        col_vec(cstart, 0); col_vec(clen, 0); col_vec(thick, 1); col_vec(parent, NONE); col_vec(score, 0); col_vec(order, NONE);
        col_vec(prev, NONE); col_vec(next, NONE); col_vec(hash, 0); col_vec(hnext, NONE); col_vec(headhash, NONE);
        rstart.assign((size_t)nr + 1, 0); rlen.assign((size_t)nr + 1, 0); rdeg.assign((size_t)nr + 1, 0); mark.assign((size_t)nr + 1, 0);
        rsave.assign((size_t)nr + 1, 0);
        head.assign((size_t)nc + 1, NONE);

        // ---- column form, then the row form right behind it ----
        for (Long c = 0; c < nc; c++) {
            cstart[(size_t)c] = Ap[c];
            clen[(size_t)c] = Ap[c + 1] - Ap[c];
            Long last = -1;
            for (Long p = Ap[c]; p < Ap[c + 1]; p++) {
                const Long r = Ai[p];
                if (r < 0 || r >= nr || r <= last) return false;       // (sorted, duplicate-free input is this caller's contract)
                last = r;
                pool[(size_t)p] = r;
                rlen[(size_t)r]++;
            }
        }
        {
            Long at = nnz;
            for (Long r = 0; r < nr; r++) { rstart[(size_t)r] = at; at += rlen[(size_t)r]; }
            std::vector<Long> fill(rstart.begin(), rstart.end());
            for (Long c = 0; c < nc; c++)
                for (Long p = Ap[c]; p < Ap[c + 1]; p++) pool[(size_t)fill[(size_t)Ai[p]]++] = c;
            pfree = 2 * nnz;
        }
        for (Long r = 0; r < nr; r++) { rdeg[(size_t)r] = rlen[(size_t)r]; mark[(size_t)r] = 0; }

        // ---- dense / empty members out, initial scores, degree lists ----
        const Long dense_row = nc - 1;                                                   // (knob < 0: completely dense rows only)
        const Long dense_col = (Long)std::max(16.0, 10.0 * std::sqrt((double)std::min(nr, nc)));
        Long n_col2 = nc, max_deg = 0;
        for (Long c = nc - 1; c >= 0; c--)
            if (clen[(size_t)c] == 0) { order[(size_t)c] = --n_col2; cstart[(size_t)c] = -1; }
        for (Long c = nc - 1; c >= 0; c--) {
            if (!col_alive(c)) continue;
            if (clen[(size_t)c] > dense_col) {
                order[(size_t)c] = --n_col2;
                for (Long q = 0; q < clen[(size_t)c]; q++) rdeg[(size_t)pool[(size_t)(cstart[(size_t)c] + q)]]--;
                cstart[(size_t)c] = -1;
            }
        }
        for (Long r = 0; r < nr; r++) {
            const Long d = rdeg[(size_t)r];
            if (d > dense_row || d == 0) mark[(size_t)r] = -1;
            else max_deg = std::max(max_deg, d);
        }
        for (Long c = nc - 1; c >= 0; c--) {
            if (!col_alive(c)) continue;
            Long sc = 0, w = cstart[(size_t)c];
            for (Long q = 0; q < clen[(size_t)c]; q++) {
                const Long r = pool[(size_t)(cstart[(size_t)c] + q)];
                if (!row_alive(r)) continue;
                pool[(size_t)w++] = r;
                sc += rdeg[(size_t)r] - 1;
                sc = std::min(sc, nc);
            }
            const Long len = w - cstart[(size_t)c];
            if (len == 0) { order[(size_t)c] = --n_col2; cstart[(size_t)c] = -1; }
            else { clen[(size_t)c] = len; score[(size_t)c] = sc; }
        }
        for (Long c = nc - 1; c >= 0; c--) {
            if (!col_alive(c)) continue;
            const Long sc = score[(size_t)c], nx = head[(size_t)sc];
            prev[(size_t)c] = NONE; next[(size_t)c] = nx;
            if (nx != NONE) prev[(size_t)nx] = c;
            head[(size_t)sc] = c;
        }

        // ---- the elimination ----
        const Long max_mark = 2147483647L - nc;                                         // (INT_MAX - n_col, :1466)
        Long tag = reset_marks(0, max_mark), min_score = 0;
        for (Long k = 0; k < n_col2;) {
            while (min_score < nc && head[(size_t)min_score] == NONE) min_score++;
            const Long piv = head[(size_t)min_score];
            {
                const Long nx = next[(size_t)piv];
                head[(size_t)min_score] = nx;
                if (nx != NONE) prev[(size_t)nx] = NONE;
            }
            const Long piv_score = score[(size_t)piv], piv_thick = thick[(size_t)piv];
            order[(size_t)piv] = k;
            k += piv_thick;
            const Long need = std::min(piv_score, nc - k);
            if (pfree + need >= (Long)cap) { compact(); tag = reset_marks(0, max_mark); }
            // pivot row = union of the rows of the pivot column (live columns only, each once: thickness negated as a flag)
            const Long prow_start = pfree;
            Long prow_deg = 0;
            thick[(size_t)piv] = -piv_thick;
            for (Long q = 0; q < clen[(size_t)piv]; q++) {
                const Long r = pool[(size_t)(cstart[(size_t)piv] + q)];
                if (!row_alive(r)) continue;
                for (Long e = 0; e < rlen[(size_t)r]; e++) {
                    const Long c = pool[(size_t)(rstart[(size_t)r] + e)];
                    const Long t = thick[(size_t)c];
                    if (t > 0 && col_alive(c)) { thick[(size_t)c] = -t; pool[(size_t)pfree++] = c; prow_deg += t; }
                }
            }
            thick[(size_t)piv] = piv_thick;
            max_deg = std::max(max_deg, prow_deg);
            for (Long q = 0; q < clen[(size_t)piv]; q++) mark[(size_t)pool[(size_t)(cstart[(size_t)piv] + q)]] = -1;
            const Long prow_len = pfree - prow_start;
            const Long prow = prow_len > 0 ? pool[(size_t)cstart[(size_t)piv]] : NONE;     // the row index the merged row keeps
            // set differences of the rows met through the pivot row's columns
            for (Long e = 0; e < prow_len; e++) {
                const Long c = pool[(size_t)(prow_start + e)];
                const Long t = -thick[(size_t)c];
                thick[(size_t)c] = t;
                {   // out of its degree list
                    const Long pv = prev[(size_t)c], nx = next[(size_t)c];
                    if (pv == NONE) head[(size_t)score[(size_t)c]] = nx; else next[(size_t)pv] = nx;
                    if (nx != NONE) prev[(size_t)nx] = pv;
                }
                for (Long q = 0; q < clen[(size_t)c]; q++) {
                    const Long r = pool[(size_t)(cstart[(size_t)c] + q)];
                    const Long mk = mark[(size_t)r];
                    if (mk < 0) continue;
                    Long diff = mk - tag;
                    if (diff < 0) diff = rdeg[(size_t)r];
                    diff -= t;
                    if (diff == 0) mark[(size_t)r] = -1;                    // aggressive absorption
                    else mark[(size_t)r] = diff + tag;
                }
            }
            // approximate degrees, hash buckets
            for (Long e = 0; e < prow_len; e++) {
                const Long c = pool[(size_t)(prow_start + e)];
                unsigned long h = 0;
                Long sc = 0, w = cstart[(size_t)c];
                for (Long q = 0; q < clen[(size_t)c]; q++) {
                    const Long r = pool[(size_t)(cstart[(size_t)c] + q)];
                    const Long mk = mark[(size_t)r];
                    if (mk < 0) continue;
                    pool[(size_t)w++] = r;
                    h += (unsigned long)r;
                    sc += mk - tag;
                    sc = std::min(sc, nc);
                }
                clen[(size_t)c] = w - cstart[(size_t)c];
                if (clen[(size_t)c] == 0) {
                    // nothing left of it but the pivot row: ordered right now (mass elimination)
                    cstart[(size_t)c] = -1;
                    prow_deg -= thick[(size_t)c];
                    order[(size_t)c] = k;
                    k += thick[(size_t)c];
                } else {
                    score[(size_t)c] = sc;
                    h %= (unsigned long)(nc + 1);
                    const Long hd = head[(size_t)h];
                    Long first;
                    if (hd > NONE) { first = headhash[(size_t)hd]; headhash[(size_t)hd] = c; }
                    else { first = -(hd + 2); head[(size_t)h] = -(c + 2); }
                    hnext[(size_t)c] = first;
                    hash[(size_t)c] = (Long)h;
                }
            }
            // supercolumns among the columns of the pivot row
            for (Long e = 0; e < prow_len; e++) {
                const Long c0 = pool[(size_t)(prow_start + e)];
                if (!col_alive(c0)) continue;
                const Long h = hash[(size_t)c0], hd = head[(size_t)h];
                const Long first = (hd > NONE) ? headhash[(size_t)hd] : -(hd + 2);
                for (Long sup = first; sup != NONE; sup = hnext[(size_t)sup]) {
                    const Long len = clen[(size_t)sup];
                    Long before = sup;
                    for (Long c = hnext[(size_t)sup]; c != NONE; c = hnext[(size_t)c]) {
                        bool same = (clen[(size_t)c] == len && score[(size_t)c] == score[(size_t)sup]);
                        for (Long q = 0; same && q < len; q++)
                            same = pool[(size_t)(cstart[(size_t)sup] + q)] == pool[(size_t)(cstart[(size_t)c] + q)];
                        if (!same) { before = c; continue; }
                        thick[(size_t)sup] += thick[(size_t)c];
                        parent[(size_t)c] = sup;
                        cstart[(size_t)c] = -2;
                        order[(size_t)c] = NONE;
                        hnext[(size_t)before] = hnext[(size_t)c];
                    }
                }
                if (hd > NONE) headhash[(size_t)hd] = NONE; else head[(size_t)h] = NONE;
            }
            cstart[(size_t)piv] = -1;
            tag = reset_marks(tag + max_deg + 1, max_mark);
            // final scores, back into the degree lists; the pivot row becomes a row of the matrix
            Long w = prow_start;
            for (Long e = 0; e < prow_len; e++) {
                const Long c = pool[(size_t)(prow_start + e)];
                if (!col_alive(c)) continue;
                pool[(size_t)w++] = c;
                pool[(size_t)(cstart[(size_t)c] + clen[(size_t)c]++)] = prow;
                Long sc = score[(size_t)c] + prow_deg;
                const Long mx = nc - k - thick[(size_t)c];
                sc -= thick[(size_t)c];
                sc = std::min(sc, mx);
                score[(size_t)c] = sc;
                const Long nx = head[(size_t)sc];
                next[(size_t)c] = nx; prev[(size_t)c] = NONE;
                if (nx != NONE) prev[(size_t)nx] = c;
                head[(size_t)sc] = c;
                min_score = std::min(min_score, sc);
            }
            if (prow_deg > 0) {
                rstart[(size_t)prow] = prow_start;
                rlen[(size_t)prow] = w - prow_start;
                rdeg[(size_t)prow] = prow_deg;
                mark[(size_t)prow] = 0;
            }
        }

        // ---- absorbed columns follow their principal column ----
        for (Long i = 0; i < nc; i++) {
            if (cstart[(size_t)i] == -1 || order[(size_t)i] != NONE) continue;
            Long top = i;
            do top = parent[(size_t)top]; while (cstart[(size_t)top] != -1);
            // (the reference numbers column i only and re-hangs it under its principal column -- its inner loop leaves after
            //  one pass because it follows the pointer it has just rewritten, :1990-2010 -- so the absorbed columns of one
            //  supercolumn come out in index order, each in its own turn of the outer loop)
            Long ord = order[(size_t)top];
            order[(size_t)i] = ord++;
            parent[(size_t)i] = top;
            order[(size_t)top] = ord;
        }
        perm.assign((size_t)std::max<Long>(nc, 1), 0);
        for (Long c = 0; c < nc; c++) perm[(size_t)order[(size_t)c]] = c;
        return true;
    }
};

}  // namespace

// perm[k] = k-th column of the ordering of the n_row x n_col pattern (Ap, Ai sorted).  false: invalid input.
bool stm_colamd_order(stm_long n_row, stm_long n_col, const stm_long *Ap, const stm_long *Ai, std::vector<stm_long> &perm)
{
    Colamd w;
    return w.run(n_row, n_col, Ap, Ai, perm);
}
