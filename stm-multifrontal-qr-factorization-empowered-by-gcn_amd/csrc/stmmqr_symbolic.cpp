// stmmqr_symbolic.cpp -- own symbolic phase: everything qr_factorize consumes, from A's pattern.
//
// Host-only integer code (no HIP): the symbolic analysis is graph work on the pattern and runs once per pattern; its
// output is the qr_symbolic object the numeric phase (stmmqr_host.cpp) plans from.  Reference counterparts, paths
// relative to /root/reference/STMMQR:
//   stmmqr_analyze          qr_analyze                       src/qr/SparseQR_analyze.c:20-700 (serial return :692-699)
//     column_etree          SparseChol_etree (A'A case)      src/chol/SparseChol_analyze.c:1118-1260
//     tree_postorder        SparseChol_postorder             :1339-1480  (child order: by weight, ties by index)
//     column_counts         SparseChol_rowcolcounts          :1602-1960  (Gilbert / Ng / Peyton skeleton counts of S S')
//     relaxed_supernodes    SparseChol_super_symbolic2       src/chol/SparseChol_super_symbolic.c:91-690
//     s_row_form            qr_stranspose1                   src/qr/SparseQR_analyze.c:1172-1287
//   stmmqr_relax_for_qr     Relaxfactor_setting(RELAX_FOR_QR) src/core/SparseCore_common.c:1172-1203 on the defaults :147-152
// as SparseChol_analyze_p2 (src/chol/SparseChol_analyze.c:278-760) strings them together for the QR case: ordering GIVEN
// (or FIXED = natural), no post-ordering of the permutation (qr_analyze sets cc->postorder = FALSE, :128-141), always
// supernodal.  Elimination tree, column counts and the row structure of every supernode are uniquely defined by the
// pattern and the permutation; the amalgamation heuristic and the weighted post-order are restated rule by rule so that
// the result is the reference's bit for bit (tests/test_symbolic.py: every sym_* array of every committed fixture).
//
// The task / stack decomposition of the reference's parallel analysis (:701-1161) has no counterpart here by design:
// tree parallelism is the step scheduler's (stmmqr_host.cpp); the object says ntasks = ns = 1 like the
// reference's own serial analysis (SPQR_grain <= 1).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/stmmqr_hip.h"
#include "stmmqr_internal.h"

namespace {

typedef stm_long Long;
const Long NONE = -1;

// F = A(:,Q) in column form with sorted columns (Fp, Fi: m x n) and its transpose S (Sp, Si: n x m, sorted)
struct PermutedPattern {
    std::vector<Long> Fp, Fi, Sp, Si;
};

bool permute_pattern(Long m, Long n, const Long *Ap, const Long *Ai, const Long *Q, PermutedPattern &P)
{
    const Long anz = Ap[n];
    P.Sp.assign((size_t)m + 1, 0);
    for (Long k = 0; k < n; k++) {
        const Long j = Q ? Q[k] : k;
        if (j < 0 || j >= n) return false;
        for (Long p = Ap[j]; p < Ap[j + 1]; p++) {
            const Long i = Ai[p];
            if (i < 0 || i >= m) return false;
            P.Sp[(size_t)i + 1]++;
        }
    }
    for (Long i = 0; i < m; i++) P.Sp[(size_t)i + 1] += P.Sp[(size_t)i];
    P.Si.assign((size_t)std::max<Long>(anz, 1), 0);
    {
        std::vector<Long> w(P.Sp.begin(), P.Sp.end() - 1);
        for (Long k = 0; k < n; k++) {                       // ascending k: every column of S comes out sorted
            const Long j = Q ? Q[k] : k;
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) P.Si[(size_t)w[(size_t)Ai[p]]++] = k;
        }
    }
    P.Fp.assign((size_t)n + 1, 0);
    for (Long i = 0; i < m; i++)
        for (Long p = P.Sp[(size_t)i]; p < P.Sp[(size_t)i + 1]; p++) P.Fp[(size_t)P.Si[(size_t)p] + 1]++;
    for (Long k = 0; k < n; k++) P.Fp[(size_t)k + 1] += P.Fp[(size_t)k];
    P.Fi.assign((size_t)std::max<Long>(anz, 1), 0);
    {
        std::vector<Long> w(P.Fp.begin(), P.Fp.end() - 1);
        for (Long i = 0; i < m; i++)                          // ascending i: sorted columns of F
            for (Long p = P.Sp[(size_t)i]; p < P.Sp[(size_t)i + 1]; p++) P.Fi[(size_t)w[(size_t)P.Si[(size_t)p]]++] = i;
    }
    return true;
}

// Elimination tree of F'F without forming it: every row of F links the columns it touches into a path (Liu), the tree
// is grown with path compression on a virtual-ancestor array.
void column_etree(Long m, Long n, const PermutedPattern &P, std::vector<Long> &parent)
{
    parent.assign((size_t)n, NONE);
    std::vector<Long> anc((size_t)n, NONE), prev((size_t)m, NONE);
    for (Long j = 0; j < n; j++) {
        for (Long p = P.Fp[(size_t)j]; p < P.Fp[(size_t)j + 1]; p++) {
            const Long i = P.Fi[(size_t)p];
            Long k = prev[(size_t)i];
            prev[(size_t)i] = j;
            // edge (k, j), k < j: climb from k to the root of its current tree and hang it under j
            while (k != NONE && k != j) {
                const Long a = anc[(size_t)k];
                if (a == j) break;
                anc[(size_t)k] = j;
                if (a == NONE) { parent[(size_t)k] = j; break; }
                k = a;
            }
        }
    }
}

// Post-order of a forest.  The order of the children of a node decides the result: without weights they are visited in
// increasing index; with weights in increasing (clamped) weight, ties in increasing index -- the lists are built by
// pushing at the head, hence the reversed loops (SparseChol_postorder :1400-1455).
Long tree_postorder(const std::vector<Long> &parent, Long n, const Long *weight, std::vector<Long> &post)
{
    std::vector<Long> head((size_t)n + 1, NONE), next((size_t)std::max<Long>(n, 1), NONE), stack((size_t)std::max<Long>(n, 1));
    if (!weight) {
        for (Long j = n - 1; j >= 0; j--) {
            const Long p = parent[(size_t)j];
            if (p >= 0 && p < n) { next[(size_t)j] = head[(size_t)p]; head[(size_t)p] = j; }
        }
    } else {
        std::vector<Long> whead((size_t)std::max<Long>(n, 1), NONE);
        for (Long j = 0; j < n; j++) {
            const Long p = parent[(size_t)j];
            if (p >= 0 && p < n) {
                Long w = weight[j];
                w = std::max<Long>(0, w);
                w = std::min<Long>(w, n - 1);
                next[(size_t)j] = whead[(size_t)w];
                whead[(size_t)w] = j;
            }
        }
        for (Long w = n - 1; w >= 0; w--)
            for (Long j = whead[(size_t)w], nj; j != NONE; j = nj) {
                nj = next[(size_t)j];
                const Long p = parent[(size_t)j];
                next[(size_t)j] = head[(size_t)p];
                head[(size_t)p] = j;
            }
    }
    post.assign((size_t)std::max<Long>(n, 1), 0);
    Long k = 0;
    for (Long r = 0; r < n; r++) {
        if (parent[(size_t)r] != NONE) continue;
        Long top = 0;
        stack[0] = r;
        while (top >= 0) {
            const Long p = stack[(size_t)top], c = head[(size_t)p];
            if (c == NONE) { top--; post[(size_t)k++] = p; }
            else { head[(size_t)p] = next[(size_t)c]; stack[(size_t)++top] = c; }
        }
    }
    return k;
}

// Column counts of the Cholesky factor of S S' (S: n x m, a column = the permuted columns one row of A touches), from the
// elimination tree and a post-order of it, in almost linear time (Gilbert, Ng, Peyton 1994): a column j gets +1 for
// every row subtree in which it is a leaf, the overlap of consecutive leaves is taken off at their least common ancestor
// (found with a disjoint-set forest that follows the tree as it is traversed), and the counts are summed up the tree.
// Also returns fl = sum count^2 and lnz = sum count (SparseChol_rowcolcounts :1945-1960).
void column_counts(Long m, Long n, const PermutedPattern &P, const std::vector<Long> &parent, const std::vector<Long> &post,
                   std::vector<Long> &count, double &fl, double &lnz)
{
    count.assign((size_t)n, 0);
    std::vector<Long> first((size_t)n, NONE), ipost((size_t)n), setp((size_t)n), prevleaf((size_t)n, NONE), prevnbr((size_t)n, NONE);
    for (Long k = 0; k < n; k++) {
        const Long i = post[(size_t)k];
        ipost[(size_t)i] = k;
        count[(size_t)i] = (first[(size_t)i] == NONE) ? 1 : 0;               // a leaf of the tree starts at 1
        for (Long r = i; r != NONE && first[(size_t)r] == NONE; r = parent[(size_t)r]) first[(size_t)r] = k;
    }
    // rows of A (columns of S) bucketed by the smallest post-order index they touch
    std::vector<Long> rhead((size_t)n + 1, NONE), rnext((size_t)std::max<Long>(m, 1), NONE);
    for (Long i = 0; i < m; i++) {
        const Long p0 = P.Sp[(size_t)i], p1 = P.Sp[(size_t)i + 1];
        if (p1 <= p0) continue;
        Long k = ipost[(size_t)P.Si[(size_t)p0]];
        for (Long p = p0; p < p1; p++) k = std::min(k, ipost[(size_t)P.Si[(size_t)p]]);
        rnext[(size_t)i] = rhead[(size_t)k];
        rhead[(size_t)k] = i;
    }
    for (Long j = 0; j < n; j++) setp[(size_t)j] = j;
    for (Long k = 0; k < n; k++) {
        const Long j = post[(size_t)k];
        if (parent[(size_t)j] != NONE) count[(size_t)parent[(size_t)j]]--;     // j is not a leaf of its parent's row subtree twice
        prevnbr[(size_t)j] = k;
        for (Long r = rhead[(size_t)k]; r != NONE; r = rnext[(size_t)r]) {
            for (Long p = P.Sp[(size_t)r]; p < P.Sp[(size_t)r + 1]; p++) {
                const Long u = P.Si[(size_t)p];
                if (prevnbr[(size_t)u] >= k) continue;                          // edge (j, u) already seen at this step
                if (first[(size_t)j] > prevnbr[(size_t)u]) {
                    // j is a new leaf of the row subtree of u
                    count[(size_t)j]++;
                    const Long pl = prevleaf[(size_t)u];
                    if (pl != NONE) {
                        Long q = pl;
                        while (q != setp[(size_t)q]) q = setp[(size_t)q];
                        for (Long s = pl, sn; s != q; s = sn) { sn = setp[(size_t)s]; setp[(size_t)s] = q; }
                        count[(size_t)q]--;                                     // the paths from the two leaves meet at q
                    }
                    prevleaf[(size_t)u] = j;
                }
                prevnbr[(size_t)u] = k;
            }
        }
        if (parent[(size_t)j] != NONE) setp[(size_t)j] = parent[(size_t)j];
    }
    for (Long j = 0; j < n; j++)
        if (parent[(size_t)j] != NONE) count[(size_t)parent[(size_t)j]] += count[(size_t)j];
    fl = 0; lnz = 0;
    for (Long j = 0; j < n; j++) { const double c = (double)count[(size_t)j]; lnz += c; fl += c * c; }
}

// Supernodes of the factor of S S': fundamental supernodes (a chain of the tree whose counts drop by one, the upper node
// having a single child), then relaxed amalgamation of a supernode with its parent when the parent is the next supernode
// and the explicit zeros the merge adds stay within the nrelax / zrelax limits (SparseChol_super_symbolic.c:226-330),
// then the row structure of every supernode (Rj) by walking each row of S S' down the supernodal tree (:560-605).
struct Supernodes {
    Long nsuper = 0;
    std::vector<Long> Super, Rp, Rj;
};

bool relaxed_supernodes(Long m, Long n, const PermutedPattern &P, const std::vector<Long> &parent, const std::vector<Long> &count,
                        const stmmqr_relax &rx, Supernodes &S)
{
    std::vector<Long> nchild((size_t)std::max<Long>(n, 1), 0);
    for (Long j = 0; j < n; j++)
        if (parent[(size_t)j] != NONE) nchild[(size_t)parent[(size_t)j]]++;
    std::vector<Long> sup;                                     // first column of every fundamental supernode
    if (n > 0) sup.push_back(0);
    for (Long j = 1; j < n; j++)
        if (parent[(size_t)j - 1] != j || count[(size_t)j - 1] != count[(size_t)j] + 1 || nchild[(size_t)j] > 1) sup.push_back(j);
    const Long nfs = (Long)sup.size();
    sup.push_back(n);
    std::vector<Long> smap((size_t)std::max<Long>(n, 1)), sparent((size_t)std::max<Long>(nfs, 1), NONE);
    for (Long s = 0; s < nfs; s++)
        for (Long k = sup[(size_t)s]; k < sup[(size_t)s + 1]; k++) smap[(size_t)k] = s;
    for (Long s = 0; s < nfs; s++) {
        const Long pj = parent[(size_t)sup[(size_t)s + 1] - 1];
        sparent[(size_t)s] = (pj == NONE) ? NONE : smap[(size_t)pj];
    }
    std::vector<Long> merged((size_t)std::max<Long>(nfs, 1), NONE), nscol((size_t)std::max<Long>(nfs, 1)), zeros((size_t)std::max<Long>(nfs, 1), 0),
        snz((size_t)std::max<Long>(nfs, 1));
    for (Long s = 0; s < nfs; s++) { nscol[(size_t)s] = sup[(size_t)s + 1] - sup[(size_t)s]; snz[(size_t)s] = count[(size_t)sup[(size_t)s]]; }
    const double z0 = (rx.zrelax[0] != rx.zrelax[0]) ? 0 : rx.zrelax[0], z1 = (rx.zrelax[1] != rx.zrelax[1]) ? 0 : rx.zrelax[1],
                 z2 = (rx.zrelax[2] != rx.zrelax[2]) ? 0 : rx.zrelax[2];
    const double int_max = 9223372036854775807.0;
    for (Long s = nfs - 2; s >= 0; s--) {
        if (sparent[(size_t)s] == NONE) continue;
        // the supernode the parent has been merged into so far (with path compression)
        Long top = sparent[(size_t)s];
        while (merged[(size_t)top] != NONE) top = merged[(size_t)top];
        for (Long q = sparent[(size_t)s], qn; merged[(size_t)q] != NONE; q = qn) { qn = merged[(size_t)q]; merged[(size_t)q] = top; }
        if (top != s + 1) continue;                            // only a parent that is the very next supernode
        const Long n0 = nscol[(size_t)s], n1 = nscol[(size_t)s + 1], ns = n0 + n1;
        Long totzeros = zeros[(size_t)s + 1];
        const double lnz1 = (double)snz[(size_t)s + 1];
        bool merge;
        if (ns <= rx.nrelax[0]) merge = true;
        else {
            const double lnz0 = (double)snz[(size_t)s];
            const double xnew = (double)n0 * (lnz1 + (double)n0 - lnz0);
            const Long newzeros = n0 * (snz[(size_t)s + 1] + n0 - snz[(size_t)s]);
            if (xnew == 0) merge = true;
            else {
                const double xtot = (double)totzeros + xnew, xns = (double)ns;
                const double xsize = (xns * (xns + 1) / 2) + xns * (lnz1 - (double)n1);
                const double z = xtot / xsize;
                totzeros += newzeros;
                merge = ((ns <= rx.nrelax[1] && z < z0) || (ns <= rx.nrelax[2] && z < z1) || (z < z2)) && (xsize < int_max / sizeof(double));
            }
        }
        if (merge) {
            zeros[(size_t)s] = totzeros;
            merged[(size_t)s + 1] = s;
            snz[(size_t)s] = n0 + snz[(size_t)s + 1];
            nscol[(size_t)s] += nscol[(size_t)s + 1];
        }
    }
    S.Super.clear();
    std::vector<Long> rsz;
    for (Long s = 0; s < nfs; s++)
        if (merged[(size_t)s] == NONE) { S.Super.push_back(sup[(size_t)s]); rsz.push_back(snz[(size_t)s]); }
    S.nsuper = (Long)S.Super.size();
    S.Super.push_back(n);
    const Long nsup = S.nsuper;
    for (Long s = 0; s < nsup; s++)
        for (Long k = S.Super[(size_t)s]; k < S.Super[(size_t)s + 1]; k++) smap[(size_t)k] = s;
    std::vector<Long> rsparent((size_t)std::max<Long>(nsup, 1), NONE);
    for (Long s = 0; s < nsup; s++) {
        const Long pj = parent[(size_t)S.Super[(size_t)s + 1] - 1];
        rsparent[(size_t)s] = (pj == NONE) ? NONE : smap[(size_t)pj];
    }
    S.Rp.assign((size_t)nsup + 1, 0);
    for (Long s = 0; s < nsup; s++) {
        S.Rp[(size_t)s + 1] = S.Rp[(size_t)s] + rsz[(size_t)s];
        if (S.Rp[(size_t)s + 1] < 0) return false;
    }
    S.Rj.assign((size_t)std::max<Long>(S.Rp[(size_t)nsup], 1), 0);
    // row structure: for every column k (in order) the supernodes whose structure contains row k are those on the paths,
    // in the supernodal tree, from the supernodes of the entries i < k1 of (S S')(:, k) up to (excluding) k's own
    std::vector<Long> fill(S.Rp.begin(), S.Rp.end() - 1), flag((size_t)std::max<Long>(nsup, 1), NONE);
    for (Long s = 0; s < nsup; s++) {
        const Long k1 = S.Super[(size_t)s], k2 = S.Super[(size_t)s + 1];
        for (Long k = k1; k < k2; k++) S.Rj[(size_t)fill[(size_t)s]++] = k;
        for (Long k = k1; k < k2; k++) {
            flag[(size_t)s] = k;
            for (Long p = P.Fp[(size_t)k]; p < P.Fp[(size_t)k + 1]; p++) {
                const Long r = P.Fi[(size_t)p];                                // a row of A with an entry in column k
                for (Long q = P.Sp[(size_t)r]; q < P.Sp[(size_t)r + 1]; q++) {
                    const Long i = P.Si[(size_t)q];
                    if (i >= k1) break;                                          // (sorted)
                    for (Long si = smap[(size_t)i]; flag[(size_t)si] != k; si = rsparent[(size_t)si]) {
                        if (fill[(size_t)si] >= S.Rp[(size_t)si + 1]) return false;   // (cannot happen: counts are exact)
                        S.Rj[(size_t)fill[(size_t)si]++] = k;
                        flag[(size_t)si] = k;
                    }
                }
            }
        }
    }
    for (Long s = 0; s < nsup; s++)
        if (fill[(size_t)s] != S.Rp[(size_t)s + 1]) return false;
    (void)m;
    return true;
}

}  // namespace

struct stmmqr_analysis {
    stm_qr_symbolic sym;
    std::vector<Long> Sp, Sj, Qfill, PLinv, Sleft, Parent, Child, Childp, Super, Rp, Rj, Post, Hip, Fm, Cm;
    double info[8];
};

extern "C" {

void stmmqr_relax_for_qr(stm_long n, stm_long nnz, stmmqr_relax *r)
{
    if (!r) return;
    // SparseCore_start's defaults, then Relaxfactor_setting(n, nnz, RELAX_FOR_QR)
    r->nrelax[0] = 4; r->nrelax[1] = 16; r->nrelax[2] = 48;
    r->zrelax[0] = 0.8; r->zrelax[1] = 0.1; r->zrelax[2] = 0.05;
    const size_t nn = (size_t)n * (size_t)n;
    const double dense = (double)nnz / (double)nn;
    r->nrelax[0] = 4;
    if (dense > 0.0005) { r->nrelax[1] = 32; r->nrelax[2] = 64; }
    r->zrelax[0] = 0.85; r->zrelax[1] = 0.1; r->zrelax[2] = 0.05;
    if (dense > 0.001) r->zrelax[2] += 0.03;
}

static int analyze_impl(stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const stm_long *Quser, int do_rank,
                        const stmmqr_relax *relax, stmmqr_analysis **out)
{
    if (!out) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: null output");
    *out = nullptr;
    if (m < 0 || n < 0 || !Ap || (!Ai && Ap[n] > 0)) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: bad matrix");
    if (Ap[0] != 0) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: Ap[0] must be 0");
    for (Long j = 0; j < n; j++)
        if (Ap[j + 1] < Ap[j]) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: column pointers must not decrease");
    if (Quser) {
        std::vector<char> seen((size_t)std::max<Long>(n, 1), 0);
        for (Long k = 0; k < n; k++) {
            const Long j = Quser[k];
            if (j < 0 || j >= n || seen[(size_t)j]) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: Quser is not a permutation");
            seen[(size_t)j] = 1;
        }
    }
    stmmqr_relax rx;
    if (relax) rx = *relax;
    else { rx.nrelax[0] = 4; rx.nrelax[1] = 16; rx.nrelax[2] = 48; rx.zrelax[0] = 0.8; rx.zrelax[1] = 0.1; rx.zrelax[2] = 0.05; }
    const Long anz = Ap[n];

    stmmqr_analysis *R = new stmmqr_analysis();
    std::unique_ptr<stmmqr_analysis> guard(R);
    memset(&R->sym, 0, sizeof R->sym);
    memset(R->info, 0, sizeof R->info);

    // ---- the Cholesky-style analysis of A(:,Q)' A(:,Q) (SparseChol_analyze_p2 for SPQR) ----
    PermutedPattern P;
    if (!permute_pattern(m, n, Ap, Ai, Quser, P)) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: index out of range");
    std::vector<Long> eparent, epost, count;
    column_etree(m, n, P, eparent);
    if (tree_postorder(eparent, n, nullptr, epost) != n) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: invalid elimination tree");
    double fl = 0, lnz = 0;
    column_counts(m, n, P, eparent, epost, count, fl, lnz);
    Supernodes SN;
    if (!relaxed_supernodes(m, n, P, eparent, count, rx, SN)) return stm_fail(STMMQR_ERR_TOO_LARGE, "stmmqr_analyze: problem too large");
    const Long nf = SN.nsuper;
    R->Super = SN.Super; R->Rp = SN.Rp; R->Rj = SN.Rj;
    R->Super.resize((size_t)nf + 1); R->Rp.resize((size_t)nf + 1);
    // (SparseCore_allocate_factor: Perm is always an array, the identity for the natural ordering)
    R->Qfill.resize((size_t)std::max<Long>(n, 1));
    for (Long k = 0; k < n; k++) R->Qfill[(size_t)k] = Quser ? Quser[k] : k;
    const std::vector<Long> &Super = R->Super, &Rp = R->Rp, &Rj = R->Rj;

    // ---- frontal tree (qr_analyze :284-372) ----
    std::vector<Long> &Parent = R->Parent, &Child = R->Child, &Childp = R->Childp, &Post = R->Post;
    Parent.assign((size_t)nf + 1, NONE); Childp.assign((size_t)nf + 2, 0); Child.assign((size_t)nf + 1, 0);
    {
        std::vector<Long> front_of((size_t)std::max<Long>(n, 1), 0);
        for (Long f = 0; f < nf; f++)
            for (Long j = Super[(size_t)f]; j < Super[(size_t)f + 1]; j++) front_of[(size_t)j] = f;
        for (Long f = 0; f < nf; f++) {
            const Long fp = Super[(size_t)f + 1] - Super[(size_t)f], p = Rp[(size_t)f] + fp;
            const Long par = (p < Rp[(size_t)f + 1]) ? front_of[(size_t)Rj[(size_t)p]] : nf;   // front of the first non-pivotal column
            Parent[(size_t)f] = par;
            Childp[(size_t)par]++;
        }
        Parent[(size_t)nf] = NONE;
        std::vector<Long> weight((size_t)nf + 1);
        for (Long f = 0; f < nf; f++) weight[(size_t)f] = Rp[(size_t)f + 1] - Rp[(size_t)f];
        weight[(size_t)nf] = 1;
        if (tree_postorder(Parent, nf + 1, weight.data(), Post) != nf + 1) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: invalid frontal tree");
        Long tot = 0;
        for (Long k = 0; k <= nf; k++) { const Long c = Childp[(size_t)k]; Childp[(size_t)k] = tot; tot += c; }
        Childp[(size_t)nf + 1] = tot;
        std::vector<Long> w(Childp.begin(), Childp.end());
        for (Long kf = 0; kf < nf; kf++) { const Long c = Post[(size_t)kf]; Child[(size_t)w[(size_t)Parent[(size_t)c]]++] = c; }
    }

    // ---- S = A(P,Q) in row form, rows sorted by leftmost column (qr_stranspose1 :1172-1287) ----
    std::vector<Long> &Sp = R->Sp, &Sj = R->Sj, &PLinv = R->PLinv, &Sleft = R->Sleft;
    Sp.assign((size_t)m + 1, 0); Sj.assign((size_t)std::max<Long>(anz, 1), 0); PLinv.assign((size_t)std::max<Long>(m, 1), NONE); Sleft.assign((size_t)n + 2, 0);
    {
        std::vector<Long> w((size_t)std::max<Long>(m, 1), 0);
        Long k = 0;
        for (Long col = 0; col < n; col++) {
            const Long j = R->Qfill[(size_t)col], kstart = k;
            for (Long p = Ap[j]; p < Ap[j + 1]; p++) {
                const Long i = Ai[p];
                Long row = PLinv[(size_t)i];
                if (row == NONE) { row = k++; PLinv[(size_t)i] = row; w[(size_t)row] = 1; }
                else w[(size_t)row]++;
            }
            Sleft[(size_t)col] = k - kstart;
        }
        Long s = 0;
        for (Long col = 0; col < n; col++) { const Long t = s; s += Sleft[(size_t)col]; Sleft[(size_t)col] = t; }
        Sleft[(size_t)n] = k;
        Sleft[(size_t)n + 1] = m;
        if (k < m)
            for (Long i = 0; i < m; i++)
                if (PLinv[(size_t)i] == NONE) { const Long row = k++; PLinv[(size_t)i] = row; w[(size_t)row] = 0; }
        Long p = 0;
        for (Long row = 0; row < m; row++) { const Long t = p; p += w[(size_t)row]; w[(size_t)row] = t; Sp[(size_t)row] = t; }
        Sp[(size_t)m] = p;
        for (Long col = 0; col < n; col++) {
            const Long j = R->Qfill[(size_t)col];
            for (Long q = Ap[j]; q < Ap[j + 1]; q++) Sj[(size_t)w[(size_t)PLinv[(size_t)Ai[q]]]++] = col;
        }
    }

    // ---- front sizes, staircases, flop and memory bounds (:378-640) ----
    std::vector<Long> &Fm = R->Fm, &Cm = R->Cm, &Hip = R->Hip;
    Fm.assign((size_t)nf + 1, 0); Cm.assign((size_t)nf + 1, 0); Hip.assign((size_t)nf + 1, 0);
    Long maxfn = 0, stack = 0, maxstack = 0, rxsize = 0, rhxsize = 0;
    double total_flops = 0;
    bool ok = true;
    const Long long_max = 9223372036854775807L;
    auto add_ok = [&](Long a, Long b) -> Long { if (a > long_max - b) { ok = false; return long_max; } return a + b; };
    {
        std::vector<Long> Fmap((size_t)std::max<Long>(n, 1), 0), Stair;
        for (Long kf = 0; ok && kf < nf; kf++) {
            const Long f = Post[(size_t)kf];
            const Long col1 = Super[(size_t)f], fp = Super[(size_t)f + 1] - col1, p1 = Rp[(size_t)f], fn = Rp[(size_t)f + 1] - p1;
            maxfn = std::max(maxfn, fn);
            for (Long j = 0; j < fn; j++) Fmap[(size_t)Rj[(size_t)(p1 + j)]] = j;
            Stair.assign((size_t)std::max<Long>(fn, 1), 0);
            for (Long j = 0; j < fp; j++) Stair[(size_t)j] = Sleft[(size_t)(col1 + j) + 1] - Sleft[(size_t)(col1 + j)];
            Long ctot = 0;
            for (Long q = Childp[(size_t)f]; q < Childp[(size_t)f + 1]; q++) {
                const Long c = Child[(size_t)q], pc = Rp[(size_t)c], fnc = Rp[(size_t)c + 1] - pc, fpc = Super[(size_t)c + 1] - Super[(size_t)c];
                const Long cn = fnc - fpc, fmc = Fm[(size_t)c];
                Long cm;
                if (do_rank) cm = std::min(fmc, cn);
                else { const Long rc = std::min(fmc, fpc); cm = std::min(std::max<Long>(fmc - rc, 0), cn); }
                for (Long ci = 0; ci < cm; ci++) Stair[(size_t)Fmap[(size_t)Rj[(size_t)(pc + fpc + ci)]]]++;
                ctot += cm * (cm + 1) / 2 + cm * (cn - cm);
            }
            Long fm = 0;
            for (Long j = 0; j < fn; j++) { fm += Stair[(size_t)j]; Stair[(size_t)j] = fm; }
            if (fn > 0 && fm > long_max / std::max<Long>(fn, 1)) { ok = false; break; }
            const Long fsize = fm * fn;
            Fm[(size_t)f] = fm;
            const Long rm = std::min(fm, fp), rn = fn;
            rxsize += rm * (rm + 1) / 2 + rm * (rn - rm);
            const Long cn = fn - fp;
            const Long cm_max = std::min(fm, cn), cm_min = std::min(std::max<Long>(fm - rm, 0), cn);
            const Long csize_max = cm_max * (cm_max + 1) / 2 + cm_max * (cn - cm_max), csize_min = cm_min * (cm_min + 1) / 2 + cm_min * (cn - cm_min);
            const Long csize = do_rank ? csize_max : csize_min;
            Cm[(size_t)f] = do_rank ? cm_max : cm_min;
            double fflops = 0;
            Long rhsize = 0;
            for (Long j = 0; j < fn; j++) {
                Long t = std::max(j + 1, Stair[(size_t)j]);
                t = std::min(t, fm);
                rhsize += t;
                if (t > j) { const double h = (double)(t - j); fflops += 3 * h + 4 * h * (double)(fn - j - 1); }
            }
            rhsize -= csize_min;
            rhxsize += rhsize;
            total_flops += fflops;
            stack = add_ok(stack, fsize);
            maxstack = std::max(maxstack, stack);
            stack -= ctot;
            stack = add_ok(stack, csize);
            maxstack = std::max(maxstack, stack);
            stack -= fsize;
            stack += rhsize;
        }
    }
    Long hisize = 0;
    for (Long f = 0; f < nf; f++) { Hip[(size_t)f] = hisize; hisize = add_ok(hisize, Fm[(size_t)f]); }
    Hip[(size_t)nf] = hisize;
    if (!ok) return stm_fail(STMMQR_ERR_TOO_LARGE, "stmmqr_analyze: problem too large");

    stm_qr_symbolic &Q = R->sym;
    Q.m = m; Q.n = n; Q.anz = anz;
    Q.Sp = Sp.data(); Q.Sj = Sj.data(); Q.Qfill = R->Qfill.data(); Q.PLinv = PLinv.data(); Q.Sleft = Sleft.data();
    Q.nf = nf; Q.maxfn = maxfn;
    Q.Parent = Parent.data(); Q.Child = Child.data(); Q.Childp = Childp.data(); Q.Super = R->Super.data(); Q.Rp = R->Rp.data();
    Q.Rj = R->Rj.data(); Q.Post = Post.data();
    Q.rjsize = std::max<Long>(Rp[(size_t)nf], 1);                     // (L->ssize = MAX (1, ssize))
    Q.do_rank_detection = do_rank ? 1 : 0; Q.maxstack = maxstack; Q.hisize = hisize; Q.keepH = 1;
    Q.Hip = Hip.data();
    Q.ntasks = 1; Q.ns = 1;
    Q.TaskChildp = Q.TaskChild = Q.TaskStack = Q.TaskFront = Q.TaskFrontp = Q.On_stack = Q.Stack_maxstack = nullptr;
    Q.Fm = Fm.data(); Q.Cm = Cm.data();
    R->info[0] = total_flops;                                         // cc->SPQR_flopcount_bound
    R->info[1] = fl; R->info[2] = lnz;                                // Common->fl, Common->lnz of the Cholesky analysis
    R->info[3] = (lnz > 0 && fl / lnz >= 1000) ? 1 : 0;               // QR_CHUNK_FLAG (SparseChol_analyze.c:727-730)
    R->info[4] = (double)rxsize;                                      // SPQR_istat[0]: bound on nnz(R)
    R->info[5] = (double)(rhxsize - rxsize);                          // SPQR_istat[1]: bound on nnz(H)
    R->info[6] = (double)maxstack;
    R->info[7] = (double)nf;
    *out = guard.release();
    return 0;
}

int stmmqr_analyze(stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const stm_long *Quser, int do_rank_detection,
                   const stmmqr_relax *relax, stmmqr_analysis **out)
{
    try {
        return analyze_impl(m, n, Ap, Ai, Quser, do_rank_detection, relax, out);
    } catch (const std::bad_alloc &) {
        return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "stmmqr_analyze: out of memory");
    } catch (...) {
        return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analyze: internal error");
    }
}

const stm_qr_symbolic *stmmqr_analysis_symbolic(const stmmqr_analysis *a) { return a ? &a->sym : nullptr; }

int stmmqr_analysis_info(const stmmqr_analysis *a, double *info)
{
    if (!a || !info) return stm_fail(STMMQR_ERR_INVALID, "stmmqr_analysis_info: null argument");
    memcpy(info, a->info, sizeof a->info);
    return 0;
}

void stmmqr_analysis_free(stmmqr_analysis *a) { delete a; }

}  // extern "C"
