// stmmqr_rfactor.cpp -- SURVEY.md 8 (f1): QR_qmult / QR_solve on the factors resident in HBM (host side; kernels: stmmqr_resident.hip).
#include "stmmqr_plan.h"

extern "C" {

// ---------------------------------------------------------------------------------------------
// QR_qmult (SparseQR.c:1790-2020, methods QR_QTX / QR_QX) and QR_solve (RETX_EQUALS_B, :2024-2216)
// on the factors that are still in HBM -- no download of the packed R+H.
// ---------------------------------------------------------------------------------------------
namespace {
// host half of qr_hpinv for the device: Wmap[S-row id] = position in the permuted row order (same rule as in
// stmmqr_plan_download); uploaded once per factorization together with the static maps
int ensure_scratch(stmmqr_plan &P);
int ensure_rowmap(stmmqr_plan &P)
{
    LCHK(ensure_scratch(P));
    if (P.rowmap_ready) return 0;
    hipStream_t st = P.stream;
    const long nf = P.nf, m = P.m, n = P.n;
    for (long f = 0; f < nf; f++)
        if (P.group[f] < 0) return fail(STMMQR_ERR_INVALID, "Q-apply / solve need every front on this device");
    std::vector<int> hii32((size_t)std::max(1L, P.hisize));
    if (P.hisize > 0)
        HIPCHK(hipMemcpyAsync(hii32.data(), P.d_Hii.p, (size_t)P.hisize * sizeof(int), hipMemcpyDeviceToHost, st));
    if (P.h_fnum.size() != (size_t)nf) P.h_fnum.resize((size_t)nf);
    if (nf > 0)
        HIPCHK(hipMemcpyAsync(P.h_fnum.data(), P.d_fnum.p, (size_t)nf * sizeof(FrontNum), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int> W((size_t)std::max(1L, m), 0);
    long row1 = 0, row2 = m;
    for (long i = P.Sleft[n]; i < m; i++) W[i] = (int)--row2;
    for (long f = 0; f < nf; f++) {
        const int *Hi = hii32.data() + P.Hip[f];
        const FrontNum &nm = P.h_fnum[f];
        const long rm = nm.rank, fm = nm.fm;
        for (long i = 0; i < rm; i++) W[Hi[i]] = (int)row1++;
        const long cn = P.fs[f].fn - P.fs[f].fp;
        const long cm = std::min(fm - rm, cn);
        for (long i = fm - 1; i >= rm + cm; i--) W[Hi[i]] = (int)--row2;
    }
    LCHK(P.d_Wmap.upload(W, st));
    {
        std::vector<int> rb((size_t)std::max(1L, nf), 0);
        long run = 0;
        for (long f = 0; f < nf; f++) { rb[f] = (int)run; run += P.h_fnum[f].rank; }
        LCHK(P.d_rowbase.upload(rb, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (!P.d_Rj.p) {
        std::vector<int> t((size_t)std::max(1L, P.rjsize));
        for (long i = 0; i < P.rjsize; i++) t[i] = (int)P.Rj[i];
        LCHK(P.d_Rj.upload(t, st));
        HIPCHK(hipStreamSynchronize(st));
        t.assign((size_t)std::max(1L, m), 0);
        for (long i = 0; i < m; i++) t[i] = (int)P.PLinv[i];
        LCHK(P.d_PLinv.upload(t, st));
        HIPCHK(hipStreamSynchronize(st));
        if (P.has_qfill) {
            t.assign((size_t)std::max(1L, n), 0);
            for (long j = 0; j < n; j++) t[j] = (int)P.Qfill[j];
            LCHK(P.d_Qfill.upload(t, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        LCHK(P.d_W.alloc((size_t)std::max(1L, m)));
        LCHK(P.d_Xs.alloc((size_t)std::max(1L, n)));
        LCHK(P.d_Io.alloc((size_t)std::max(1L, std::max(m, n))));
        LCHK(P.d_err.alloc(1));
        // dynamic LDS per level: k_qapply holds fm doubles + fn ints, k_rsolve fp + (fn - fp) doubles
        const auto &LV = P.glevels[0];
        P.level_lds_qa.assign(LV.size(), 0);
        P.level_lds_qa_all.assign(LV.size(), 0);
        P.level_lds_rs.assign(LV.size(), 0);
        P.level_lds_rt.assign(LV.size(), 0);
        P.level_qbig.assign(LV.size(), stmmqr_plan::QbLevel());
        P.t4items.clear(); P.t4fronts.clear(); P.t4dqo.clear(); P.qbt4off.clear();
        P.t4_doubles = 0; P.dq4_ints = 0; P.t4_ok = false; P.t4_tried = false; P.t4_valid = false;
        std::vector<QbDesc> qb;
        long xf = 1, dq = 1, wq = 1;
        for (size_t l = 0; l < LV.size(); l++) {
            long xo = 0, dqo = 0, wo = 0;
            P.level_qbig[l].off = (int)qb.size();
            P.level_qbig[l].t4i_off = (int)P.t4items.size();
            for (int q = 0; q < LV[l].n_all; q++) {
                const int f = P.lists[LV[l].all_off + q];
                const FrontSym &s = P.fs[f];
                const int need = (int)(((s.fm_ub + 1) & ~1) * 8 + s.fn * 4 + 16);
                P.level_lds_rt[l] = std::max(P.level_lds_rt[l], (int)((((s.fn + 1) & ~1) + ((std::min(s.fp, std::max(s.fm_ub, 1)) + 2) & ~1)) * 8 + s.fp * 4 + 32));
                P.level_lds_qa_all[l] = std::max(P.level_lds_qa_all[l], need);
                if (s.qbig) {
                    QbDesc d;
                    d.f = f; d.xoff = (int)xo; d.dqoff = (int)dqo; d.wqoff = (int)wo; d.nslab = (s.fm_ub + STM_QB_ROWS - 1) / STM_QB_ROWS; d.np_live = s.npanels;
                    qb.push_back(d);
                    {
                        const int ngr = (s.npanels + 3) / 4;
                        P.qbt4off.push_back(P.t4_doubles);
                        P.t4fronts.push_back(f);
                        P.t4dqo.push_back(P.dq4_ints);
                        for (int g = 0; g < ngr; g++) {
                            Qt4ItemHost it;
                            it.f = f; it.g = g; it.off = P.t4_doubles + (long long)g * stm_qt4_doubles(); it.dqo = P.dq4_ints;
                            P.t4items.push_back(it);
                        }
                        P.t4_doubles += (long long)ngr * stm_qt4_doubles();
                        P.dq4_ints += s.fn;
                    }
                    xo += s.fm_ub; dqo += s.fn; wo += 2L * d.nslab * STM_NB;
                    auto &Q = P.level_qbig[l];
                    Q.n++; Q.max_np = std::max(Q.max_np, s.npanels); Q.max_nslab = std::max(Q.max_nslab, d.nslab);
                    Q.max_fm = std::max(Q.max_fm, s.fm_ub);
                    Q.max_rsteps = std::max(Q.max_rsteps, (std::min(s.fp, s.fm_ub) + 31) / 32);
                } else {
                    P.level_lds_qa[l] = std::max(P.level_lds_qa[l], need);
                    P.level_lds_rs[l] = std::max(P.level_lds_rs[l], (int)((((s.fp + 1) & ~1) + (s.fn - s.fp) + 2) * 8 + s.fp * 4 + 16));
                }
            }
            P.level_qbig[l].t4i_n = (int)P.t4items.size() - P.level_qbig[l].t4i_off;
            xf = std::max(xf, xo); dq = std::max(dq, dqo); wq = std::max(wq, wo);
        }
        LCHK(P.d_Xf.alloc((size_t)xf));
        LCHK(P.d_Dq.alloc((size_t)dq));
        LCHK(P.d_Wq.alloc((size_t)wq));
        P.wq4_doubles = 4 * wq;
        P.xf_doubles = xf; P.wq_doubles = wq; P.rhs_cap = 1;
        P.d_U.release(); P.d_Xr.release();
        if (qb.empty()) qb.push_back(QbDesc());
        LCHK(P.d_qb.alloc(qb.size()));
        LCHK(P.d_Rm.alloc(qb.size()));
        HIPCHK(hipMemcpy(P.d_qb.p, qb.data(), qb.size() * sizeof(QbDesc), hipMemcpyHostToDevice));
        P.h_qb = qb;
    }
    {
        // Of THIS factorization: the panels of a split front that can hold a live reflector and the blocks of live pivot columns.  A
        // front of fm rows has fm live reflectors at most; they sit in its first fm + (dead pivot columns) columns, and a front has at
        // most fp - rank dead pivot columns.  The panels behind that column were launches that did nothing: on the default workload
        // 330 launches per Q'b, 177 of them for panels without a reflector (same finding as the factorization's own schedule,
        // stmmqr_host.cpp "how many panels").  STMMQR_LIVE_PANELS=0: every panel.
        const bool live = !(getenv("STMMQR_LIVE_PANELS") && atoi(getenv("STMMQR_LIVE_PANELS")) == 0);
        bool changed = false;
        for (size_t l = 0; l < P.level_qbig.size(); l++) {
            auto &Q = P.level_qbig[l];
            Q.live_np = 0; Q.live_rsteps = 0;
            for (int q = 0; q < Q.n; q++) {
                QbDesc &d = P.h_qb[(size_t)(Q.off + q)];
                const FrontSym &s = P.fs[d.f];
                const FrontNum &nm = P.h_fnum[d.f];
                int npl = s.npanels;
                if (live) {
                    const long lastcol = std::min((long)s.fn, (long)nm.fm + std::max(0L, (long)s.fp - (long)nm.rank));
                    npl = (nm.fm <= 0) ? 0 : (int)std::min((long)s.npanels, (lastcol - 1) / STM_NB + 1);
                }
                if (d.np_live != npl) { d.np_live = npl; changed = true; }
                Q.live_np = std::max(Q.live_np, npl);
                Q.live_rsteps = std::max(Q.live_rsteps, live ? (int)((nm.rank + 31) / 32) : Q.max_rsteps);
            }
        }
        if (changed && !P.h_qb.empty()) HIPCHK(hipMemcpy(P.d_qb.p, P.h_qb.data(), P.h_qb.size() * sizeof(QbDesc), hipMemcpyHostToDevice));
    }
    for (int b : P.level_lds_qa)
        if (b > 131072) return fail(STMMQR_ERR_TOO_LARGE, "a front has more rows than the Q-apply kernel holds in LDS");
    P.rowmap_ready = true;
    return 0;
}

int check_device_err(stmmqr_plan &P, const char *what)
{
    int e = 0;
    HIPCHK(hipMemcpyAsync(&e, P.d_err.p, sizeof(int), hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    if (e) return fail(STMMQR_ERR_INVALID, what);
    return 0;
}

// Slab recycling: the resident-factor kernels read fronts in front form.  A front whose slab was recycled is put back into
// that form, level by level, in a scratch that holds the widest tree level (res_ctx: the FrontSym array whose offsets point into
// the scratch; level_to_front_form: zeros + the inverse of k_rh_copy for the level's fronts).  Kept fronts are read where they are
// (their offset is taken relative to the scratch's base: one flat device address space).
int ensure_scratch(stmmqr_plan &P)
{
    if (!P.recycle || (P.d_scr.p && P.d_fs_scr.p)) return 0;
    // Two layouts.  Where HBM has room (all recycled slabs within a quarter of what is free; STMMQR_RESIDENT_CACHE=0 / 1 forces) the
    // scratch holds EVERY front in front form, rebuilt once per factorization at the first Q-apply / solve and kept for the
    // following ones: the memory comes back only while the factors are being used, never during the factorization.  Otherwise it
    // holds the widest tree level and every level is rebuilt whenever a kernel walks it.
    size_t freeb = 0, totalb = 0;
    HIPCHK(hipMemGetInfo(&freeb, &totalb));
    long long all = 0;
    for (long f = 0; f < P.nf; f++) if (!P.kept[(size_t)f]) all += (long long)P.fs[f].ld * P.fs[f].fn;
    const char *ev = getenv("STMMQR_RESIDENT_CACHE");
    P.scr_all = ev ? atoi(ev) != 0 : (8.0 * (double)all <= 0.25 * (double)freeb);
    if (P.scr_all) {
        long long o = 0;
        for (long f = 0; f < P.nf; f++)
            if (!P.kept[(size_t)f]) { P.fs_scr[(size_t)f].foff = o; o += (long long)P.fs[f].ld * P.fs[f].fn; }
    }
    P.scr_valid = false;
    P.t4_valid = false;
    LCHK(P.d_scr.alloc((size_t)std::max(1LL, P.scr_all ? all : P.scr_doubles)));
    std::vector<FrontSym> t = P.fs_scr;
    for (long f = 0; f < P.nf; f++)
        if (P.kept[(size_t)f]) t[(size_t)f].foff = (long long)((P.d_F.p + P.fs[f].foff) - P.d_scr.p);
    LCHK(P.d_fs_scr.upload(t, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    return 0;
}
DevCtx res_ctx(stmmqr_plan &P)
{
    DevCtx c = P.ctx();
    if (P.recycle) { c.fs = P.d_fs_scr.p; c.Farena = P.d_scr.p; }
    return c;
}
int level_to_front_form(stmmqr_plan &P, size_t l)
{
    if (!P.recycle) return 0;
    const auto &LV = P.glevels[0];
    const DevCtx c = P.ctx();
    if (P.scr_all) {
        if (P.scr_valid) return 0;
        // every front at once, kept until the next factorization (marked valid only once the launch was accepted)
        const int e = stm_launch_rh_unpack(c, P.d_fs_scr.p, P.d_lists.p + P.own_off, P.n_own, 64, P.d_kept.p, P.d_RH.p, P.d_scr.p, P.stream);
        P.scr_valid = (e == 0);
        return e;
    }
    if (LV[l].n_all <= 0) return 0;
    return stm_launch_rh_unpack(c, P.d_fs_scr.p, P.d_lists.p + LV[l].all_off, LV[l].n_all, 64, P.d_kept.p, P.d_RH.p, P.d_scr.p, P.stream);
}

// the per-vector buffers of the resident-factor operations for a batch of nb right-hand sides (grown on demand, never shrunk)
int ensure_rhs_batch(stmmqr_plan &P, int nb)
{
    if (nb <= P.rhs_cap) return 0;
    const size_t k = (size_t)nb;
    LCHK(P.d_W.alloc(k * (size_t)std::max(1L, P.m)));
    LCHK(P.d_Xs.alloc(k * (size_t)std::max(1L, P.n)));
    LCHK(P.d_Xf.alloc(k * (size_t)P.xf_doubles));
    LCHK(P.d_Wq.alloc(k * (size_t)P.wq_doubles));
    if (P.d_Wq4.p) LCHK(P.d_Wq4.alloc(k * (size_t)std::max(1LL, P.wq4_doubles)));
    if (P.d_U.p) { LCHK(P.d_U.alloc(k * (size_t)std::max(1L, P.rjsize))); LCHK(P.d_Xr.alloc(k * (size_t)std::max(1L, P.m))); }
    P.rhs_cap = nb;
    return 0;
}
RhsBatch rhs_strides(const stmmqr_plan &P)
{
    RhsBatch B;
    B.w = P.m; B.x = P.n; B.xf = P.xf_doubles; B.wq = P.wq_doubles; B.wq4 = std::max(1LL, P.wq4_doubles); B.u = std::max(1L, P.rjsize);
    return B;
}
// the largest batch the operations take in one pass (STMMQR_RHS_BATCH, default 32; 1: one vector after the other, as until round 4)
int rhs_batch_max()
{
    static int v = -1;
    if (v < 0) v = getenv("STMMQR_RHS_BATCH") ? std::max(1, atoi(getenv("STMMQR_RHS_BATCH"))) : 32;
    return v;
}

// W (device, S-row order; nb vectors at stride m) <- Q' W or Q W
int run_qapply(stmmqr_plan &P, int method, int nb = 1)
{
    const RhsBatch B = rhs_strides(P);
    DevCtx c = res_ctx(P);
    const int *L0 = P.d_lists.p;
    const auto &LV = P.glevels[0];
    // blocked form with the kept T factors; STMMQR_DBG bit 13 selects the reflector-by-reflector kernel (same result up
    // to rounding: used by the tests to cross-check the two)
    const bool blocked = c.Tall && !(c.dbg & 8192);
    // grouped split Q-apply: its buffers at the first use (STMMQR_QT4=0: the per-panel launches)
    const bool want_t4 = !(getenv("STMMQR_QT4") && atoi(getenv("STMMQR_QT4")) == 0);       // (read at every call: tests compare both)
    if (blocked && want_t4 && !P.t4_tried && !P.t4items.empty()) {
        P.t4_tried = true;
        size_t freeb = 0, totalb = 0;
        if (hipMemGetInfo(&freeb, &totalb) == hipSuccess &&
            8.0 * ((double)P.t4_doubles + (double)P.wq4_doubles) + 4.0 * (double)P.dq4_ints < 0.25 * (double)freeb) {
            LCHK(P.d_T4.alloc((size_t)P.t4_doubles));
            LCHK(P.d_Wq4.alloc((size_t)P.rhs_cap * (size_t)std::max(1LL, P.wq4_doubles)));
            LCHK(P.d_Dq4.alloc((size_t)std::max(1LL, P.dq4_ints)));
            LCHK(P.d_t4items.upload(P.t4items, P.stream));
            LCHK(P.d_t4fronts.upload(P.t4fronts, P.stream));
            LCHK(P.d_t4dqo.upload(P.t4dqo, P.stream));
            LCHK(P.d_qbt4off.upload(P.qbt4off, P.stream));
            P.t4_ok = true;
            P.t4_valid = false;
        }
    }
    const bool use_t4 = blocked && want_t4 && P.t4_ok;
    if (use_t4 && getenv("STMMQR_MEMDUMP") && !P.t4_valid)
        fprintf(stderr, "[stmmqr_hip] grouped Q-apply: T4 of %zu groups of %zu split fronts, %.3f GB (+ %.3f GB of slab partials)\n", P.t4items.size(),
                P.t4fronts.size(), 8e-9 * (double)P.t4_doubles, 8e-9 * (double)P.wq4_doubles);
    auto launch = [&](size_t l, int m) -> int {
        LCHK(level_to_front_form(P, l));
        if (blocked) {
            LCHK(stm_launch_qapply_t(c, L0 + LV[l].all_off, LV[l].n_all, m, P.d_W.p, P.level_lds_qa[l], P.stream, nb, B));
            // the large fronts of the level (independent of the others): rows split over workgroups, a launch per group of four panels
            // (k_qbig_step4, T4 built at the first use after a factorization) or per panel
            const auto &Q = P.level_qbig[l];
            if (Q.n > 0 && use_t4) {
                if (!P.t4_valid) { P.t4_level_valid.assign(LV.size(), 0); P.t4_valid = true; }
                if (!P.t4_level_valid[l]) {                          // (the level's fronts are in front form now: level_to_front_form)
                    LCHK(stm_launch_qt4_build(c, P.d_t4fronts.p + Q.off, P.d_t4dqo.p + Q.off, Q.n, P.d_t4items.p + Q.t4i_off, Q.t4i_n, P.d_Dq4.p,
                                              P.d_T4.p, P.stream));
                    P.t4_level_valid[l] = 1;
                }
                LCHK(stm_launch_qapply_big4(c, P.d_qb.p + Q.off, P.d_qbt4off.p + Q.off, Q.n, Q.live_np, Q.max_nslab, Q.max_fm, m, P.d_W.p, P.d_Xf.p,
                                            P.d_Dq.p, P.d_Wq4.p, P.d_T4.p, P.stream, nb, B));
                return 0;
            }
            LCHK(stm_launch_qapply_big(c, P.d_qb.p + Q.off, Q.n, Q.live_np, Q.max_nslab, Q.max_fm, m, P.d_W.p, P.d_Xf.p, P.d_Dq.p,
                                       P.d_Wq.p, P.stream, nb, B));
            return 0;
        }
        if (P.level_lds_qa_all[l] > 131072) return fail(STMMQR_ERR_TOO_LARGE, "a front has more rows than the unblocked Q-apply kernel holds in LDS");
        for (int j = 0; j < nb; j++)                                  // (the reflector-by-reflector cross-check kernel: one vector per launch)
            LCHK(stm_launch_qapply(c, L0 + LV[l].all_off, LV[l].n_all, m, P.d_W.p + (size_t)j * (size_t)P.m, P.level_lds_qa_all[l], P.d_err.p, P.stream));
        return 0;
    };
    if (method == 0) {
        for (size_t l = 0; l < LV.size(); l++) LCHK(launch(l, 0));
    } else {
        for (size_t l = LV.size(); l-- > 0;) LCHK(launch(l, 1));
    }
    return 0;
}
}  // namespace

namespace {
// host matrix (rows x cols, leading dimension ld) <-> contiguous device matrix (rows x cols), one transfer each way
int upload_cols(stmmqr_plan &P, DevBuf<double> &d, const double *H, long ld, long rows, long cols)
{
    if ((size_t)(rows * cols) > d.n) LCHK(d.alloc((size_t)std::max(1L, rows * cols)));
    if (rows > 0 && cols > 0)
        HIPCHK(hipMemcpy2DAsync(d.p, (size_t)rows * sizeof(double), H, (size_t)ld * sizeof(double), (size_t)rows * sizeof(double),
                                (size_t)cols, hipMemcpyHostToDevice, P.stream));
    return 0;
}
int download_cols(stmmqr_plan &P, const DevBuf<double> &d, double *H, long ld, long rows, long cols)
{
    if (rows > 0 && cols > 0)
        HIPCHK(hipMemcpy2DAsync(H, (size_t)ld * sizeof(double), d.p, (size_t)rows * sizeof(double), (size_t)rows * sizeof(double),
                                (size_t)cols, hipMemcpyDeviceToHost, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    return 0;
}
// nb vectors (stride m in `in` / `out`, device, the reference's row order) through Q' (method 0) or Q (method 1) in ONE pass over the tree
int qapply_vectors(stmmqr_plan &P, int method, const double *in, double *out, int nb)
{
    hipStream_t st = P.stream;
    const int m = (int)P.m;
    LCHK(ensure_rhs_batch(P, nb));
    if (method == 0) {
        LCHK(stm_launch_perm(in, P.d_PLinv.p, P.d_W.p, m, 1, st, nb, m, m));            // W[PLinv[i]] = x[i]
        LCHK(run_qapply(P, 0, nb));
        LCHK(stm_launch_perm(P.d_W.p, P.d_Wmap.p, out, m, 1, st, nb, m, m));            // out[Wmap[r]] = W[r]
    } else {
        LCHK(stm_launch_perm(in, P.d_Wmap.p, P.d_W.p, m, 0, st, nb, m, m));             // W[r] = x[Wmap[r]]
        LCHK(run_qapply(P, 1, nb));
        LCHK(stm_launch_perm(P.d_W.p, P.d_PLinv.p, out, m, 0, st, nb, m, m));           // out[i] = W[PLinv[i]]
    }
    return 0;
}
// back substitution R x = y on the device work vectors W (internal row order; nb of them at stride m) -> d_Xs (R's column order, stride n)
int rsolve_vector(stmmqr_plan &P, int nb = 1)
{
    const RhsBatch B = rhs_strides(P);
    DevCtx c = res_ctx(P);
    const int *L0 = P.d_lists.p;
    const auto &LV = P.glevels[0];
    hipStream_t st = P.stream;
    for (size_t l = LV.size(); l-- > 0;) {
        LCHK(level_to_front_form(P, l));
        LCHK(stm_launch_rsolve(c, L0 + LV[l].all_off, LV[l].n_all, P.d_Rj.p, P.d_W.p, P.d_Xs.p, P.level_lds_rs[l], P.d_err.p, st, nb, B));
        const auto &Q = P.level_qbig[l];         // the large fronts of the level: rows split over workgroups
        LCHK(stm_launch_rsolve_big(c, P.d_qb.p + Q.off, Q.n, Q.live_rsteps, Q.max_nslab, P.d_Rj.p, P.d_W.p, P.d_Xs.p, P.d_Xf.p,
                                   P.d_Dq.p, P.d_Rm.p + Q.off, P.d_err.p, st, nb, B));
    }
    return 0;
}
}  // namespace

// QR_qmult (STMMQR/include/SparseQR.h:403-409, SparseQR.c:1815-2116) on the resident factors, in place:
//   method 0 QR_QTX: X (m x k, ldx >= m) <- Q' X      method 1 QR_QX: X <- Q X
//   method 2 QR_XQT: X (k x m, ldx >= k) <- X Q'      method 3 QR_XQ: X <- X Q
// Row (methods 0, 1) / column (2, 3) order as in the reference: Q'X and X Q come out in the permuted order of the
// factorization (HPinv), Q X and X Q' take it.  All vectors cross PCIe in ONE transfer each way.
int stmmqr_plan_qmult(stmmqr_plan *plan, int method, double *X, stm_long ldx, stm_long k)
{
    if (!plan || !plan->factored) return fail(STMMQR_ERR_INVALID, "no factorization held by the plan");
    if (!X || k < 0 || method < 0 || method > 3 || ldx < ((method <= 1) ? plan->m : k))
        return fail(STMMQR_ERR_INVALID, "bad qmult arguments");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    LCHK(ensure_rowmap(P));
    const long m = P.m;
    if (k == 0 || m == 0) return 0;
    if (method <= 1) {
        LCHK(upload_cols(P, P.d_Xall, X, ldx, m, k));
        // (batches of right-hand sides: every launch of the pass over the tree carries all of them, RhsBatch)
        for (stm_long j = 0; j < k; j += rhs_batch_max()) {
            const int nb = (int)std::min<stm_long>(rhs_batch_max(), k - j);
            LCHK(qapply_vectors(P, method, P.d_Xall.p + j * m, P.d_Xall.p + j * m, nb));
        }
        return download_cols(P, P.d_Xall, X, ldx, m, k);
    }
    // X Q' = (Q X')' and X Q = (Q' X')': the rows of X are the vectors (SparseQR.c:2040-2075: the same permutation pattern)
    std::vector<double> T((size_t)m * (size_t)k);
    for (stm_long r = 0; r < k; r++)
        for (long i = 0; i < m; i++) T[(size_t)r * m + i] = X[r + (size_t)i * ldx];
    LCHK(upload_cols(P, P.d_Xall, T.data(), m, m, k));
    const int vm = (method == 2) ? 1 : 0;
    for (stm_long r = 0; r < k; r += rhs_batch_max()) {
        const int nb = (int)std::min<stm_long>(rhs_batch_max(), k - r);
        LCHK(qapply_vectors(P, vm, P.d_Xall.p + r * m, P.d_Xall.p + r * m, nb));
    }
    LCHK(download_cols(P, P.d_Xall, T.data(), m, m, k));
    for (stm_long r = 0; r < k; r++)
        for (long i = 0; i < m; i++) X[r + (size_t)i * ldx] = T[(size_t)r * m + i];
    return 0;
}

// QR_solve (STMMQR/include/SparseQR.h:411-417, SparseQR.c:2118-2216) on the resident factors:
//   system 0 QR_RX_EQUALS_B   : X (n x nrhs) = R \ B            B (m x nrhs) in R's row order (what QR_QTX returns)
//   system 1 QR_RETX_EQUALS_B : X = E (R \ B)
//   system 2 QR_RTX_EQUALS_B  : X (m x nrhs) = R' \ B           B (n x nrhs), rows of X beyond the rank are zero
//   system 3 QR_RTX_EQUALS_ETB: X = R' \ (E' B)
// Dead pivot columns: x = 0 (systems 0, 1: the basic solution of qr_rsolve) / no equation (2, 3: the squeezed R).
int stmmqr_plan_rsolve(stmmqr_plan *plan, int system, const double *B, stm_long ldb, double *X, stm_long ldx, stm_long nrhs)
{
    if (!plan || !plan->factored) return fail(STMMQR_ERR_INVALID, "no factorization held by the plan");
    if (system < 0 || system > 3 || !B || !X || nrhs < 0) return fail(STMMQR_ERR_INVALID, "bad solve arguments");
    stmmqr_plan &P = *plan;
    const long m = P.m, n = P.n;
    const long brows = (system <= 1) ? m : n, xrows = (system <= 1) ? n : m;
    if (ldb < brows || ldx < xrows) return fail(STMMQR_ERR_INVALID, "bad leading dimension");
    HIPCHK(hipSetDevice(P.device));
    LCHK(ensure_rowmap(P));
    hipStream_t st = P.stream;
    if (nrhs == 0) return 0;
    HIPCHK(hipMemsetAsync(P.d_err.p, 0, sizeof(int), st));
    LCHK(upload_cols(P, P.d_Xall, B, ldb, brows, nrhs));
    if ((size_t)(xrows * nrhs) > P.d_Yall.n) LCHK(P.d_Yall.alloc((size_t)std::max(1L, xrows * nrhs)));
    const int nbmax = rhs_batch_max();
    const RhsBatch RB = rhs_strides(P);
    if (system <= 1) {
        for (stm_long j = 0; j < nrhs; j += nbmax) {
            const int nb = (int)std::min<stm_long>(nbmax, nrhs - j);
            LCHK(ensure_rhs_batch(P, nb));
            LCHK(stm_launch_perm(P.d_Xall.p + j * m, P.d_Wmap.p, P.d_W.p, (int)m, 0, st, nb, m, m));          // W[r] = b[Wmap[r]]
            LCHK(rsolve_vector(P, nb));
            LCHK(stm_launch_perm(P.d_Xs.p, (system == 1 && P.has_qfill) ? P.d_Qfill.p : nullptr, P.d_Yall.p + j * n, (int)n, 1, st, nb, n, n));
        }
    } else {
        DevCtx c = res_ctx(P);
        const int *L0 = P.d_lists.p;
        const auto &LV = P.glevels[0];
        if (!P.d_U.p) {
            LCHK(P.d_U.alloc((size_t)P.rhs_cap * (size_t)std::max(1L, P.rjsize)));
            LCHK(P.d_Xr.alloc((size_t)P.rhs_cap * (size_t)std::max(1L, m)));
        }
        for (int need : P.level_lds_rt)
            if (need > 131072) return fail(STMMQR_ERR_TOO_LARGE, "a front is too wide for the one-workgroup R' solve");
        for (stm_long j = 0; j < nrhs; j += nbmax) {
            const int nb = (int)std::min<stm_long>(nbmax, nrhs - j);
            LCHK(ensure_rhs_batch(P, nb));
            // b in R's column order: E'B gathers through Qfill
            LCHK(stm_launch_perm(P.d_Xall.p + j * n, (system == 3 && P.has_qfill) ? P.d_Qfill.p : nullptr, P.d_Xs.p, (int)n, 0, st, nb, n, n));
            HIPCHK(hipMemsetAsync(P.d_Xr.p, 0, (size_t)nb * (size_t)std::max(1L, m) * sizeof(double), st));
            for (size_t l = 0; l < LV.size(); l++) {
                LCHK(level_to_front_form(P, l));
                LCHK(stm_launch_rtsolve(c, L0 + LV[l].all_off, LV[l].n_all, P.d_Xs.p, P.d_U.p, P.d_Xr.p, P.d_rowbase.p,
                                        P.level_lds_rt[l], st, nb, RB));
            }
            HIPCHK(hipMemcpyAsync(P.d_Yall.p + j * m, P.d_Xr.p, (size_t)nb * (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
    }
    LCHK(download_cols(P, P.d_Yall, X, ldx, xrows, nrhs));
    return check_device_err(P, "internal: live pivot count of a front differs from its rank");
}

// X (n x nrhs, ldx >= n) = E * R^{-1} * (Q' B)(1:n)  for B (m x nrhs, ldb >= m): QR_qmult(QR_QTX) followed by
// QR_solve(QR_RETX_EQUALS_B), the driver's least-squares solve (qrtest.c:11-53), without the trip to the host in between;
// dead columns get x = 0 (basic solution).
int stmmqr_plan_solve(stmmqr_plan *plan, const double *B, stm_long ldb, double *X, stm_long ldx, stm_long nrhs)
{
    if (!plan || !plan->factored) return fail(STMMQR_ERR_INVALID, "no factorization held by the plan");
    if (!B || !X || ldb < plan->m || ldx < plan->n || nrhs < 0) return fail(STMMQR_ERR_INVALID, "bad solve arguments");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    LCHK(ensure_rowmap(P));
    hipStream_t st = P.stream;
    const long m = P.m, n = P.n;
    if (nrhs == 0) return 0;
    HIPCHK(hipMemsetAsync(P.d_err.p, 0, sizeof(int), st));
    LCHK(upload_cols(P, P.d_Xall, B, ldb, m, nrhs));
    if ((size_t)(n * nrhs) > P.d_Yall.n) LCHK(P.d_Yall.alloc((size_t)std::max(1L, n * nrhs)));
    for (stm_long j = 0; j < nrhs; j += rhs_batch_max()) {
        const int nb = (int)std::min<stm_long>(rhs_batch_max(), nrhs - j);
        LCHK(ensure_rhs_batch(P, nb));
        LCHK(stm_launch_perm(P.d_Xall.p + j * m, P.d_PLinv.p, P.d_W.p, (int)m, 1, st, nb, m, m));
        LCHK(run_qapply(P, 0, nb));
        LCHK(rsolve_vector(P, nb));
        LCHK(stm_launch_perm(P.d_Xs.p, P.has_qfill ? P.d_Qfill.p : nullptr, P.d_Yall.p + j * n, (int)n, 1, st, nb, n, n));   // X[Qfill[j]] = x[j]
    }
    LCHK(download_cols(P, P.d_Yall, X, ldx, n, nrhs));
    return check_device_err(P, "internal: live pivot count of a front differs from its rank");
}

}  // extern "C"
