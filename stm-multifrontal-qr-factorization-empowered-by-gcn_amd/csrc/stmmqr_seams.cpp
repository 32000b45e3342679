// stmmqr_seams.cpp -- the reference's INNER seams (STMMQR/include/SparseQR.h:145-268) on host buffers.
//
// Each call copies its operands to the device, runs the SAME kernels the full factorization uses
// (stmmqr_panel.hip, stmmqr_update.hip, ...) on a one-front context and copies the results back: they exist so that every kernel
// can be parity-tested against the reference function it replaces (tests/test_gpu_seams.py).  Integer-only
// helpers of the reference (qr_fsize, qr_csize, qr_fcsize, qr_hpinv) are host code here exactly as they are
// host code there; their device counterparts live inside k_setup / dev_cpack / the download step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/stmmqr_hip.h"
#include "stmmqr_device.h"
#include "stmmqr_kernels.h"
#include "stmmqr_internal.h"

extern "C" int stmmqr_device_count(void);

namespace {

template <class T> struct Buf {
    T *p = nullptr;
    size_t n = 0;
    bool alloc(size_t cnt)
    {
        n = cnt;
        return hipMalloc((void **)&p, std::max<size_t>(1, cnt) * sizeof(T)) == hipSuccess;
    }
    bool up(const T *h, size_t cnt) { return alloc(cnt) && (cnt == 0 || hipMemcpy(p, h, cnt * sizeof(T), hipMemcpyHostToDevice) == hipSuccess); }
    bool up(const std::vector<T> &h) { return up(h.data(), h.size()); }
    bool down(T *h, size_t cnt) const { return cnt == 0 || hipMemcpy(h, p, cnt * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
    ~Buf() { if (p) (void)hipFree(p); }
};

bool device_ready()
{
    static int state = 0;
    if (state == 0) state = (stmmqr_device_count() > 0 && stm_configure_kernels() == 0 && stm_configure_capanel() == 0) ? 1 : -1;
    return state == 1;
}

// one dense front on the device: F (ld = even >= m), Stair, Tau, Rdead, one FrontSym/FrontNum, C, T
struct OneFront {
    FrontSym s{};
    FrontNum nm{};
    Buf<FrontSym> d_fs;
    Buf<FrontNum> d_nm;
    Buf<double> d_F, d_C, d_T, d_Gp, d_Tau, d_RH, d_sig;
    Buf<int> d_St, d_tslot, d_flist, d_parts;
    Buf<long long> d_Rhoff;
    Buf<long long> d_Rboff;
    Buf<unsigned long long> d_dbg;
    Buf<char> d_Rdead;
    DevCtx c{};
    long m = 0, n = 0;

    bool init(long m_, long n_, long npiv, const double *F, long ldf, const stm_long *Stair)
    {
        m = m_; n = n_;
        memset(&s, 0, sizeof s); memset(&nm, 0, sizeof nm);
        s.ld = (int)std::max(2L, (m + 1) & ~1L);
        s.fn = (int)n; s.fp = (int)std::min(n, std::max(0L, npiv)); s.fm_ub = (int)m; s.fm_est = (int)m;
        s.npanels = (int)((n + STM_NB - 1) / STM_NB);
        s.nsched = s.npanels;
        s.parent = -1;
        nm.fm = (int)m; nm.rank = (int)std::min(m, (long)s.fp);
        std::vector<double> Fd((size_t)s.ld * std::max(1L, n), 0.0);
        if (F)
            for (long j = 0; j < n; j++) memcpy(&Fd[(size_t)j * s.ld], F + j * ldf, sizeof(double) * (size_t)m);
        std::vector<int> st32((size_t)std::max(1L, n), 0);
        if (Stair) for (long j = 0; j < n; j++) st32[j] = (int)Stair[j];
        const long cn = n - s.fp, cm = std::min(m, cn);
        std::vector<int> zero1(1, 0);
        bool ok = d_F.up(Fd) && d_St.up(st32) && d_fs.up(&s, 1) && d_nm.up(&nm, 1) &&
                  d_C.alloc((size_t)std::max(1L, cm * (cm + 1) / 2 + cm * (cn - cm))) && d_T.alloc((size_t)STM_PD_RING * STM_NB * STM_NB) && d_Gp.alloc((size_t)(stm_ca_slabs(s) + 1) * STM_NB * STM_NB) &&
                  d_Tau.alloc((size_t)std::max(1L, n)) && d_Rdead.alloc((size_t)std::max(1L, n)) && d_tslot.up(zero1) &&
                  d_flist.up(zero1) && d_Rhoff.alloc((size_t)std::max(1L, n)) && d_Rboff.alloc(1) &&
                  d_RH.alloc((size_t)std::max(1L, m * n));
        if (!ok) return false;
        (void)hipMemset(d_Rdead.p, 0, (size_t)std::max(1L, n));
        (void)hipMemset(d_Tau.p, 0, sizeof(double) * (size_t)std::max(1L, n));
        (void)hipMemset(d_Rboff.p, 0, sizeof(long long));
        memset(&c, 0, sizeof c);
        c.fs = d_fs.p; c.fnum = d_nm.p; c.Farena = d_F.p; c.Carena = d_C.p; c.Tws = d_T.p; c.tslot = d_tslot.p;
        c.Gp = d_Gp.p; c.gp_slabs = stm_ca_slabs(s);
        {
            // magnitude guard of the panel kernels from the front's own entries (the full path takes it from A's values)
            double amax = 0;
            for (double v : Fd) { const double a = std::fabs(v); if (a > amax && std::isfinite(a)) amax = a; }
            double sg = 1.0;
            if (amax > 0) { const int e = std::ilogb(amax); if (e > 300 || e < -300) sg = std::ldexp(1.0, -e); }
            std::vector<double> sv = {sg, 1.0 / sg};
            if (!d_sig.up(sv)) return false;
            c.sig = d_sig.p;
        }
        c.Stair = d_St.p; c.Tau = d_Tau.p; c.Rdead = d_Rdead.p; c.Rhoff = d_Rhoff.p; c.Rboff = d_Rboff.p;
        c.dbg = getenv("STMMQR_DBG") ? atoi(getenv("STMMQR_DBG")) : 0;
        c.tune = getenv("STMMQR_TUNE") ? atoi(getenv("STMMQR_TUNE")) : 0;
        { stmmqr_options o; stmmqr_get_options(&o); c.tall_min = o.tall_min_rows; c.panel_algo = o.panel_algo; c.ca_min_rows = STM_CA_MIN_ROWS; }
#ifdef STMMQR_STAMPS
        if (!d_dbg.alloc(1024)) return false;
        (void)hipMemset(d_dbg.p, 0, 1024 * sizeof(unsigned long long));
        c.dbgbuf = d_dbg.p;
#endif
        return true;
    }
    bool push_num() { return hipMemcpy(d_nm.p, &nm, sizeof nm, hipMemcpyHostToDevice) == hipSuccess; }
    bool pull_num() { return hipMemcpy(&nm, d_nm.p, sizeof nm, hipMemcpyDeviceToHost) == hipSuccess; }
    bool pull_F(double *F, long ldf)
    {
        std::vector<double> Fd((size_t)s.ld * std::max(1L, n));
        if (!d_F.down(Fd.data(), Fd.size())) return false;
        for (long j = 0; j < n; j++) memcpy(F + j * ldf, &Fd[(size_t)j * s.ld], sizeof(double) * (size_t)m);
        return true;
    }
};

// device time of the kernels of the last seam call (hipEvents on the null stream), for the kernel micro-benchmarks
double g_seam_ms = -1.0;
struct SeamTimer {
    hipEvent_t a = nullptr, b = nullptr;
    SeamTimer()
    {
        g_seam_ms = -1.0;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, nullptr);
    }
    void stop()
    {
        if (!a || !b) return;
        float ms = 0;
        if (hipEventRecord(b, nullptr) == hipSuccess && hipEventSynchronize(b) == hipSuccess &&
            hipEventElapsedTime(&ms, a, b) == hipSuccess)
            g_seam_ms = ms;
    }
    ~SeamTimer()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

int lds_for(long m) { return (int)std::min(15360L, (((m + 63) & ~63L) | 1) * STM_NB + 64); }

}  // namespace

extern "C" {

void stmmqr_get_options(stmmqr_options *o);

double stmmqr_last_seam_ms(void) { return g_seam_ms; }

// ---------------------------------------------------------------------------------------------
// qr_front (SparseQR.h:209-226 / SparseQR_factorize.c:1383-1618)
// ---------------------------------------------------------------------------------------------
stm_long stmmqr_front(stm_long m, stm_long n, stm_long npiv, double tol, stm_long ntol, double *F, stm_long *Stair,
                      char *Rdead, double *Tau, double *flops)
{
    if (m < 0 || n < 0 || !device_ready()) return -1;
    OneFront X;
    if (!X.init(m, n, npiv, F, m, Stair)) return -1;
    X.c.tol = tol;
    X.c.ntol = (int)std::min<stm_long>(ntol, 1 << 30);
    stmmqr_options opt;
    stmmqr_get_options(&opt);
    const bool big = n >= opt.big_front_cols && m >= 64;
    int e = 0;
    SeamTimer timer;
    if (!big) {
        e = stm_launch_front_wg(X.c, X.d_flist.p, 1, lds_for(m), nullptr);
    } else {
        // as the plan does: a front of >= 3 row slabs takes the row-parallel update (and may leave T to it)
        const int msl = (int)((m + STM_UPD_SLAB - 1) / STM_UPD_SLAB);
        const bool split = opt.split_update && msl >= 3;
        Buf<double> d_Wp;
        if (split && !d_Wp.alloc((size_t)((n + 31) / 32 + 1) * (size_t)msl * STM_NB * 32)) return -1;
        // the launchers take the panel index of every front of a launch from a device list
        std::vector<int> hp((size_t)std::max(1, X.s.npanels));
        for (size_t q = 0; q < hp.size(); q++) hp[q] = (int)q;
        Buf<int> d_pl;
        Buf<long long> d_wpoff;
        Buf<int> d_wcnt;
        const std::vector<long long> zoff(1, 0);
        const std::vector<int> zcnt((size_t)(n + 31) / 32 + 2, 0);
        if (!d_pl.up(hp) || !d_wpoff.up(zoff) || !d_wcnt.up(zcnt)) return -1;
        for (int p = 0; p < X.s.npanels && !e; p++) {
            const int k2 = (int)std::min<long>(n, (long)(p + 1) * STM_NB);
            const int ncb = (int)((n - k2 + 31) / 32);
            const int defer_ok = (ncb > 0) ? 1 : 0;
            if (stm_use_ca(X.s, p, opt.panel_algo)) e = stm_launch_panel_ca(X.c, X.d_flist.p, d_pl.p + p, 1, stm_ca_slabs(X.s), defer_ok, nullptr);
            else {
                // (as run_schedule: a short panel of the pipeline is taken by one workgroup with the panel's image in LDS)
                const int lds = stm_tall_panel(X.s, p, X.c.tall_min) ? std::max(lds_for(m), STM_NB * STM_WP_ROWS) : lds_for(m);
                e = stm_launch_panel(X.c, X.d_flist.p, d_pl.p + p, 1, stm_tall_launches(X.s, p, X.c.tall_min), defer_ok, lds, nullptr);
            }
            if (e || ncb <= 0) continue;
            if (split) e = stm_launch_update_split(X.c, X.d_flist.p, d_pl.p + p, 1, 0, ncb, msl, d_Wp.p, d_wpoff.p, d_wcnt.p, 1, nullptr);
            else e = stm_launch_update(X.c, X.d_flist.p, d_pl.p + p, 1, 0, ncb, nullptr);
        }
        if (hipDeviceSynchronize() != hipSuccess) e = -1;          // (d_Wp is released at the end of this scope)
    }
    timer.stop();
    if (e || hipDeviceSynchronize() != hipSuccess) return -1;
#ifdef STMMQR_STAMPS
    if (X.c.dbg & 48) {
        unsigned long long hb[16];
        (void)hipMemcpy(hb, X.c.dbgbuf, sizeof hb, hipMemcpyDeviceToHost);
        fprintf(stderr, "[panel cycles] stage-in %llu  columns %llu  write-back %llu  flush %llu  gram %llu  apply %llu  |", hb[0],
                hb[1], hb[2], hb[3], hb[4], hb[5]);
        fprintf(stderr, " tall: load %llu apply %llu store %llu factor %llu gram %llu", hb[6], hb[7], hb[8], hb[9], hb[10]);
        fprintf(stderr, "\n");
        if (X.c.dbg & 32) {
            std::vector<unsigned long long> tl(1024);
            (void)hipMemcpy(tl.data(), X.c.dbgbuf, 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull;
            for (int b = 0; b < 8; b++) if (tl[16 + 64 * b] && tl[16 + 64 * b] < t0) t0 = tl[16 + 64 * b];
            for (int b = 0; b < 8; b++) {
                if (!tl[16 + 64 * b]) continue;
                fprintf(stderr, "[timeline panel 1, group %d, us]", b);
                for (int i = 0; i < 64; i++)
                    if (tl[16 + 64 * b + i]) fprintf(stderr, " %d:%.2f", i, 0.01 * (double)(long long)(tl[16 + 64 * b + i] - t0));
                fprintf(stderr, "\n");
            }
        }
    }
#endif
    if (!X.pull_num() || !X.pull_F(F, m)) return -1;
    std::vector<int> st32((size_t)std::max<stm_long>(1, n));
    if (!X.d_St.down(st32.data(), (size_t)n)) return -1;
    for (stm_long j = 0; j < n; j++) Stair[j] = st32[j];
    if (Tau && !X.d_Tau.down(Tau, (size_t)n)) return -1;
    if (Rdead && X.s.fp > 0 && !X.d_Rdead.down(Rdead, (size_t)X.s.fp)) return -1;
    if (flops) *flops = X.nm.flops;
    return X.nm.rank;
}

stm_long qr_front(stm_long m, stm_long n, stm_long npiv, double tol, stm_long ntol, stm_long fchunk, double *F,
                  stm_long *Stair, char *Rdead, double *Tau, double *W, double *wscale, double *wssq,
                  stm_sparse_common *cc)
{
    (void)fchunk; (void)W; (void)wscale; (void)wssq; (void)cc;
    // the reference ORs into Rdead (it never clears it): keep entries that are already 1
    std::vector<char> rd((size_t)std::max<stm_long>(1, n), 0);
    stm_long r = stmmqr_front(m, n, npiv, tol, ntol, F, Stair, rd.data(), Tau, nullptr);
    const stm_long np = std::min(n, std::max<stm_long>(0, npiv));
    if (Rdead) for (stm_long k = 0; k < np; k++) if (rd[k]) Rdead[k] = 1;
    return r;
}

// ---------------------------------------------------------------------------------------------
// qr_larftb, all four methods (SparseQR.h:255-268 / SparseQR_factorize.c:1851-1904; callers: qr_front :1473-1594 with
// QR_QTX, qr_panel SparseQR.c:1659,1663 with any of the four).
//   left side  (QR_QTX 0 / QR_QX 1):  C (m x n, ldc) <- (I - V T V')' C  /  (I - V T V') C,  V m x k unit lower trapezoidal
//   right side (QR_XQT 2 / QR_XQ 3):  C (m x n, ldc) <- C (I - V T V')'  /  C (I - V T V'),  V n x k
// The right-side forms are the left-side ones on C' (C Q' = (Q C')', C Q = (Q' C')'): the transposition happens on the
// host copy, the reflectors run through the same MFMA update kernel as the factorization (k_update; its no-transpose
// twin k_update_n for Q).  More than 32 reflectors are applied 32 at a time: first block first for Q', last first for Q.
// ---------------------------------------------------------------------------------------------
static int larftb_left(bool notrans, stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, const double *V,
                       const double *Tau, double *C)
{
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    if (!device_ready()) return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: no usable gfx950 device (there is no CPU fallback)");
    if (ldc < m || ldv < m) return stm_fail(STMMQR_ERR_INVALID, "qr_larftb: leading dimension smaller than the row count");
    // device image [V | C] in one column-major array
    OneFront X;
    if (!X.init(m, k + n, 0, nullptr, m, nullptr)) return stm_fail(STMMQR_ERR_OUT_OF_MEMORY, "qr_larftb: device allocation failed");
    std::vector<double> Fd((size_t)X.s.ld * (size_t)(k + n), 0.0);
    for (stm_long j = 0; j < k; j++) memcpy(&Fd[(size_t)j * X.s.ld], V + j * ldv, sizeof(double) * (size_t)m);
    for (stm_long j = 0; j < n; j++) memcpy(&Fd[(size_t)(k + j) * X.s.ld], C + j * ldc, sizeof(double) * (size_t)m);
    if (hipMemcpy(X.d_F.p, Fd.data(), Fd.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: upload failed");
    std::vector<double> tau((size_t)(k + n), 0.0);
    for (stm_long j = 0; j < k; j++) tau[j] = Tau[j];
    if (hipMemcpy(X.d_Tau.p, tau.data(), tau.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: upload failed");
    const stm_long nblk = (k + STM_NB - 1) / STM_NB;
    for (stm_long b = 0; b < nblk; b++) {
        // Q' C = H_k ... H_1 C: first block first;  Q C = H_1 ... H_k C: last block first
        const stm_long k1 = (notrans ? nblk - 1 - b : b) * STM_NB;
        const int nb = (int)std::min<stm_long>(STM_NB, k - k1);
        PanelDesc &pd = X.nm.pd[0];
        pd.pg1 = (int)std::min(k1, m); pd.pt = (int)m; pd.pk1 = (int)k1; pd.pnb = nb; pd.pc0 = (int)k;
        for (int j = 0; j < STM_NB; j++)
            pd.pdiag[j] = (j < nb && k1 + j < m && tau[k1 + j] != 0.0) ? (int)(k1 + j) : STM_BIGROW;
        // a reflector with tau = 0 is the identity; one whose diagonal falls below m does not exist
        if (!X.push_num()) return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: upload failed");
        if (stm_launch_larft(X.c, 0, nullptr)) return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: T kernel launch failed");
        const int ncb = (int)((n + 31) / 32);
        const int e = notrans ? stm_launch_update_notrans(X.c, 0, ncb, nullptr)
                              : stm_launch_update(X.c, X.d_flist.p, X.d_flist.p /* panel 0 */, 1, 0, ncb, nullptr);
        if (e) return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: update kernel launch failed");
    }
    if (hipDeviceSynchronize() != hipSuccess) return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: kernel failed");
    if (!X.d_F.down(Fd.data(), Fd.size())) return stm_fail(STMMQR_ERR_DEVICE, "qr_larftb: download failed");
    for (stm_long j = 0; j < n; j++) memcpy(C + j * ldc, &Fd[(size_t)(k + j) * X.s.ld], sizeof(double) * (size_t)m);
    return 0;
}

int stmmqr_larftb(int method, stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, const double *V,
                  const double *Tau, double *C)
{
    if (m <= 0 || n <= 0 || k <= 0) return 0;                              // (SparseQR_factorize.c:1864-1867)
    if (method < 0 || method > 3) return stm_fail(STMMQR_ERR_INVALID, "qr_larftb: unknown method");
    if (!V || !Tau || !C) return stm_fail(STMMQR_ERR_INVALID, "qr_larftb: null operand");
    if (method == 0 || method == 1) return larftb_left(method == 1, m, n, k, ldc, ldv, V, Tau, C);
    // right side: work on Ct = C' (n x m)
    if (ldc < m) return stm_fail(STMMQR_ERR_INVALID, "qr_larftb: leading dimension smaller than the row count");
    std::vector<double> Ct((size_t)n * (size_t)m);
    for (stm_long j = 0; j < n; j++)
        for (stm_long i = 0; i < m; i++) Ct[(size_t)j + (size_t)i * (size_t)n] = C[i + j * ldc];
    // C Q' = (Q C')' (method 2 -> no transpose), C Q = (Q' C')' (method 3 -> transpose)
    const int e = larftb_left(method == 2, n, m, k, n, ldv, V, Tau, Ct.data());
    if (e) return e;
    for (stm_long j = 0; j < n; j++)
        for (stm_long i = 0; i < m; i++) C[i + j * ldc] = Ct[(size_t)j + (size_t)i * (size_t)n];
    return 0;
}

int stmmqr_larftb_qtx(stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, const double *V,
                      const double *Tau, double *C)
{
    return stmmqr_larftb(0, m, n, k, ldc, ldv, V, Tau, C);
}

void qr_larftb(int method, stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, double *V, double *Tau,
               double *C, double *W, stm_sparse_common *cc)
{
    (void)W;                                    // (T and the dlarfb workspace live on the device)
    const int e = stmmqr_larftb(method, m, n, k, ldc, ldv, V, Tau, C);
    if (e) stm_cc_set_status(cc, e);            // never silent: cc->status < SPARSE_OK + stmmqr_last_error()
}

// ---------------------------------------------------------------------------------------------
// qr_fcsize / qr_csize (scalars), qr_cpack, qr_rhpack
// ---------------------------------------------------------------------------------------------
stm_long qr_fcsize(stm_long m, stm_long n, stm_long npiv, stm_long g)
{
    const stm_long cn = n - npiv, cm = std::min(m - g, cn);
    return (cm * (cm + 1)) / 2 + cm * (cn - cm);
}
stm_long qr_csize(stm_long c, stm_long *Rp, stm_long *Cm, stm_long *Super)
{
    const stm_long cm = Cm[c], cn = (Rp[c + 1] - Rp[c]) - (Super[c + 1] - Super[c]);
    return (cm * (cm + 1)) / 2 + cm * (cn - cm);
}

stm_long qr_cpack(stm_long m, stm_long n, stm_long npiv, stm_long g, double *F, double *C)
{
    const stm_long cn = n - npiv, cm = std::min(m - g, cn);
    if (cm <= 0 || cn <= 0) return 0;
    if (!device_ready()) return -1;
    OneFront X;
    if (!X.init(m, n, npiv, F, m, nullptr)) return -1;
    X.nm.rank = (int)g;
    if (!X.push_num()) return -1;
    std::vector<int> parts(1, 8);
    if (!X.d_parts.up(parts)) return -1;
    if (stm_launch_cpack(X.c, X.d_flist.p, X.d_parts.p, 1, 8, nullptr) || hipDeviceSynchronize() != hipSuccess) return -1;
    if (!X.pull_num()) return -1;
    if (!X.d_C.down(C, (size_t)(cm * (cm + 1) / 2 + cm * (cn - cm)))) return -1;
    return X.nm.cm;
}

stm_long qr_rhpack(int keepH, stm_long m, stm_long n, stm_long npiv, stm_long *Stair, double *F, double *R,
                   stm_long *p_rm)
{
    if (m <= 0 || n <= 0) { if (p_rm) *p_rm = 0; return 0; }
    if (!keepH || !device_ready()) return -1;   // the reference always keeps H (SparseQR_analyze.c:205)
    OneFront X;
    if (!X.init(m, n, npiv, F, m, Stair)) return -1;
    stm_long rm = 0;                            // live pivots (integer input of the copy kernel)
    for (stm_long k = 0; k < X.s.fp; k++) if (Stair[k] != 0 && rm < m) rm++;
    X.nm.rank = (int)rm;
    if (!X.push_num()) return -1;
    std::vector<int> parts(1, 8);
    if (!X.d_parts.up(parts)) return -1;
    if (stm_launch_rh_count(X.c, X.d_flist.p, 1, nullptr)) return -1;
    if (stm_launch_rh_copy(X.c, X.d_flist.p, X.d_parts.p, 1, 8, X.d_RH.p, nullptr)) return -1;
    if (hipDeviceSynchronize() != hipSuccess || !X.pull_num()) return -1;
    std::vector<double> out((size_t)std::max(1LL, X.nm.rsize));
    if (!X.d_RH.down(out.data(), (size_t)X.nm.rsize)) return -1;
    memmove(R, out.data(), sizeof(double) * (size_t)X.nm.rsize);   // R may alias F (in-place pack)
    if (p_rm) *p_rm = rm;
    return X.nm.rsize;
}

// ---------------------------------------------------------------------------------------------
// integer helpers: host code in the reference, host code here
// ---------------------------------------------------------------------------------------------
stm_long qr_fsize(stm_long f, stm_long *Super, stm_long *Rp, stm_long *Rj, stm_long *Sleft, stm_long *Child,
                  stm_long *Childp, stm_long *Cm, stm_long *Fmap, stm_long *Stair)
{
    const stm_long col1 = Super[f], fp = Super[f + 1] - col1, p1 = Rp[f], fn = Rp[f + 1] - p1;
    for (stm_long j = 0; j < fn; j++) Fmap[Rj[p1 + j]] = j;
    for (stm_long j = 0; j < fn; j++) Stair[j] = j < fp ? Sleft[col1 + j + 1] - Sleft[col1 + j] : 0;
    for (stm_long q = Childp[f]; q < Childp[f + 1]; q++) {
        const stm_long c = Child[q], pc = Rp[c] + Super[c + 1] - Super[c];
        for (stm_long ci = 0; ci < Cm[c]; ci++) Stair[Fmap[Rj[pc + ci]]]++;
    }
    stm_long fm = 0;
    for (stm_long j = 0; j < fn; j++) { const stm_long t = fm; fm += Stair[j]; Stair[j] = t; }
    return fm;
}

void qr_hpinv(stm_qr_symbolic *S, stm_qr_numeric *N, stm_long *W)
{
    const stm_long nf = S->nf, m = S->m, n = S->n;
    stm_long row1 = 0, row2 = m, maxfm = 0;
    for (stm_long i = S->Sleft[n]; i < m; i++) W[i] = --row2;
    for (stm_long f = 0; f < nf; f++) {
        stm_long *Hi = N->Hii + S->Hip[f];
        const stm_long rm = N->Hr[f], fm = N->Hm[f];
        for (stm_long i = 0; i < rm; i++) W[Hi[i]] = row1++;
        const stm_long cn = (S->Rp[f + 1] - S->Rp[f]) - (S->Super[f + 1] - S->Super[f]);
        const stm_long cm = std::min(fm - rm, cn);
        maxfm = std::max(maxfm, fm);
        for (stm_long i = fm - 1; i >= rm + cm; i--) W[Hi[i]] = --row2;
    }
    N->maxfm = maxfm;
    for (stm_long i = 0; i < m; i++) N->HPinv[i] = W[S->PLinv[i]];
    for (stm_long f = 0; f < nf; f++) {
        stm_long *Hi = N->Hii + S->Hip[f];
        for (stm_long i = 0; i < N->Hm[f]; i++) Hi[i] = W[Hi[i]];
    }
}

// ---------------------------------------------------------------------------------------------
// qr_stranspose2 (SparseQR_factorize.c:755-785): gather map on the host, value gather on the device
// ---------------------------------------------------------------------------------------------
void qr_stranspose2(stm_sparse_csc *A, stm_long *Qfill, stm_long *Sp, stm_long *PLinv, double *Sx, stm_long *W)
{
    const stm_long m = (stm_long)A->nrow, n = (stm_long)A->ncol;
    const stm_long *Ap = (const stm_long *)A->p, *Ai = (const stm_long *)A->i;
    const stm_long anz = Ap[n];
    if (anz <= 0 || !device_ready()) return;
    for (stm_long r = 0; r < m; r++) W[r] = Sp[r];
    std::vector<int> smap((size_t)anz);
    for (stm_long col = 0; col < n; col++) {
        const stm_long j = Qfill ? Qfill[col] : col;
        for (stm_long p = Ap[j]; p < Ap[j + 1]; p++) smap[W[PLinv[Ai[p]]]++] = (int)p;
    }
    Buf<double> dA, dS;
    Buf<int> dm;
    if (!dA.up((const double *)A->x, (size_t)anz) || !dS.alloc((size_t)anz) || !dm.up(smap)) return;
    if (stm_launch_gather_sx(dA.p, dm.p, dS.p, (int)anz, nullptr)) return;
    (void)hipDeviceSynchronize();
    (void)dS.down(Sx, (size_t)anz);
}

// ---------------------------------------------------------------------------------------------
// qr_assemble (SparseQR.h:178-200 / SparseQR_factorize.c:1151-1285) on host buffers.
// Inputs as in the reference: Stair = the exclusive cumsum qr_fsize left, Fmap filled for front f.
// Runs k_setup + k_assemble for front f with its children's packed C blocks uploaded.
// ---------------------------------------------------------------------------------------------
void qr_assemble(stm_long f, stm_long fm, int keepH, stm_long *Super, stm_long *Rp, stm_long *Rj, stm_long *Sp,
                 stm_long *Sj, stm_long *Sleft, stm_long *Child, stm_long *Childp, double *Sx, stm_long *Fmap,
                 stm_long *Cm, double **Cblock, stm_long *Hr, stm_long *Stair, stm_long *Hii, stm_long *Hip, double *F,
                 stm_long *Cmap)
{
    (void)keepH;
    if (!device_ready()) return;
    const stm_long col1 = Super[f], fp = Super[f + 1] - col1, p1 = Rp[f], fn = Rp[f + 1] - p1;
    const stm_long nch = Childp[f + 1] - Childp[f];
    // local front numbering: 0 = f, 1..nch = children; local Rj/Hii slots are packed one after another
    std::vector<FrontSym> fs((size_t)nch + 1);
    std::vector<FrontNum> fnum((size_t)nch + 1);
    memset(fs.data(), 0, fs.size() * sizeof(FrontSym));
    memset(fnum.data(), 0, fnum.size() * sizeof(FrontNum));
    std::vector<int> child((size_t)std::max<stm_long>(1, nch)), rjrel, sjrel, sp32, sj0, sleft32;
    std::vector<double> Carena, sx;
    std::vector<int> hii;
    long rjpos = 0, hipos = 0;
    FrontSym &s = fs[0];
    s.ld = (int)std::max<stm_long>(2, (fm + 1) & ~1L); s.fn = (int)fn; s.fp = (int)fp; s.col1 = 0; s.rp = 0; s.hip = 0;
    s.child0 = 0; s.child1 = (int)nch; s.fm_ub = (int)fm; s.fm_est = (int)fm; s.parent = -1;
    rjpos = fn; hipos = fm;
    rjrel.assign((size_t)fn, 0);
    hii.assign((size_t)std::max<stm_long>(1, fm), 0);
    // S rows of the front, renumbered 0..ns-1; their global ids are kept to translate Hii back
    const stm_long r0 = Sleft[col1], r1 = Sleft[col1 + fp];
    s.srow0 = 0; s.srow1 = (int)(r1 - r0);
    sp32.push_back(0);
    for (stm_long r = r0; r < r1; r++) {
        for (stm_long p = Sp[r]; p < Sp[r + 1]; p++) { sjrel.push_back((int)Fmap[Sj[p]]); sx.push_back(Sx[p]); }
        sp32.push_back((int)sjrel.size());
        sj0.push_back((int)(Sj[Sp[r]] - col1));
    }
    for (stm_long j = 0; j <= fp + 1; j++) sleft32.push_back((int)(Sleft[std::min(col1 + j, col1 + fp)] - r0));
    for (stm_long q = 0; q < nch; q++) {
        const stm_long c = Child[Childp[f] + q];
        const stm_long fpc = Super[c + 1] - Super[c], fnc = Rp[c + 1] - Rp[c], cn = fnc - fpc, cm = Cm[c];
        FrontSym &cs = fs[(size_t)q + 1];
        cs.fn = (int)fnc; cs.fp = (int)fpc; cs.rp = (int)rjpos; cs.hip = (int)hipos; cs.parent = 0;
        cs.coff = (long long)Carena.size();
        child[(size_t)q] = (int)q + 1;
        rjrel.resize((size_t)(rjpos + fnc), 0);
        for (stm_long cj = 0; cj < cn; cj++) rjrel[(size_t)(rjpos + fpc + cj)] = (int)Fmap[Rj[Rp[c] + fpc + cj]];
        const stm_long csize = cm * (cm + 1) / 2 + cm * (cn - cm);
        Carena.insert(Carena.end(), Cblock[c], Cblock[c] + csize);
        // child row ids: Hii[Hip[c] + Hr[c] + ci], stored locally at hip + rank + ci with rank = 0
        fnum[(size_t)q + 1].cm = (int)cm; fnum[(size_t)q + 1].rank = 0;
        hii.resize((size_t)(hipos + cm), 0);
        for (stm_long ci = 0; ci < cm; ci++) hii[(size_t)(hipos + ci)] = (int)Hii[Hip[c] + Hr[c] + ci];
        rjpos += fnc; hipos += cm;
    }
    if (Carena.empty()) Carena.push_back(0);
    if (sx.empty()) { sx.push_back(0); sjrel.push_back(0); }
    if (sj0.empty()) sj0.push_back(0);
    Buf<FrontSym> d_fs; Buf<FrontNum> d_nm; Buf<double> d_F, d_C, d_Sx;
    Buf<int> d_child, d_rjrel, d_sjrel, d_sp, d_sj0, d_sleft, d_st, d_hii, d_cmap, d_cur, d_flist, d_parts;
    const int nparts = (int)std::min(512L, std::max(4L, ((long)fm * (long)fn + 16383) / 16384));    // (as the plan does)
    std::vector<int> zero1(1, 0), parts(1, nparts);
    std::vector<int> hii_g(hii);       // S rows get LOCAL ids on the device; translate after download
    bool ok = d_fs.up(fs) && d_nm.up(fnum) && d_F.alloc((size_t)s.ld * (size_t)std::max<stm_long>(1, fn)) && d_C.up(Carena) &&
              d_Sx.up(sx) && d_child.up(child) && d_rjrel.up(rjrel) && d_sjrel.up(sjrel) && d_sp.up(sp32) &&
              d_sj0.up(sj0) && d_sleft.up(sleft32) && d_st.alloc((size_t)rjpos) && d_hii.up(hii_g) &&
              d_cmap.alloc((size_t)rjpos) && d_cur.alloc((size_t)rjpos) && d_flist.up(zero1) && d_parts.up(parts);
    if (!ok) return;
    (void)hipMemset(d_F.p, 0, sizeof(double) * (size_t)s.ld * (size_t)std::max<stm_long>(1, fn));
    DevCtx c;
    memset(&c, 0, sizeof c);
    c.fs = d_fs.p; c.fnum = d_nm.p; c.Farena = d_F.p; c.Carena = d_C.p; c.Sx = d_Sx.p; c.Sp = d_sp.p;
    c.Sjrel = d_sjrel.p; c.Sj0 = d_sj0.p; c.Sleft = d_sleft.p; c.Child = d_child.p; c.Rjrel = d_rjrel.p;
    c.Stair = d_st.p; c.Hii = d_hii.p; c.Cmap = d_cmap.p; c.Cursor = d_cur.p;
    SeamTimer timer;
    if (stm_launch_setup(c, d_flist.p, 1, nullptr)) return;
    if (stm_launch_assemble(c, d_flist.p, d_parts.p, 1, nparts, nullptr)) return;
    timer.stop();
    if (hipDeviceSynchronize() != hipSuccess) return;
    // results: F, Stair (advanced), Hii rows of f, Cmap of the LAST child (what the reference leaves behind)
    std::vector<double> Fd((size_t)s.ld * (size_t)std::max<stm_long>(1, fn));
    std::vector<int> st32((size_t)rjpos), hi32((size_t)hipos), cmap32((size_t)rjpos);
    if (!d_F.down(Fd.data(), Fd.size()) || !d_st.down(st32.data(), st32.size()) || !d_hii.down(hi32.data(), hi32.size()) ||
        !d_cmap.down(cmap32.data(), cmap32.size()))
        return;
    for (stm_long j = 0; j < fn; j++) {
        memcpy(F + j * fm, &Fd[(size_t)j * s.ld], sizeof(double) * (size_t)fm);
        Stair[j] = st32[(size_t)j];
    }
    // S rows were numbered locally on the device: rows placed from S carry a local id < ns
    {
        // recompute which front rows came from S: positions start(k) .. start(k)+nS(k)-1
        stm_long *Hi = Hii + Hip[f];
        std::vector<char> fromS((size_t)std::max<stm_long>(1, fm), 0);
        for (stm_long k = 0; k < fp; k++) {
            const stm_long start = k > 0 ? st32[(size_t)k - 1] : 0;
            for (stm_long r = Sleft[col1 + k]; r < Sleft[col1 + k + 1]; r++) fromS[(size_t)(start + r - Sleft[col1 + k])] = 1;
        }
        for (stm_long i = 0; i < fm; i++) Hi[i] = fromS[(size_t)i] ? (stm_long)hi32[(size_t)i] + r0 : (stm_long)hi32[(size_t)i];
    }
    if (nch > 0) {
        const FrontSym &cs = fs[(size_t)nch];
        const stm_long cm = fnum[(size_t)nch].cm;
        for (stm_long ci = 0; ci < cm; ci++) Cmap[ci] = cmap32[(size_t)(cs.rp + cs.fp + ci)];
    }
}

}  // extern "C"
