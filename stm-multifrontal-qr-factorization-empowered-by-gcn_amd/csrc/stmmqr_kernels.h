// stmmqr_kernels.h -- kernel argument block and launcher prototypes (host <-> the kernel translation units: stmmqr_kdev.h lists them)
#pragma once
#include <hip/hip_runtime_api.h>
#include "stmmqr_device.h"

// Everything a kernel needs, passed by value.  Pointers are device pointers.
struct DevCtx {
    const FrontSym *fs;        // [nf]
    FrontNum *fnum;            // [nf]
    double *Farena;            // fronts
    double *Carena;            // packed contribution blocks
    double *Tws;               // T factors of the large fronts in flight, [slots][NB*NB]
    const int *tslot;          // [nf] slot in Tws (large fronts only)
    double *Tall;              // [sum of npanels][NB*NB] T of EVERY panel, kept for the Q-apply (nullptr: not kept)
    double *Gp;                // Gram-based panel: per T slot, gp_slabs partial Gram matrices + the M mailbox, NB*NB each
    int gp_slabs;              // max slab workgroups of a front (stm_ca_slabs) over the plan
    const double *sig;         // {sg, 1/sg}: the power-of-two magnitude guard of the panel kernels (stm_larfg_guarded)
    int panel_algo;            // stmmqr_options::panel_algo (stm_use_ca)
    int ca_min_rows;           // ... panels with more estimated rows take the Gram-based kernel (panel_algo 0)
    const double *Sx;          // [anz] values of S = A(P,Q), row form
    const int *Sp;             // [m+1]
    const int *Sjrel;          // [anz] column of each S entry inside its front
    const int *Sj0;            // [m]   leftmost column of each S row
    const int *Sleft;          // [n+2]
    const int *Child;          // [nf+1]
    const int *Rjrel;          // [rjsize] for child c, slot Rp[c]+fp+cj: column of the parent
    int *Stair;                // [rjsize] HStair
    double *Tau;               // [rjsize] HTau
    int *Hii;                  // [hisize]
    char *Rdead;               // [n]
    int *Cmap;                 // [rjsize] for child c, slot Rp[c]+fp+ci: row of the parent
    int *Cursor;               // [rjsize] scratch of k_setup
    long long *Rhoff;          // [rjsize] column offsets inside a packed R+H block
    long long *Rboff;          // [nf] offset of each packed R+H block
    double tol;
    int ntol;
    unsigned long long *dbgbuf; // [64] diagnosis counters (STMMQR_DBG bit 4: phase cycle sums of STAMPS builds, panels by actual rows,
                                //  launched / useful update workgroups, refresh rounds), else unused
    int *abort;                 // set by the first bounded wait of the fused update that runs out: the others give up at once
    int tall_min;              // stmmqr_options::tall_min_rows at plan time (stm_tall_panel)
    int cbskip;                // update launches: workgroup x takes column block cb0 + x * (1 + cbskip) -- 0 everywhere except
                               //  when the trailing columns of a front are shared between plans (stmmqr_factorize_step)
    double *Ypend;             // pair update: -Y (64 x 32 per column block) of every pair-update front, by ABSOLUTE column block (column >> 5):
                               //  written by k_upd_y2, read by k_upd_c2 of the same step
    const long long *ypoff;    // [nf] offset (doubles) of a front's blocks in Ypend (-1: not a pair-update front)
    long long *rh_top;         // slab recycling: {bump pointer of the R+H arena (doubles), overflow word}; nullptr = the packed blocks are
                               //  placed by k_rh_scan at the end (Post order), every front keeps its slab until then
    long long rh_cap;          // ... capacity of that arena (doubles)
    int sweep;                 // panels per sweep of the pair / quad update fronts (2 or 4): their panel-by-panel updates bring that many column blocks
    int tune;                  // env STMMQR_TUNE, measurement sweeps only (0 = the shipped rules): bits 0-3 force 2^(x-1) slabs per
                               //  workgroup of the pair update's kernels
    int dbg;                   // env STMMQR_DBG, ablations / cross-checks only: 1 no in-panel apply (LDS panel path),
                               //  2 no T, 4 no dlarf in the LDS sub-panel, 16/32 phase timers (-DSTMMQR_STAMPS builds),
                               //  64 in-place panel path, 128 no folded norms, 256 no panel pipeline (LDS panels only),
                               //  8192 reflector-by-reflector Q-apply instead of the blocked one,
                               //  512 panel launches in chunks of env STMMQR_CHUNK fronts (default 1),
                               //  2048 column group (dbg >> 20) & 7 of every pipelined panel starts late (tests)
                               //  4096 every column group but the first gives up waiting at once (tests: recovery)
                               //  16384 no wave-pipelined panels (short panels through the multi-workgroup pipeline too)
};

int stm_configure_kernels(void);       // (stmmqr_panel.hip; calls the other translation units' stm_configure_*)
int stm_update_lds_bytes(void);
int stm_launch_sigma(const double *Ax, int anz, unsigned long long *amaxbits, double *sig, hipStream_t st);
int stm_launch_gather_sx(const double *Ax, const int *smap, double *Sx, int anz, hipStream_t st);
int stm_launch_setup(const DevCtx &c, const int *flist, int nfr, hipStream_t st);
int stm_launch_assemble(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, hipStream_t st);
int stm_launch_front_wg(const DevCtx &c, const int *flist, int nfr, int lds_doubles, hipStream_t st);
int stm_launch_panel(const DevCtx &c, const int *flist, const int *plist, int nfr, int nsub, int defer_ok, int lds_doubles, hipStream_t st);
int stm_configure_capanel(void);
int stm_launch_panel_ca(const DevCtx &c, const int *flist, const int *plist, int nfr, int nw, int defer_ok, hipStream_t st);
int stm_launch_panel_ca_pc(const DevCtx &c, const int *flist, const int *plist, int nfr, int nw, int defer_ok, const int *uflist,
                           const int *uplist, int unfr, int ucb0, int uncb, int umaxsl, const double *Wp, const long long *uwpoff, hipStream_t st);
int stm_launch_update(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, hipStream_t st);
int stm_launch_update_fused(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, double *Wp,
                            const long long *wpoff, int *wcnt, int *wflag, int epoch, int with_gram, hipStream_t st);
// passenger launches (options.lookahead = 2): the update beyond block 0 rides on the chain's own launches
int stm_launch_panel_pc(const DevCtx &c, const int *flist, const int *plist, int nfr, int nsub, int defer_ok, int lds_doubles,
                        const int *uflist, const int *uplist, int unfr, int ucb0, int uncb, int umaxsl, const double *Wp,
                        const long long *uwpoff, hipStream_t st);
int stm_launch_update_fw(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncb, int maxsl, double *Wp,
                         const long long *wpoff, int *wcnt, int *wflag, int epoch, double *Wp2, int *wcnt2, hipStream_t st);
int stm_launch_update_w(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, double *Wp,
                        const long long *wpoff, int *wcnt, hipStream_t st);
int stm_launch_update_c(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, const double *Wp,
                        const long long *wpoff, hipStream_t st);
int stm_launch_update_pair(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncbp, int maxsl, double *Wp,
                           const long long *wpoff, int *wcnt, hipStream_t st);
int stm_launch_update_quad(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncbp, int maxsl, double *Wp,
                           const long long *wpoff, int *wcnt, hipStream_t st);
int stm_launch_update_split(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, double *Wp,
                            const long long *wpoff, int *wcnt, int with_gram, hipStream_t st);
int stm_launch_larft(const DevCtx &c, int f, hipStream_t st);
int stm_launch_update_notrans(const DevCtx &c, int f, int ncb, hipStream_t st);   // qr_larftb seam, QR_QX
int stm_launch_cpack(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, hipStream_t st);
int stm_launch_rh_count(const DevCtx &c, const int *flist, int nfr, hipStream_t st);
int stm_launch_rh_scan(const DevCtx &c, const int *post, int nf, long long *rh_total, long long *outoff, hipStream_t st);
// slab recycling (stmmqr_host.cpp, timeline allocator)
int stm_launch_zero_slabs(const DevCtx &c, const int *flist, int nfr, int maxparts, hipStream_t st);
int stm_launch_rh_unpack(const DevCtx &c, const FrontSym *cs, const int *flist, int nfr, int maxparts, const char *kept, const double *RH,
                         double *scratch, hipStream_t st);
int stm_launch_rh_window(const DevCtx &c, const int *flist, int nfr, int maxparts, const long long *fin, const char *kept, const double *RH,
                         long long w0, long long w1, double *out, hipStream_t st);
int stm_launch_rh_copy(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, double *RH,
                       hipStream_t st);
// Q-apply / triangular solve on the resident factors
// Several right-hand sides per launch (QR_qmult / QR_solve take blocks of them: qr_panel, SparseQR.c:1591-1706): every kernel of
// these operations takes right-hand side blockIdx.y (or .z) of a BATCH -- the same workgroups, one set per vector, in the same
// launch -- with the per-vector buffers laid out at these strides (doubles).  The launches of one vector are a chain of ~10-25 us
// kernels that leave the GPU nearly empty, so a batch costs little more than one vector.  nb = 1 with zero strides: one vector.
struct RhsBatch { long long w, x, xf, wq, wq4, u; };

int stm_launch_qapply(const DevCtx &c, const int *flist, int nfr, int method, double *W, int lds_bytes, int *err, hipStream_t st);
int stm_launch_qapply_t(const DevCtx &c, const int *flist, int nfr, int method, double *W, int lds_bytes, hipStream_t st, int nb,
                        const RhsBatch &B);
int stm_qt4_doubles(void);
struct Qt4ItemHost { int f, g; long long off, dqo; };     // (= Qt4Item of stmmqr_resident.hip)
int stm_launch_qt4_build(const DevCtx &c, const int *fl, const long long *dqo, int nfronts, const void *items, int nitems, int *Dq4, double *T4all,
                         hipStream_t st);
int stm_launch_qapply_big4(const DevCtx &c, const QbDesc *qd, const long long *t4off, int nq, int max_npanels, int max_nslab, int max_fm,
                           int method, double *W, double *Xf, int *Dq, double *Wq4, const double *T4all, hipStream_t st, int nb,
                           const RhsBatch &B);
int stm_launch_qapply_big(const DevCtx &c, const QbDesc *qd, int nq, int max_npanels, int max_nslab, int max_fm, int method, double *W,
                          double *Xf, int *Dq, double *Wq, hipStream_t st, int nb, const RhsBatch &B);
int stm_launch_rsolve_big(const DevCtx &c, const QbDesc *qd, int nq, int max_steps, int max_nslab, const int *Rj, const double *W,
                          double *X, double *Acc, int *Lc, int *Rm, int *err, hipStream_t st, int nb, const RhsBatch &B);
int stm_launch_rsolve(const DevCtx &c, const int *flist, int nfr, const int *Rj, const double *W, double *X, int lds_bytes,
                      int *err, hipStream_t st, int nb, const RhsBatch &B);
int stm_launch_rtsolve(const DevCtx &c, const int *flist, int nfr, const double *Bp, double *U, double *Xr, const int *rowbase,
                       int lds_bytes, hipStream_t st, int nb, const RhsBatch &B);
int stm_launch_panel_msg(void *const homes[6], const long long offs[6], const long long bytes[6], void *buf, int out, hipStream_t st);
// subtree exchange: contribution blocks between plans as fixed-size messages, packed / unpacked on the device (stmmqr_pack.hip)
struct StmFrontMsg { long long off, slot; int f, pad; };      // front f at buf + off: [8 | slot | fn - fp] doubles
struct StmFrontMsgs { StmFrontMsg m[16]; };
int stm_launch_front_msg(const DevCtx &c, const StmFrontMsgs &g, int nmsg, long long max_slot, double *buf, int out, hipStream_t st);
int stm_launch_front_cols(const DevCtx &c, int f, int part, int nparts, int nown, long long max_run, double *buf, int out, hipStream_t st);
int stm_launch_perm(const double *in, const int *perm, double *out, int n, int scatter, hipStream_t st, int nb = 1, long long sin = 0,
                    long long sout = 0);
