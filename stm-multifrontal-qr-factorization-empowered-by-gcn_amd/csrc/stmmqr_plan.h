// stmmqr_plan.h -- the plan object and the helpers shared by the host translation units of libstmmqr_hip.so (local to the library):
//   stmmqr_host.cpp      planner, step scheduler, factorization entry points, download
//   stmmqr_multi.cpp     contribution blocks / panels in and out of a plan, shared fronts, the RCCL transport
//   stmmqr_rfactor.cpp   Q-apply and triangular solves on the resident factors
//   stmmqr_seam.cpp      the drop-in seam qr_factorize with the reference's structs, its plan cache
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>
#ifdef __linux__
#include <sys/mman.h>
#endif
#include <dlfcn.h>

#include "../../include/stmmqr_hip.h"
#include "stmmqr_device.h"
#include "stmmqr_kernels.h"
#include "stmmqr_internal.h"


extern thread_local std::string g_err;
extern stmmqr_options g_opt;
extern size_t g_chunk[4];

// offsets inside the reference's sparse_common for the stock LP64 build; verified against the real header
// by tests/test_abi_layout.py where /root/reference is present.
extern stm_common_layout g_layout;

int fail(int code, const std::string &msg);

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return fail(e_ == hipErrorOutOfMemory ? STMMQR_ERR_OUT_OF_MEMORY : STMMQR_ERR_DEVICE,        \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                              \
    } while (0)
#define LCHK(expr)                                                                                        \
    do {                                                                                                  \
        int e_ = (expr);                                                                                  \
        if (e_ != 0) return fail(STMMQR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString((hipError_t)e_)); \
    } while (0)

inline double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; n = 0; return (int)e; }
        return 0;
    }
    int upload(const std::vector<T> &h, hipStream_t st)
    {
        int e = alloc(h.size());
        if (e) return e;
        if (!h.empty()) return (int)hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    ~DevBuf() { release(); }
};

struct Level {                           // tree level of a group: what the solve / Q-apply kernels walk (f1)
    int all_off = 0, n_all = 0;          // every front of the level (small first, then big by npanels desc)
    int n_small = 0, n_big = 0;
};

// One step of the factorization timeline.  Every front starts at the step after the last of its children has finished
// (a small front takes one step, a big one a step per panel), so a front deep in a short branch does not wait for the
// tallest front of its tree level: at every step the launches cover all the big fronts that are in flight, each at its
// own panel.  Everything here is symbolic (lists built once per plan).
struct Step {
    int start_off = 0, n_start = 0, n_small = 0;      // fronts starting here (small first, then big): set up + assembled
    int asm_parts_off = 0, asm_maxparts = 1, lds_small = 0;
    int act_off = 0, plist_off = 0, n_act = 0;        // big fronts in flight + the panel each is at
    int wp_off = 0;                                   // (index into d_wlists) their slices of the update workspace
    int nsub = 1, nca = 1, nca_use = 0, npipe_use = 0, maxcb = 0, maxsl = 0, split = 0;
    int lds_big = 0;       // dynamic LDS of the panel launch when every front may take the one-workgroup panel (recovery, tests)
    int lds_plan = 0;      // ... when only the fronts planned for it do (the pipeline groups need the update's LDS only)
    // the fronts in flight are listed in three classes: [0, n_norm) every panel's update on all trailing columns;
    // then the sweep fronts (is_pair; w = 2 or 4 panels per sweep, stmmqr_plan::sweep) by panel number mod w: class r updates the
    // column blocks 0 .. w-1-r only (the columns of the next panels), the last class then applies the w last panels at once to
    // everything beyond
    int n_norm = 0, n_pk[4] = {0, 0, 0, 0};
    int maxsl_pk[4] = {0, 0, 0, 0}, maxcbp_po = 0;
    int n_sweep() const { return n_pk[0] + n_pk[1] + n_pk[2] + n_pk[3]; }
    int cpk_off = 0, cpk_parts_off = 0, n_cpk = 0, cpk_maxparts = 1;   // big fronts whose last panel runs here
    // slab recycling: the fronts whose packed R+H block is staged at the end of this step (the small fronts that started here and
    // the big fronts packed here, kept fronts excluded) + copy parts
    int rhp_off = 0, rhp_parts_off = 0, n_rhp = 0, rhp_maxparts = 1;
};


struct stmmqr_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    // look-ahead: the panel chain runs on `stream` (high priority), everything that does not feed the next panel on `side`
    hipStream_t side = nullptr;                               // (the device's shared side stream: side_stream_for)
    std::vector<hipEvent_t> ev_main, ev_prep, ev_side;        // one of each per timeline step (no timing)
    hipEvent_t ev[8] = {};
    // detail timing: one event pair per launch category and level step, recorded on the plan's stream WITHOUT any
    // synchronisation (the schedule runs exactly as in the timed region); the pairs are read after the final sync
    struct EvPair { hipEvent_t a, b; int cat, step; };
    std::vector<EvPair> evpairs;
    size_t evused = 0;
    long m = 0, n = 0, anz = 0, nf = 0, maxfn = 0, rjsize = 0, hisize = 0;
    int do_rank = 1;
    int ca_min = STM_CA_MIN_ROWS;                      // (env STMMQR_CA_MIN at plan time: experiments)
    int plan_algo = 0;                                 // g_opt.panel_algo when the schedule was built
    int tall_min = STM_TALL_MIN;                       // g_opt.tall_min_rows when the schedule was built
    int tune = 0;                                      // env STMMQR_TUNE when the schedule was built (measurement sweeps)
    std::vector<long> Sp, Sj, Qfill, PLinv, Sleft, Child, Childp, Super, Rp, Rj, Post, Hip, Fm;
    bool has_qfill = false;
    std::vector<FrontSym> fs;
    std::vector<std::vector<Level>> glevels;   // [group][level]
    std::vector<std::vector<Step>> gsteps;     // [group][step]
    std::vector<long long> wlists;             // host copy of d_wlists
    std::vector<char> pair_front;              // per front: takes the pair / quad update (plan time)
    int sweep = 2;                             // panels per sweep of those fronts: 2 (k_upd_w2 / y2 / c2) or 4 (k_upd_wq / yq / cq)
    std::vector<int> group;                    // per front: phase on this device, -1 = elsewhere
    std::vector<int> h_tslot;                  // host copy of d_tslot
    std::vector<char> shared;                  // per front: STMMQR_GROUP_SHARED -- alone in its group, driven step by step
                                               //  (stmmqr_factorize_step), its trailing column blocks shared with other plans
    std::vector<char> has_c;                   // per front: its packed contribution block has a slot in the C arena of this plan
                                               //  (the front is factorized here, or it is a child of one that is: assign_arenas)
    std::vector<long long> c_slot;             // ... and the size of that slot in doubles (the symbolic bound of csize)
    int own_off = 0, n_own = 0;
    // ---- slab recycling (the reference's stack discipline, SparseQR_factorize.c:405-422,925-933, re-cast for a timeline of steps):
    // a front's slab lives from the step it starts to the step its contribution block is packed and its R+H block staged into the
    // R+H arena; a contribution block from there to the step its parent starts.  Offsets are assigned by an address-ordered
    // first-fit over that timeline (assign_arenas_timeline), all symbolic.  Fronts in `kept` keep their slab (never staged: their
    // packed block is produced on the fly when the factors are downloaded) -- chosen where that makes the peak smaller (a root
    // front that IS most of the factors).  Only plans that hold the whole tree in one group recycle (sharded plans: as before).
    bool recycle = false;
    long maxstack = 0;                   // QRsym->maxstack (0: unknown)
    std::vector<char> kept;              // per front
    std::vector<int> f_t0, f_t1, c_t1;   // slab: [f_t0, f_t1]; contribution block: [f_t1, c_t1]  (steps of group 0)
    long long rh_cap = 0;                // capacity of the R+H arena (doubles)
    long long rh_est_total = 0;          // all packed R+H blocks if no pivot column dies (symbolic; exact for full-rank input)
    int rh_grow = 0;                     // 0: the arena is sized from that estimate; 1: from the hard bounds (it overflowed once)
    long long scr_doubles = 0;           // scratch of the resident-factor operations: the widest tree level in front form
    std::vector<FrontSym> fs_scr;        // FrontSym with foff into that scratch (kept fronts: their own slab, relative to it)
    bool scr_all = false, scr_valid = false;   // the scratch holds every front (rebuilt once per factorization) / is up to date
    bool overflowed = false;             // a factorization did not fit the R+H arena at its hard bound: this plan does not recycle
    bool early_end = false;              // the schedule of group 0 stops every front at the panel where it is expected to run out of rows
    bool full_schedule = false;          // ... it did not hold once (rank-deficient fronts): this plan schedules every panel from now on
    bool early_phased = false;           // ... the caller of the phased interface asked for the cut schedule (stmmqr_plan_set_early_end)
    bool early_end_failed = false;       // ... the last factorization found a front unfinished at its last scheduled panel
    bool arena_overflow = false;         // the last factorization did not fit the arena (it is repeated with a larger one / without)
    std::vector<int> lists;              // host copy of d_lists
    int post_off = 0, rh_parts_off = 0, rh_maxparts = 1;
    long long farena = 0, carena = 0;
    int tslots = 1;
    int gp_slabs = 1;                    // Gram-based panel: max slab workgroups of a front
    long long tpanels = 0;               // panels of all fronts: one kept T each (Q-apply on the resident factors)
    long long wp_doubles = 0;            // workspace of the row-parallel update (partial W blocks)
    long long wp2_doubles = 0;           // ... of the side stream's copy: steps with pair-update fronts never go there
    bool pattern_set = false;
    double bytes_assemble_idx = 0;       // index bytes of the assembly (symbolic part of SURVEY 8d formula)

    DevBuf<FrontSym> d_fs;
    DevBuf<FrontNum> d_fnum;
    DevBuf<double> d_F, d_C, d_T, d_Gp, d_Tall, d_Sx, d_Ax, d_Tau, d_RH, d_Wp, d_Wp2;
    DevBuf<int> d_tslot, d_Sp, d_Sjrel, d_Sj0, d_Sleft, d_Child, d_Rjrel, d_Stair, d_Hii, d_Cmap, d_Cursor,
        d_lists, d_smap;
    DevBuf<long long> d_Rhoff;
    DevBuf<long long> d_wlists;
    DevBuf<double> d_Ypend;                    // -Y of the pair-update fronts, by absolute column block (DevCtx::Ypend)
    DevBuf<long long> d_ypoff;                 // [nf] offsets into it (-1: not a pair-update front)
    std::vector<long long> ypoff;
    long long yp_doubles = 0;
    DevBuf<double> d_msg;                // subtree exchange: message buffers (stmmqr_factorize_exchange, grown on demand)
    DevBuf<int> d_wcnt, d_wcnt2;         // per column block of the update workspaces: slab tickets (zero between launches)
    DevBuf<int> d_wflag, d_wflag2;       // ... fused update: step + 1 once W2 of the column block is in its slot
    DevBuf<int> d_abort;
    size_t wcnt_n = 1;
    DevBuf<long long> d_Rboff, d_total;
    DevBuf<long long> d_rhtop, d_fin;    // slab recycling: {bump pointer, overflow word}; Post-order offsets of the packed blocks
    DevBuf<char> d_kept;
    DevBuf<double> d_scr, d_bounce;      // resident-factor scratch (one tree level in front form); download window
    DevBuf<FrontSym> d_fs_scr;
    DevBuf<unsigned long long> d_dbg, d_amax;
    DevBuf<double> d_sig;                           // {sg, 1/sg}: magnitude guard of the panel kernels
    DevBuf<char> d_Rdead;

    // results of the last factorization
    bool factored = false, begun = false, first_group = true;
    bool whole_call = false;             // inside stmmqr_factorize_device (which recovers the WHOLE factorization itself)
    bool panel_wait_failed = false;      // a bounded inter-workgroup wait of a panel kernel ran out in the last factorization
    bool serial_panels = false;          // recovery: every panel by ONE workgroup (no inter-workgroup waits at all)
    long long rh_total = 0;
    long rank = 0;
    std::vector<FrontNum> h_fnum;
    // Q-apply / solve on the resident factors
    DevBuf<int> d_Rj, d_PLinv, d_Qfill, d_Wmap, d_err;
    DevBuf<double> d_W, d_Xs, d_Io, d_Xf, d_Wq, d_Xall, d_Yall, d_U, d_Xr;
    DevBuf<int> d_rowbase;                          // rows of R above each front (R rows are numbered front by front)
    std::vector<int> level_lds_rt;                  // dynamic LDS of k_rtsolve per level
    DevBuf<int> d_Dq;
    DevBuf<QbDesc> d_qb;
    // grouped split Q-apply (k_qbig_step4): T4 of every group of four panels of every split front, built at the first Q-apply after a
    // factorization (t4_valid); t4_ok: the buffers exist (they are allocated at that first use; no room: the per-panel launches stay)
    std::vector<Qt4ItemHost> t4items;
    std::vector<int> t4fronts;
    std::vector<long long> t4dqo, qbt4off;
    long long t4_doubles = 0, dq4_ints = 0, wq4_doubles = 0;
    DevBuf<Qt4ItemHost> d_t4items;
    DevBuf<int> d_t4fronts, d_Dq4;
    DevBuf<long long> d_t4dqo, d_qbt4off;
    DevBuf<double> d_T4, d_Wq4;
    bool t4_valid = false, t4_ok = false, t4_tried = false;
    struct QbLevel { int off = 0, n = 0, max_np = 0, max_nslab = 0, max_fm = 0, max_rsteps = 0, t4i_off = 0, t4i_n = 0;
                     int live_np = 0, live_rsteps = 0; };     // (live_*: of the current factorization, ensure_rowmap)
    std::vector<QbDesc> h_qb;            // host copy of d_qb (np_live is refreshed per factorization)
    std::vector<char> t4_level_valid;          // T4 of the level's split fronts is built (per level: with per-level scratch only the
                                               //  level at hand is in front form)
    DevBuf<int> d_Rm;                               // rows of R (live pivots) of the split fronts of a level (k_rbig_*)
    // several right-hand sides per launch (RhsBatch, stmmqr_kernels.h): the per-vector buffers hold rhs_cap vectors at these strides
    int rhs_cap = 1;
    long long xf_doubles = 1, wq_doubles = 1;
    std::vector<QbLevel> level_qbig;               // descriptors (d_qb) of the fronts of each level that take the split Q-apply
    hipGraphExec_t graph_exec = nullptr;           // options.use_graph: the captured schedule of group 0
    double graph_tol = 0; int graph_ntol = 0, graph_dbg = 0; long long graph_opt = 0; long graph_nlaunch = 0;
    long sched_gen = 0, graph_gen = -1;            // schedule generation (bumped by every build_schedule) / the captured one
    bool rowmap_ready = false;         // d_Wmap belongs to the factorization currently held
    std::vector<int> level_lds_qa, level_lds_qa_all, level_lds_rs;   // dynamic LDS of k_qapply(_t) / k_rsolve per level of group 0
                                                                     // (_all: the unblocked kernel takes the split fronts too)
    double last_tol = 0;
    long last_ntol = 0;
    stmmqr_stats stats = {};

    // device memory held right now (every DevBuf of the plan)
    double device_bytes() const
    {
        double b = 0;
        auto add = [&](const auto &buf) { b += (double)buf.n * sizeof(*buf.p); };
        add(d_fs); add(d_fnum); add(d_F); add(d_C); add(d_T); add(d_Gp); add(d_Tall); add(d_Sx); add(d_Ax); add(d_Tau); add(d_RH);
        add(d_Wp); add(d_Wp2); add(d_tslot); add(d_Sp); add(d_Sjrel); add(d_Sj0); add(d_Sleft); add(d_Child); add(d_Rjrel);
        add(d_Stair); add(d_Hii); add(d_Cmap); add(d_Cursor); add(d_lists); add(d_smap); add(d_Rhoff); add(d_wlists);
        add(d_wcnt); add(d_wcnt2); add(d_wflag); add(d_wflag2); add(d_Rboff); add(d_Rdead); add(d_Ypend); add(d_ypoff);
        add(d_rhtop); add(d_fin); add(d_kept); add(d_scr); add(d_bounce); add(d_fs_scr);
        return b;
    }
    DevCtx ctx() const
    {
        DevCtx c;
        c.fs = d_fs.p; c.fnum = d_fnum.p; c.Farena = d_F.p; c.Carena = d_C.p; c.Tws = d_T.p; c.tslot = d_tslot.p;
        c.Tall = d_Tall.p;
        c.Gp = d_Gp.p; c.gp_slabs = gp_slabs; c.sig = d_sig.p; c.panel_algo = serial_panels ? 1 : plan_algo; c.ca_min_rows = ca_min;
        c.Sx = d_Sx.p; c.Sp = d_Sp.p; c.Sjrel = d_Sjrel.p; c.Sj0 = d_Sj0.p; c.Sleft = d_Sleft.p;
        c.Child = d_Child.p; c.Rjrel = d_Rjrel.p; c.Stair = d_Stair.p; c.Tau = d_Tau.p; c.Hii = d_Hii.p;
        c.Rdead = d_Rdead.p; c.Cmap = d_Cmap.p; c.Cursor = d_Cursor.p; c.Rhoff = d_Rhoff.p; c.Rboff = d_Rboff.p;
        c.tol = last_tol; c.ntol = (int)last_ntol;
        c.dbg = getenv("STMMQR_DBG") ? atoi(getenv("STMMQR_DBG")) : 0;
        c.sweep = sweep;
        c.tune = tune;                                            // (env STMMQR_TUNE when the schedule was built)
        c.Ypend = d_Ypend.p; c.ypoff = d_ypoff.p;
        c.rh_top = recycle ? d_rhtop.p : nullptr; c.rh_cap = rh_cap;
        if (serial_panels) c.dbg = (c.dbg & ~(2048 | 4096)) | 256;   // the one-workgroup LDS / in-place panel for every panel
        c.tall_min = tall_min;
        c.cbskip = 0;
        c.dbgbuf = d_dbg.p;
        c.abort = d_abort.p;
        return c;
    }
    ~stmmqr_plan()
    {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &q : evpairs) { if (q.a) (void)hipEventDestroy(q.a); if (q.b) (void)hipEventDestroy(q.b); }
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        for (auto *v : {&ev_main, &ev_prep, &ev_side})
            for (auto &e : *v)
                if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// One piece of one timeline step (stmmqr_factorize_step): what = STMMQR_STEP_* bits; the update takes the column blocks
// cb_first, cb_first + cb_stride, ... (at most cb_count of them when cb_count >= 0) of the step's fronts.
struct StepReq { int step, what, cb_first, cb_stride, cb_count; };
// planner / scheduler entry points used by the other host translation units (stmmqr_host.cpp)
int stm_run_schedule(stmmqr_plan &P, bool detail, int grp, const StepReq *req);
int stm_ensure_device(int device);
extern "C" int stm_check_c_slot(const stmmqr_plan &P, stm_long f, long long csize, const char *what);     // (defined inside the C ABI block)
