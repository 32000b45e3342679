// stmmqr_host.cpp -- host side of libstmmqr_hip.so: symbolic planner, level scheduler, C ABI.
//
// Reference counterparts (paths relative to /root/reference/STMMQR):
//   stmmqr_plan_create      the allocation / setup half of qr_factorize   src/qr/SparseQR_factorize.c:222-498
//   run_schedule            qr_kernel's per-front loop + qr_multithreads   :791-985, SparseQR_multithreads.c:14-115
//                           (tree parallelism re-cast as level-batched launches on a HIP stream: every front
//                            of one tree level is independent, so a level = a few batched kernel launches;
//                            no blocking waits inside workers, cf. SURVEY.md 3.3 deadlock note)
//   stmmqr_plan_download    the wrap-up half of qr_factorize (:554-742) incl. qr_hpinv (:991-1060)
//   qr_factorize            the drop-in seam                                include/SparseQR.h:127-135
//
// There is NO CPU fallback in this file: every numeric operation is a kernel of the stmmqr_*.hip translation units.  If no
// gfx950 device is usable the entry points fail with STMMQR_ERR_DEVICE.
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

namespace {

thread_local std::string g_err;
stmmqr_options g_opt = {STM_NB, 64, 0, 0, 0, 1, STM_TALL_MIN, 2, 0, 4};      // (panel_algo 0: by panel height)
size_t g_chunk[4] = {32, 5000, 4, 4};     // FCHUNK, SMALL, MINCHUNK, MINCHUNK_RATIO (SparseQR.h:16-19)

// offsets inside the reference's sparse_common for the stock LP64 build; verified against the real header
// by tests/test_abi_layout.py where /root/reference is present.
stm_common_layout g_layout = {
    /* status */ 1004, /* malloc_count */ 1032, /* memory_usage */ 1040, /* memory_inuse */ 1048,
    /* blas_ok */ 1100, /* SPQR_grain */ 1104, /* SPQR_small */ 1112, /* SPQR_shrink */ 1120,
    /* SPQR_flopcount */ 1128, /* SPQR_flopcount_bound */ 1136};

int fail(int code, const std::string &msg)
{
    g_err = msg;
    if (g_opt.verbose) fprintf(stderr, "[stmmqr_hip] error %d: %s\n", code, msg.c_str());
    return code;
}

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

double now_ms()
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

}  // namespace

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
    struct QbLevel { int off = 0, n = 0, max_np = 0, max_nslab = 0, max_fm = 0, max_rsteps = 0, t4i_off = 0, t4i_n = 0; };
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

namespace {

const int LDS_CAP_DOUBLES = STM_LDS_CAP_DOUBLES;   // 128 KiB of dynamic LDS for the staged (sub-)panel, k_panel
const int LDS_CAP_SMALL = 15360;          // 120 KiB in k_front_wg (it carries 8 KiB more static LDS)

// The side stream of the look-ahead schedule: ONE per device and process, created at the first look-ahead factorization
// and kept (a CU-masked stream owns a hardware queue; creating one per plan exhausted them).  Its CU mask leaves
// STMMQR_SIDE_RESERVE compute units (default 32; the mask bits interleave over the 8 XCDs, so 4 per XCD) to the plan
// streams: the workgroups of a panel kernel need most of a CU's LDS and would otherwise wait until the side stream's
// update grid has drained.  Plans of one device that factorize at the same time share it (still ordered by events).
std::mutex g_side_mu;
hipStream_t g_side[64] = {};
bool g_side_tried[64] = {};

// (registered with atexit at the first creation, i.e. after the HIP runtime registered its own handlers: it runs before
//  them; stmmqr_shutdown() calls it too)
void destroy_side_streams()
{
    std::lock_guard<std::mutex> lock(g_side_mu);
    for (int d = 0; d < 64; d++) {
        if (g_side[d]) {
            if (hipSetDevice(d) == hipSuccess) { (void)hipStreamSynchronize(g_side[d]); (void)hipStreamDestroy(g_side[d]); }
            g_side[d] = nullptr;
        }
        g_side_tried[d] = false;
    }
}

hipStream_t side_stream_for(int device)
{
    hipStream_t *side = g_side;
    bool *tried = g_side_tried;
    static bool registered = false;
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_side_mu);
    if (tried[device]) return side[device];
    tried[device] = true;
    if (!registered) { registered = true; atexit(destroy_side_streams); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
    const int ncu = prop.multiProcessorCount;
    int reserve = getenv("STMMQR_SIDE_RESERVE") ? atoi(getenv("STMMQR_SIDE_RESERVE")) : 32;
    if (reserve < 0) reserve = 0;
    if (reserve > ncu / 2) reserve = ncu / 2;
    hipStream_t q = nullptr;
    if (reserve > 0) {
        std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
        for (int i = reserve; i < ncu; i++) mask[(size_t)i >> 5] |= 1u << (i & 31);
        if (hipExtStreamCreateWithCUMask(&q, (uint32_t)mask.size(), mask.data()) != hipSuccess) q = nullptr;
    }
    if (!q && hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) q = nullptr;
    side[device] = q;
    return q;
}

int ensure_device(int device)
{
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(STMMQR_ERR_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
    if (device >= cnt) return fail(STMMQR_ERR_DEVICE, "device index out of range");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    static bool configured = false;
    if (!configured) {
        LCHK(stm_configure_kernels());
        LCHK(stm_configure_capanel());
        configured = true;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// level schedule.  P.group[f] = g >= 0: front f is factorized here in phase g; -1: not on this device (its
// contribution block is imported).  For every group the fronts are bucketed by tree level (leaves = 0, counted
// inside the group), small ones first, large ones by decreasing panel count.  Everything is symbolic.
// ------------------------------------------------------------------------------------------------
// Offsets of the fronts (F arena) and of the packed contribution blocks (C arena) for the CURRENT groups: a front gets room
// in F when it is factorized here (group >= 0), a contribution block when its front is factorized here or arrives here
// (stmmqr_plan_import_front: a child of one of this plan's fronts).  One rank of a sharded run holds its subtrees and the
// fronts above them that it owns or shares, not the whole tree.  (Every front keeps its F until the factors are packed at the
// end of the factorization: stmmqr_factorize_finish.)
void assign_arenas(stmmqr_plan &P)
{
    long long foff = 0, coff = 0;
    std::vector<char> needc((size_t)std::max(1L, P.nf), 0);
    P.c_slot.assign((size_t)std::max(1L, P.nf), 0);
    for (long f = 0; f < P.nf; f++) {
        if (P.group[f] < 0) continue;
        needc[(size_t)f] = 1;
        for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) needc[(size_t)P.Child[q]] = 1;
    }
    for (long kf = 0; kf < P.nf; kf++) {
        const long f = P.Post[kf];
        FrontSym &s = P.fs[f];
        s.foff = 0; s.coff = 0;
        if (P.group[f] >= 0) {
            s.foff = foff;
            foff += (long long)s.ld * s.fn;
        }
        if (needc[(size_t)f]) {
            const long cn = s.fn - s.fp, fm = s.fm_ub;
            const long cm = P.do_rank ? std::min(fm, cn) : std::min(std::max(fm - std::min(fm, (long)s.fp), 0L), cn);
            s.coff = coff;
            P.c_slot[(size_t)f] = (long long)cm * (cm + 1) / 2 + (long long)cm * (cn - cm);
            coff += P.c_slot[(size_t)f];
            coff = (coff + 1) & ~1LL;
        }
    }
    P.has_c.swap(needc);
    P.farena = std::max(1LL, foff); P.carena = std::max(1LL, coff);
}

// the arenas themselves: (re)allocated when their size changed (first factorization of a plan, or after a regrouping)
int ensure_arenas(stmmqr_plan &P)
{
    if (P.d_F.p && P.d_F.n == (size_t)P.farena && P.d_C.p && P.d_C.n == (size_t)P.carena &&
        (!P.recycle || (P.d_RH.p && P.d_RH.n == (size_t)P.rh_cap)))
        return 0;
    HIPCHK(hipStreamSynchronize(P.stream));
    if (P.d_F.n != (size_t)P.farena) P.d_F.release();
    if (P.d_C.n != (size_t)P.carena) P.d_C.release();
    size_t freeb = 0, totalb = 0;
    HIPCHK(hipMemGetInfo(&freeb, &totalb));
    const double need = 8.0 * ((P.d_F.p ? 0.0 : (double)P.farena) + (P.d_C.p ? 0.0 : (double)P.carena) +
                               ((P.recycle && P.d_RH.n != (size_t)P.rh_cap) ? (double)P.rh_cap : 0.0)) * 1.02;
    if (need > 0.95 * (double)freeb) return fail(STMMQR_ERR_OUT_OF_MEMORY, "front arena does not fit in free HBM");
    if (!P.d_F.p) LCHK(P.d_F.alloc((size_t)P.farena));
    if (!P.d_C.p) LCHK(P.d_C.alloc((size_t)P.carena));
    if (P.recycle && P.d_RH.n != (size_t)P.rh_cap) LCHK(P.d_RH.alloc((size_t)P.rh_cap));
    return 0;
}

// ---- timeline allocator (slab recycling) ----------------------------------------------------------------------------------
// Objects with a size and a lifetime [t0, t1] in steps; an address may be given to another object from step t1 + 1 on.  Objects
// are placed in order of their first step (larger first inside a step) at the lowest address that holds them (address-ordered
// first fit, free blocks coalesced).  Returns the peak address.  Everything is symbolic, so the offsets are part of the plan.
struct TlObj { long long size; int t0, t1; long long *out; };
long long timeline_first_fit(std::vector<TlObj> &objs, long long base, int nstep)
{
    std::vector<std::vector<int>> at0((size_t)nstep + 1), at1((size_t)nstep + 2);
    for (size_t i = 0; i < objs.size(); i++) {
        objs[i].size = (objs[i].size + 1) & ~1LL;
        if (objs[i].size <= 0) { *objs[i].out = base; continue; }
        at0[(size_t)objs[i].t0].push_back((int)i);
        at1[(size_t)objs[i].t1 + 1].push_back((int)i);
    }
    std::vector<std::pair<long long, long long>> freeb;            // (offset, size), sorted by offset, never adjacent
    long long top = base;
    auto release = [&](long long off, long long sz) {
        size_t i = (size_t)(std::lower_bound(freeb.begin(), freeb.end(), std::make_pair(off, 0LL)) - freeb.begin());
        freeb.insert(freeb.begin() + (long)i, {off, sz});
        if (i + 1 < freeb.size() && freeb[i].first + freeb[i].second == freeb[i + 1].first) {
            freeb[i].second += freeb[i + 1].second;
            freeb.erase(freeb.begin() + (long)i + 1);
        }
        if (i > 0 && freeb[i - 1].first + freeb[i - 1].second == freeb[i].first) {
            freeb[i - 1].second += freeb[i].second;
            freeb.erase(freeb.begin() + (long)i);
        }
        if (!freeb.empty() && freeb.back().first + freeb.back().second == top) {      // (no free block ever touches the top)
            top = freeb.back().first;
            freeb.pop_back();
        }
    };
    long long peak = base;
    for (int t = 0; t <= nstep; t++) {
        for (int i : at1[(size_t)t]) release(*objs[(size_t)i].out, objs[(size_t)i].size);
        if (t == nstep) break;
        std::vector<int> &now = at0[(size_t)t];
        std::stable_sort(now.begin(), now.end(), [&](int a, int b) { return objs[(size_t)a].size > objs[(size_t)b].size; });
        for (int i : now) {
            const long long sz = objs[(size_t)i].size;
            long long at = -1;
            for (auto it = freeb.begin(); it != freeb.end(); ++it)
                if (it->second >= sz) {
                    at = it->first;
                    if (it->second == sz) freeb.erase(it);
                    else { it->first += sz; it->second -= sz; }
                    break;
                }
            if (at < 0) { at = top; top = at + sz; }             // nothing fits: the arena grows
            *objs[(size_t)i].out = at;
            peak = std::max(peak, top);
        }
    }
    return peak;
}

// Front and contribution-block offsets of a plan that holds the whole tree in ONE group, from the step timeline of that group
// (P.f_t0 / f_t1 / c_t1, filled by build_schedule).  `kept` fronts sit at the bottom of the front arena for good.
// Returns {front arena, contribution arena} in doubles.
std::pair<long long, long long> timeline_offsets(stmmqr_plan &P, const std::vector<char> &kept, int nstep, bool apply)
{
    const long nf = P.nf;
    std::vector<long long> foff((size_t)std::max(1L, nf), 0), coff((size_t)std::max(1L, nf), 0);
    long long base = 0;
    for (long f = 0; f < nf; f++)
        if (kept[(size_t)f]) { foff[(size_t)f] = base; base += (long long)P.fs[f].ld * P.fs[f].fn; }
    std::vector<TlObj> fo, co;
    for (long f = 0; f < nf; f++) {
        const FrontSym &s = P.fs[f];
        if (!kept[(size_t)f]) fo.push_back({(long long)s.ld * s.fn, P.f_t0[f], P.f_t1[f], &foff[(size_t)f]});
        const long cn = s.fn - s.fp, fm = s.fm_ub;
        const long cm = P.do_rank ? std::min(fm, cn) : std::min(std::max(fm - std::min(fm, (long)s.fp), 0L), cn);
        const long long csz = (long long)cm * (cm + 1) / 2 + (long long)cm * (cn - cm);
        if (apply) P.c_slot[(size_t)f] = csz;
        co.push_back({csz, P.f_t1[f], std::max(P.f_t1[f], P.c_t1[f]), &coff[(size_t)f]});
    }
    const long long fpeak = timeline_first_fit(fo, base, nstep), cpeak = timeline_first_fit(co, 0, nstep);
    if (apply)
        for (long f = 0; f < nf; f++) { P.fs[f].foff = foff[(size_t)f]; P.fs[f].coff = coff[(size_t)f]; }
    return {std::max(1LL, fpeak), std::max(1LL, cpeak)};
}

void build_schedule(stmmqr_plan &P, std::vector<int> &tslot)
{
    P.sched_gen++;
    P.tune = getenv("STMMQR_TUNE") ? atoi(getenv("STMMQR_TUNE")) : 0;
    P.tall_min = g_opt.tall_min_rows;
    P.plan_algo = g_opt.panel_algo;
    P.ca_min = getenv("STMMQR_CA_MIN") ? atoi(getenv("STMMQR_CA_MIN")) : STM_CA_MIN_ROWS;
    const long nf = P.nf;
    int ngroups = 1;
    for (long f = 0; f < nf; f++) ngroups = std::max(ngroups, P.group[f] + 1);
    {
        // slab recycling: plans that hold the whole tree in one group (STMMQR_RECYCLE=0: every front keeps its slab, as sharded
        // plans do); a plan whose last factorization overflowed the R+H arena stays without it
        bool whole = (ngroups == 1) && nf > 0;
        for (long f = 0; f < nf && whole; f++) whole = (P.group[f] == 0) && !((size_t)f < P.shared.size() && P.shared[(size_t)f]);
        // (STMMQR_RECYCLE: 0 never, 2 always, unset / 1: when the slabs of all fronts exceed 256 MB -- below that the three extra
        //  launches per step cost more than the memory is worth: epb1 holds 70 MB either way and took 7.4 -> 7.7 ms)
        const char *ev = getenv("STMMQR_RECYCLE");
        const int mode = ev ? atoi(ev) : 1;
        long long slabs = 0;
        for (long f = 0; f < nf; f++) slabs += (long long)P.fs[f].ld * P.fs[f].fn;
        P.recycle = whole && !P.overflowed && (mode == 2 || (mode == 1 && slabs >= (32LL << 20)));
    }
    auto is_big = [&](int f) {
        const FrontSym &s = P.fs[f];
        return s.fn >= g_opt.big_front_cols && s.fm_ub >= 64;
    };
    // STMMQR_SCHED (experiments): 1 level-synchronous, 2 as soon as possible, 3 envelope rule; unset / 0: chosen per group
    const int sched_policy = getenv("STMMQR_SCHED") ? atoi(getenv("STMMQR_SCHED")) : 0;
    // Pair update (k_upd_w2 / k_upd_c2): a property of the front alone -- it changes the rounding of the front's
    // trailing updates, and results must not depend on the schedule.  Fronts whose update is bandwidth bound: many rows.
    const long pair_min = getenv("STMMQR_PAIR_MIN") ? atol(getenv("STMMQR_PAIR_MIN")) : STM_PAIR_MIN_ROWS;
    auto is_pair = [&](int f) {
        const FrontSym &s = P.fs[f];
        if ((size_t)f < P.shared.size() && P.shared[f]) return false;   // (the pair update has no column-block stride)
        return g_opt.pair_update && is_big(f) && s.fm_est >= pair_min && s.npanels >= 4;
    };
    P.sweep = (g_opt.pair_update == 4) ? 4 : 2;
    P.pair_front.assign(std::max(1L, nf), 0);
    for (long f = 0; f < nf; f++) P.pair_front[f] = (P.group[f] >= 0 && is_pair((int)f)) ? 1 : 0;
    P.ypoff.assign(std::max(1L, nf), -1);
    P.yp_doubles = 0;
    for (long f = 0; f < nf; f++)
        if (P.pair_front[f]) {
            P.ypoff[f] = P.yp_doubles;
            P.yp_doubles += (long long)((P.fs[f].fn + 31) / 32) * (P.sweep * STM_NB * 32);
        }
    tslot.assign(std::max(1L, nf), 0);
    P.glevels.assign(ngroups, std::vector<Level>());
    P.gsteps.assign(ngroups, std::vector<Step>());
    P.lists.clear();
    P.wlists.clear();
    P.tslots = 1;
    P.gp_slabs = 1;
    P.wp_doubles = 0;
    P.wp2_doubles = 0;
    for (int grp = 0; grp < ngroups; grp++) {
        // ---- tree levels (leaves = 0, counted inside the group): the order of the solves and of Q ----
        std::vector<int> level(nf, -1), start(nf, 0), end(nf, 0);
        int nlev = 0, nstep = 0;
        for (long kf = 0; kf < nf; kf++) {
            const long f = P.Post[kf];
            if (P.group[f] != grp) continue;
            int lv = 0, t0 = 0;
            for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) {
                const long ch = P.Child[q];
                if (P.group[ch] != grp) continue;              // (earlier phase or imported: already there)
                lv = std::max(lv, level[ch] + 1);
                t0 = std::max(t0, end[ch]);
            }
            level[f] = lv;
            nlev = std::max(nlev, lv + 1);
            start[f] = t0;
            end[f] = t0 + (is_big((int)f) ? P.fs[f].npanels : 1);
            nstep = std::max(nstep, end[f]);
        }
        // the level-synchronous schedule: every front of a tree level starts when the level below has finished
        std::vector<int> lstart(nf, 0), lend(nf, 0);
        int nstep_level = 0;
        {
            std::vector<int> lvl_end(nlev + 1, 0);
            for (int lv = 0; lv < nlev; lv++) {
                int e1 = lvl_end[lv];
                for (long kf = 0; kf < nf; kf++) {
                    const long f = P.Post[kf];
                    if (P.group[f] != grp || level[f] != lv) continue;
                    lstart[f] = lvl_end[lv];
                    lend[f] = lstart[f] + (is_big((int)f) ? P.fs[f].npanels : 1);
                    e1 = std::max(e1, lend[f]);
                }
                lvl_end[lv + 1] = e1;
            }
            nstep_level = lvl_end[nlev];
        }
        // Which order (STMMQR_SCHED unset): the envelope rule reaches the minimum number of steps (nstep here) but its steps
        // are ~10-15 % longer than those of the level-synchronous order (a launch lasts as long as its slowest front and
        // rows are only a proxy for that; measured) -- it is taken when it removes more than a fifth of the steps.
        const bool use_level = sched_policy == 1 || (sched_policy == 0 && 5L * nstep > 4L * nstep_level);
        if (use_level) {
            start = lstart; end = lend;
            nstep = nstep_level;
        }
        std::vector<std::vector<int>> byl(nlev);
        for (long kf = 0; kf < nf; kf++)
            if (P.group[P.Post[kf]] == grp) byl[level[P.Post[kf]]].push_back((int)P.Post[kf]);
        std::vector<Level> &LV = P.glevels[grp];
        LV.assign(nlev, Level());
        for (int lv = 0; lv < nlev; lv++) {
            Level &L = LV[lv];
            std::vector<int> small, big;
            for (int f : byl[lv]) (is_big(f) ? big : small).push_back(f);
            std::stable_sort(big.begin(), big.end(), [&](int a, int b) { return P.fs[a].npanels > P.fs[b].npanels; });
            L.all_off = (int)P.lists.size();
            L.n_small = (int)small.size(); L.n_big = (int)big.size(); L.n_all = L.n_small + L.n_big;
            P.lists.insert(P.lists.end(), small.begin(), small.end());
            P.lists.insert(P.lists.end(), big.begin(), big.end());
        }
        // ---- the step timeline ----
        // panel_at[t] = the (big front, panel) pairs of step t; small_at[t] = the small fronts factorized at step t.
        // The envelope rule: every step runs the fronts on the longest remaining path (counted in panels up to the root); any other
        // front that is ready or in flight rides along if its panel is no taller than theirs -- a launch lasts as long
        // as its tallest panel and is configured for it (LDS, column groups), so shorter panels are free while a taller
        // one would make the step of the critical fronts longer.  The minimum number of steps, the cheapest envelope.
        std::vector<std::vector<std::pair<int, int>>> panel_at;
        std::vector<std::vector<int>> small_at;
        if (use_level || sched_policy == 2) {
            panel_at.assign(nstep, {});
            small_at.assign(nstep, {});
            for (long kf = 0; kf < nf; kf++) {
                const int f = (int)P.Post[kf];
                if (P.group[f] != grp) continue;
                if (!is_big(f)) { small_at[start[f]].push_back(f); continue; }
                for (int q = 0; q < P.fs[f].npanels; q++) panel_at[start[f] + q].push_back({f, q});
            }
        } else {
            std::vector<int> parent_in(nf, -1), pend(nf, 0), tails(nf, 0);
            for (long f = 0; f < nf; f++) {
                if (P.group[f] != grp) continue;
                for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++)
                    if (P.group[P.Child[q]] == grp) { parent_in[P.Child[q]] = (int)f; pend[f]++; }
            }
            for (long kf = nf; kf-- > 0;) {
                const long f = P.Post[kf];
                if (P.group[f] != grp) continue;
                tails[f] = (is_big((int)f) ? P.fs[f].npanels : 1) + (parent_in[f] >= 0 ? tails[parent_in[f]] : 0);
            }
            std::vector<int> ready;
            for (long kf = 0; kf < nf; kf++)
                if (P.group[P.Post[kf]] == grp && pend[P.Post[kf]] == 0) ready.push_back((int)P.Post[kf]);
            const double ride = getenv("STMMQR_RIDE") ? atof(getenv("STMMQR_RIDE")) : 1.0;
            struct Fly { int f, p; };
            std::vector<Fly> fly;                                 // big fronts ready or in flight, with their next panel
            while (!ready.empty() || !fly.empty()) {
                std::vector<int> newly, smalls;
                for (int f : ready) {
                    if (is_big(f)) fly.push_back({f, 0});
                    else smalls.push_back(f);
                }
                panel_at.push_back({});
                small_at.push_back(smalls);
                auto finish = [&](int f) {
                    const int pf = parent_in[f];
                    if (pf >= 0 && --pend[pf] == 0) newly.push_back(pf);
                };
                if (!fly.empty()) {
                    auto rem = [&](const Fly &e) { return tails[e.f] - e.p; };
                    auto rows = [&](const Fly &e) { return stm_panel_rows_est(P.fs[e.f], e.p); };
                    int rl = 0, env = 0;
                    for (const Fly &e : fly) rl = std::max(rl, rem(e));
                    for (const Fly &e : fly)
                        if (rem(e) == rl) env = std::max(env, rows(e));
                    std::vector<Fly> keep;
                    for (Fly &e : fly) {
                        if (rem(e) == rl || rows(e) <= ride * env) {
                            panel_at.back().push_back({e.f, e.p});
                            if (++e.p >= P.fs[e.f].npanels) { finish(e.f); continue; }
                        }
                        keep.push_back(e);
                    }
                    fly.swap(keep);
                }
                for (int f : smalls) finish(f);
                ready.swap(newly);
            }
            nstep = (int)panel_at.size();
        }
        std::vector<std::vector<int>> starting(nstep), ending(nstep);
        std::vector<int> pan_now(nf, 0);                          // the panel a front in flight is at, per step
        for (int t = 0; t < nstep; t++) {
            for (int f : small_at[t]) starting[t].push_back(f);
            for (const auto &fp : panel_at[t]) {
                if (fp.second == 0) starting[t].push_back(fp.first);
                if (fp.second == P.fs[fp.first].npanels - 1) ending[t].push_back(fp.first);
            }
        }
        // The packing of finished fronts is batched: k_cpack runs at the last step before some front STARTS (only an
        // assembly reads a packed block) or at the end of the group -- one launch per tree level in the level-synchronous
        // order instead of one per step in which a front happens to end.  (T / Gram slots are released at the flush:
        // k_cpack's extra workgroup may still write the T of the front's last panel.)
        for (int t = 0, carry_from = -1; t < nstep; t++) {
            const bool flush = (t + 1 == nstep) || !starting[t + 1].empty();
            if (carry_from >= 0 && carry_from != t) {
                ending[t].insert(ending[t].begin(), ending[carry_from].begin(), ending[carry_from].end());
                ending[carry_from].clear();
            }
            carry_from = (flush || ending[t].empty()) ? -1 : t;
            if (!flush && !ending[t].empty()) carry_from = t;
        }
        std::vector<Step> &SV = P.gsteps[grp];
        SV.assign(nstep, Step());
        std::vector<int> active, freeslots;                   // big fronts in flight; released T / Gram slots
        int nslots = 0;
        for (int t = 0; t < nstep; t++) {
            Step &S = SV[t];
            std::vector<int> small, big;
            for (int f : starting[t]) (is_big(f) ? big : small).push_back(f);
            S.start_off = (int)P.lists.size();
            S.n_small = (int)small.size(); S.n_start = (int)(small.size() + big.size());
            P.lists.insert(P.lists.end(), small.begin(), small.end());
            P.lists.insert(P.lists.end(), big.begin(), big.end());
            S.asm_parts_off = (int)P.lists.size();
            long maxfm_small = 0;
            for (int i = 0; i < S.n_start; i++) {
                const FrontSym &s = P.fs[P.lists[S.start_off + i]];
                const long work = (long)s.fm_ub * s.fn;
                const int parts = (int)std::min(512L, std::max(1L, (work + 16383) / 16384));     // (two workgroups per CU on the top fronts)
                P.lists.push_back(parts);
                S.asm_maxparts = std::max(S.asm_maxparts, parts);
                if (i < S.n_small) maxfm_small = std::max(maxfm_small, (long)s.fm_ub);
            }
            // (rows padded as dev_panel pads them, so that a whole panel fits whenever the cap allows: the sub-panel width
            //  of a front -- and with it the rounding -- must not depend on which other fronts share its step)
            S.lds_small = (int)std::min((long)LDS_CAP_SMALL, (((maxfm_small + 63) & ~63L) | 1) * STM_NB + 64);
            // T / Gram-partial slots: a slot is held from the step a front starts to the step it ends
            for (int f : big) {
                if (!freeslots.empty()) { tslot[f] = freeslots.back(); freeslots.pop_back(); }
                else tslot[f] = nslots++;
            }
            active.clear();
            for (const auto &fp : panel_at[t]) { active.push_back(fp.first); pan_now[fp.first] = fp.second; }
            // heaviest update first (the order inside a launch does not change any result); classes: see Step
            auto work_at = [&](int f) { return (long)stm_upd_ncb(P.fs[f], pan_now[f]) * stm_upd_nsl(P.fs[f]); };
            auto cls = [&](int f) { return !is_pair(f) ? 0 : 1 + pan_now[f] % P.sweep; };
            std::stable_sort(active.begin(), active.end(), [&](int a, int b) {
                return cls(a) != cls(b) ? cls(a) < cls(b) : work_at(a) > work_at(b);
            });
            S.act_off = (int)P.lists.size();
            S.n_act = (int)active.size();
            P.lists.insert(P.lists.end(), active.begin(), active.end());
            S.plist_off = (int)P.lists.size();
            for (int f : active) P.lists.push_back(pan_now[f]);
            S.wp_off = (int)P.wlists.size();
            long long wp = 0, ncbsum = 0;
            long maxfm_big = 0;
            for (int f : active) {
                const FrontSym &s = P.fs[f];
                const int p = pan_now[f];
                const int ncb = stm_upd_ncb(s, p), nsl = stm_upd_nsl(s);
                P.wlists.push_back(wp);
                // (+1: Gram block.  Pair-update fronts: two blocks per partial, at most stm_pair_slots partials per column block in the
                //  sweep of an odd panel -- a workgroup takes up to four slabs --, and the panel-by-panel updates of the next panels'
                //  columns use the first 2 + 1 column blocks with the full slab count)
                //  (quad update: four blocks per partial, three Gram blocks, 4 + 1 column blocks in the panel-by-panel updates)
                if (is_pair(f))
                    wp += std::max((long long)(ncb + P.sweep - 1) * (P.sweep == 4 ? stm_quad_slots(nsl, P.tune) : stm_pair_slots(nsl, P.tune)) *
                                       (P.sweep * STM_NB * 32),
                                   (P.sweep + 1LL) * nsl * (STM_NB * 32));
                else
                    wp += (long long)(ncb + 1) * nsl * (STM_NB * 32);
                const int k = cls(f);
                if (k == 0) {
                    S.n_norm++;
                    ncbsum += ncb;
                    S.maxcb = std::max(S.maxcb, ncb);
                    S.maxsl = std::max(S.maxsl, nsl);
                } else {
                    S.n_pk[k - 1]++;
                    S.maxsl_pk[k - 1] = std::max(S.maxsl_pk[k - 1], nsl);
                    if (k == P.sweep) S.maxcbp_po = std::max(S.maxcbp_po, ncb - 1);
                }
                S.nsub = std::max(S.nsub, stm_tall_launches(s, p, P.tall_min));
                S.nca = std::max(S.nca, stm_ca_slabs(s));
                (stm_use_ca(s, p, g_opt.panel_algo, P.ca_min) ? S.nca_use : S.npipe_use)++;
                P.gp_slabs = std::max(P.gp_slabs, stm_ca_slabs(s));
                maxfm_big = std::max(maxfm_big, (long)s.fm_ub);
                if (!stm_tall_panel(s, p, P.tall_min) && !stm_use_ca(s, p, g_opt.panel_algo, P.ca_min))
                    S.lds_plan = std::max(S.lds_plan, stm_front_lds(s));
                // a short panel of the pipeline is taken by one workgroup with the panel's image in LDS (dev_wave_panel)
                if (stm_tall_panel(s, p, P.tall_min) && !stm_use_ca(s, p, g_opt.panel_algo, P.ca_min))
                    S.lds_plan = std::max(S.lds_plan, STM_NB * STM_WP_ROWS);
            }
            S.lds_big = (int)std::min((long)LDS_CAP_DOUBLES, (((maxfm_big + 63) & ~63L) | 1) * STM_NB + 64);
            // row-parallel update when it pays: >= 3 slabs, or so many column blocks in the launch that the one-workgroup
            // form's redundant T (every column-block workgroup builds it) costs throughput (T is built once per front
            // by k_upd_w).  Either form gives the same bits.
            S.split = (S.maxsl >= 3 || ncbsum >= 512) ? 1 : 0;
            P.wp_doubles = std::max(P.wp_doubles, wp);
            if (S.n_sweep() == 0) P.wp2_doubles = std::max(P.wp2_doubles, wp);
            // fronts at their last panel: packed at the end of the step, slot released for the next
            S.cpk_off = (int)P.lists.size();
            S.n_cpk = (int)ending[t].size();
            P.lists.insert(P.lists.end(), ending[t].begin(), ending[t].end());
            S.cpk_parts_off = (int)P.lists.size();
            for (int f : ending[t]) {
                const FrontSym &s = P.fs[f];
                const long cn = s.fn - s.fp;
                const long work = cn * std::min((long)s.fm_ub, cn);
                const int parts = (int)std::min(512L, std::max(1L, (work + 16383) / 16384));
                P.lists.push_back(parts);
                S.cpk_maxparts = std::max(S.cpk_maxparts, parts);
                freeslots.push_back(tslot[f]);
            }
        }
        P.tslots = std::max(P.tslots, nslots);
        // ---- slab recycling: lifetimes of the slabs and contribution blocks on this timeline, the fronts that keep their slab,
        // the offsets, and per step the fronts whose packed R+H block is staged at its end ----
        if (P.recycle && grp == 0) {
            P.f_t0.assign((size_t)std::max(1L, nf), 0); P.f_t1.assign((size_t)std::max(1L, nf), 0); P.c_t1.assign((size_t)std::max(1L, nf), 0);
            for (int t = 0; t < nstep; t++) {
                for (int f : starting[t]) { P.f_t0[(size_t)f] = t; if (!is_big(f)) P.f_t1[(size_t)f] = t; }
                for (int f : ending[t]) P.f_t1[(size_t)f] = t;
            }
            for (long f = 0; f < nf; f++) {
                const int par = P.fs[f].parent;
                P.c_t1[(size_t)f] = (par >= 0) ? P.f_t0[(size_t)par] : P.f_t1[(size_t)f];
            }
            // which fronts keep their slab: none, or the 1-3 largest -- whatever makes fronts + contribution blocks + R+H arena
            // smallest (the arena holds min(maxstack, all recycled slabs) doubles: the reference's bound for all of R+H)
            std::vector<int> bysize((size_t)nf);
            for (long f = 0; f < nf; f++) bysize[(size_t)f] = (int)f;
            std::stable_sort(bysize.begin(), bysize.end(), [&](int a, int b) {
                return (long long)P.fs[a].ld * P.fs[a].fn > (long long)P.fs[b].ld * P.fs[b].fn; });
            long long best = -1;
            int bestk = 0;
            for (int k = 0; k <= std::min(3L, nf); k++) {
                std::vector<char> kp((size_t)std::max(1L, nf), 0);
                long long rec = 0;
                for (int q = 0; q < k; q++) kp[(size_t)bysize[(size_t)q]] = 1;
                for (long f = 0; f < nf; f++) if (!kp[(size_t)f]) rec += (long long)P.fs[f].ld * P.fs[f].fn;
                const auto pk = timeline_offsets(P, kp, nstep, false);
                const long long cap = (P.maxstack > 0) ? std::min((long long)P.maxstack, rec) : rec;
                const long long tot = pk.first + pk.second + cap;
                if (best < 0 || tot < best - best / 50) { best = tot; bestk = k; }      // (a kept front must buy at least 2 %)
            }
            P.kept.assign((size_t)std::max(1L, nf), 0);
            long long rec = 0;
            for (int q = 0; q < bestk; q++) P.kept[(size_t)bysize[(size_t)q]] = 1;
            for (long f = 0; f < nf; f++) if (!P.kept[(size_t)f]) rec += (long long)P.fs[f].ld * P.fs[f].fn;
            P.rh_cap = std::max(1LL, (P.maxstack > 0) ? std::min((long long)P.maxstack, rec) : rec);
            // (the estimate + 12.5 % where that is less: dead columns move rows into later fronts and can make the factors
            //  larger than the full-rank pattern says; an arena that overflows is regrown to the hard bound and the factorization
            //  repeated once -- stats.retries says so)
            if (!P.rh_grow && P.rh_est_total > 0) {
                // (STMMQR_RH_EST_SCALE: tests shrink the estimate to drive the overflow path)
                const double sc = getenv("STMMQR_RH_EST_SCALE") ? atof(getenv("STMMQR_RH_EST_SCALE")) : 1.0;
                const long long est = (long long)((double)P.rh_est_total * sc);
                P.rh_cap = std::max(1LL, std::min(P.rh_cap, est + est / 8 + 4096));
            }
            const auto pk = timeline_offsets(P, P.kept, nstep, true);
            P.farena = pk.first; P.carena = pk.second;
            P.has_c.assign((size_t)std::max(1L, nf), 1);
            for (int t = 0; t < nstep; t++) {
                Step &S = SV[t];
                std::vector<int> rhp;
                for (int f : starting[t]) if (!is_big(f) && !P.kept[(size_t)f]) rhp.push_back(f);
                for (int f : ending[t]) if (!P.kept[(size_t)f]) rhp.push_back(f);
                S.rhp_off = (int)P.lists.size();
                S.n_rhp = (int)rhp.size();
                P.lists.insert(P.lists.end(), rhp.begin(), rhp.end());
                S.rhp_parts_off = (int)P.lists.size();
                for (int f : rhp) {
                    const int parts = std::min(256, std::max(1, P.fs[f].fn / 16));
                    P.lists.push_back(parts);
                    S.rhp_maxparts = std::max(S.rhp_maxparts, parts);
                }
            }
            // scratch of the resident-factor operations: every tree level in front form, one level at a time
            P.fs_scr = P.fs;
            P.scr_doubles = 1;
            for (size_t l = 0; l < LV.size(); l++) {
                long long o = 0;
                for (int q = 0; q < LV[l].n_all; q++) {
                    const int f = P.lists[(size_t)(LV[l].all_off + q)];
                    if (P.kept[(size_t)f]) continue;
                    P.fs_scr[(size_t)f].foff = o;
                    o += (long long)P.fs[f].ld * P.fs[f].fn;
                }
                P.scr_doubles = std::max(P.scr_doubles, o);
            }
        }
    }
    // fronts factorized on this device, in Post order, + their R+H copy parts; then ALL fronts in Post order
    P.own_off = (int)P.lists.size();
    P.n_own = 0;
    for (long kf = 0; kf < nf; kf++)
        if (P.group[P.Post[kf]] >= 0) { P.lists.push_back((int)P.Post[kf]); P.n_own++; }
    P.rh_parts_off = (int)P.lists.size();
    P.rh_maxparts = 1;
    for (long kf = 0; kf < nf; kf++) {
        if (P.group[P.Post[kf]] < 0) continue;
        const FrontSym &s = P.fs[P.Post[kf]];
        int parts = std::min(256, std::max(1, s.fn / 16));     // (a wave per column; 64 left the 17.6 GB copy of a 50 000-column front at 0.5 TB/s)
        P.lists.push_back(parts);
        P.rh_maxparts = std::max(P.rh_maxparts, parts);
    }
    P.post_off = (int)P.lists.size();
    for (long kf = 0; kf < nf; kf++) P.lists.push_back((int)P.Post[kf]);
    if (P.lists.empty()) P.lists.push_back(0);
    if (P.wlists.empty()) P.wlists.push_back(0);
    P.h_tslot = tslot;
}

// device side of the slab recycling of the current schedule (after build_schedule): kept flags, scratch layout, bump words
int upload_recycle(stmmqr_plan &P)
{
    if (!P.d_rhtop.p) LCHK(P.d_rhtop.alloc(2));
    if (!P.d_fin.p || P.d_fin.n < (size_t)std::max(1L, P.nf)) LCHK(P.d_fin.alloc((size_t)std::max(1L, P.nf)));
    HIPCHK(hipMemsetAsync(P.d_rhtop.p, 0, 2 * sizeof(long long), P.stream));
    if (!P.recycle) return 0;
    LCHK(P.d_kept.upload(P.kept, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    P.d_scr.release();                                        // (allocated by the first resident-factor operation: ensure_scratch)
    P.d_RH.release();                                         // (the arena follows with the front arenas: ensure_arenas)
    return 0;
}

// ------------------------------------------------------------------------------------------------
// planner: everything that depends only on the symbolic analysis
// ------------------------------------------------------------------------------------------------
int build_plan(stmmqr_plan &P, const stmmqr_symbolic_view &v)
{
    P.m = v.m; P.n = v.n; P.anz = v.anz; P.nf = v.nf; P.maxfn = v.maxfn; P.rjsize = v.rjsize;
    P.hisize = v.hisize; P.do_rank = v.do_rank_detection ? 1 : 0;
    P.maxstack = v.maxstack > 0 ? v.maxstack : 0;
    const long m = v.m, n = v.n, nf = v.nf;
    if (m < 0 || n < 0 || nf < 0) return fail(STMMQR_ERR_INVALID, "negative dimension");
    if (v.anz >= (1L << 31) - 1 || v.rjsize >= (1L << 31) - 1 || v.hisize >= (1L << 31) - 1 || m >= (1L << 30) ||
        n >= (1L << 30))
        return fail(STMMQR_ERR_TOO_LARGE, "problem exceeds the 32-bit device index range");
    auto cp = [](std::vector<long> &dst, const stm_long *src, long cnt) {
        dst.assign(src, src + (cnt > 0 ? cnt : 0));
    };
    cp(P.Sp, v.Sp, m + 1); cp(P.Sj, v.Sj, v.anz); cp(P.PLinv, v.PLinv, m); cp(P.Sleft, v.Sleft, n + 2);
    cp(P.Child, v.Child, nf + 1); cp(P.Childp, v.Childp, nf + 2); cp(P.Super, v.Super, nf + 1);
    cp(P.Rp, v.Rp, nf + 1); cp(P.Rj, v.Rj, v.rjsize); cp(P.Post, v.Post, nf); cp(P.Hip, v.Hip, nf + 1);
    P.has_qfill = v.Qfill != nullptr;
    if (P.has_qfill) cp(P.Qfill, v.Qfill, n);

    // ---- per-front symbolic sizes -----------------------------------------------------------
    std::vector<long> parent(nf, -1);
    for (long f = 0; f < nf; f++)
        for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) parent[P.Child[q]] = f;

    // upper bound on the rows of every front: taken from the analysis when given (QRsym->Fm, worst case when
    // rank detection is on: SparseQR_analyze.c:461-471), otherwise recomputed with the same recurrence
    P.Fm.assign(nf, 0);
    if (v.Fm) {
        for (long f = 0; f < nf; f++) P.Fm[f] = v.Fm[f];
    } else {
        std::vector<long> cmub(nf, 0);
        for (long kf = 0; kf < nf; kf++) {
            const long f = P.Post[kf];
            const long fp = P.Super[f + 1] - P.Super[f], fn = P.Rp[f + 1] - P.Rp[f];
            long fm = P.Sleft[P.Super[f + 1]] - P.Sleft[P.Super[f]];
            for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) fm += cmub[P.Child[q]];
            P.Fm[f] = fm;
            const long cn = fn - fp;
            cmub[f] = P.do_rank ? std::min(fm, cn) : std::min(std::max(fm - std::min(fm, fp), 0L), cn);
        }
    }

    // the row count every front will have if no pivot column dies (exact for full-rank input): used only to plan the
    // number of panel launches (stm_tall_panel); the kernels cope with any actual row count
    std::vector<long> fmest(nf, 0), cmest(nf, 0);
    for (long kf = 0; kf < nf; kf++) {
        const long f = P.Post[kf];
        const long fp = P.Super[f + 1] - P.Super[f], fn = P.Rp[f + 1] - P.Rp[f];
        long fe = P.Sleft[P.Super[f + 1]] - P.Sleft[P.Super[f]];
        for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) fe += cmest[P.Child[q]];
        fmest[f] = std::min(fe, P.Fm[f]);
        cmest[f] = std::min(std::max(fe - std::min(fe, fp), 0L), fn - fp);
    }

    P.fs.assign(nf, FrontSym());
    long long foff = 0, coff = 0, tpan_total = 0;
    for (long kf = 0; kf < nf; kf++) {
        const long f = P.Post[kf];
        FrontSym &s = P.fs[f];
        const long fp = P.Super[f + 1] - P.Super[f], fn = P.Rp[f + 1] - P.Rp[f];
        const long fm = P.Fm[f];
        // (front-local offsets are 64-bit on the device: row + column * ld; rows and columns themselves are int32)
        if (fm * fn >= (1L << 36)) return fail(STMMQR_ERR_TOO_LARGE, "a single front exceeds 2^36 entries");
        s.fn = (int)fn; s.fp = (int)fp; s.col1 = (int)P.Super[f]; s.rp = (int)P.Rp[f]; s.hip = (int)P.Hip[f];
        s.child0 = (int)P.Childp[f]; s.child1 = (int)P.Childp[f + 1];
        s.srow0 = (int)P.Sleft[P.Super[f]]; s.srow1 = (int)P.Sleft[P.Super[f + 1]];
        s.fm_ub = (int)fm;
        s.fm_est = (int)fmest[f];
        s.ld = (int)std::max(2L, (fm + 1) & ~1L);
        s.npanels = (int)((fn + STM_NB - 1) / STM_NB);
        s.tpan = (int)tpan_total;
        tpan_total += s.npanels;
        {
            // Q-apply: fronts with this many entries or more are split over workgroups (k_qbig_*); STMMQR_QBIG_MIN
            // overrides the threshold (tests send small fronts through that path).  2 M entries until round 4; with the grouped launches
            // (k_qbig_step4) smaller fronts pay too, at 256 KB of T4 per four panels -- default workload, threshold: Q'b / solve ms,
            // GB of T4: 2 M 13.1 / 20.6, 0.31; 1 M 12.5 / 20.0, 0.39; 256 K 11.9 / 19.5, 0.61; 128 K 11.6 / 19.2, 0.81.  1 M.
            const long qbig_min = getenv("STMMQR_QBIG_MIN") ? atol(getenv("STMMQR_QBIG_MIN")) : (1L << 20);
            s.qbig = (fm * fn >= qbig_min && fn >= 1) ? 1 : 0;
        }
        s.parent = (int)parent[f];
        s.foff = 0; s.coff = 0;                                    // (assign_arenas, once the groups are known)
    }
    (void)foff; (void)coff;
    P.tpanels = tpan_total;

    // ---- relative indices (value independent): child column -> parent column, S entry -> front column ----
    std::vector<int> Rjrel(std::max(1L, v.rjsize), 0), Sjrel(std::max(1L, v.anz), 0), Sj0(std::max(1L, m), -1);
    {
        std::vector<int> Fmap(std::max(1L, n), -1);
        for (long f = 0; f < nf; f++) {
            const long p1 = P.Rp[f], fn = P.Rp[f + 1] - p1;
            for (long j = 0; j < fn; j++) Fmap[P.Rj[p1 + j]] = (int)j;
            for (long r = P.fs[f].srow0; r < P.fs[f].srow1; r++)
                for (long p = P.Sp[r]; p < P.Sp[r + 1]; p++) Sjrel[p] = Fmap[P.Sj[p]];
            for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) {
                const long c = P.Child[q];
                const long fpc = P.Super[c + 1] - P.Super[c], pc = P.Rp[c] + fpc, cn = P.Rp[c + 1] - pc;
                for (long cj = 0; cj < cn; cj++) Rjrel[pc + cj] = Fmap[P.Rj[pc + cj]];
                P.bytes_assemble_idx += 4.0 * (double)(2 * cn);
            }
        }
        for (long r = 0; r < m; r++)
            if (P.Sp[r + 1] > P.Sp[r]) Sj0[r] = (int)P.Sj[P.Sp[r]];
        P.bytes_assemble_idx += 4.0 * (double)v.anz;
    }
    // ---- size of every packed R+H block if no pivot column dies (exact for full-rank input): the symbolic staircase of the front
    // (qr_fsize: rows of S by leftmost column + the children's contribution rows, whose leftmost columns are the columns of their
    // C) run through qr_front's row bookkeeping (:1434-1609) and qr_rhpack's column lengths (:1691-1784).  Sizes the R+H arena of
    // the slab recycling (with a margin; QRsym->maxstack is the hard bound the arena falls back to) ----
    P.rh_est_total = 0;
    {
        std::vector<long> stair;
        for (long kf = 0; kf < nf; kf++) {
            const long f = P.Post[kf];
            const long fp = P.fs[f].fp, fn = P.fs[f].fn, fm = fmest[f];
            stair.assign((size_t)fn + 1, 0);
            for (long j = 0; j < fp; j++) stair[(size_t)j] = P.Sleft[P.Super[f] + j + 1] - P.Sleft[P.Super[f] + j];
            for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++) {
                const long c = P.Child[q];
                const long fpc = P.Super[c + 1] - P.Super[c], pc = P.Rp[c] + fpc;
                for (long i = 0; i < cmest[c]; i++) stair[(size_t)Rjrel[(size_t)(pc + i)]]++;
            }
            long run = 0;
            for (long j = 0; j < fn; j++) { run += stair[(size_t)j]; stair[(size_t)j] = run; }       // rows with leftmost column <= j
            long g = 0, rm = 0;
            long long sz = 0;
            for (long k = 0; k < fn; k++) {
                long t;
                if (g >= fm) t = (k < fp) ? 0 : fm;                       // rows ran out
                else { t = std::min(fm, std::max(g + 1, stair[(size_t)k])); g++; }
                if (k < fp) { if (t > 0) rm++; sz += (t > 0) ? t : rm; }
                else { const long h = std::min(rm + (k - fp) + 1, fm); sz += rm + std::max(t - h, 0L); }
            }
            P.rh_est_total += sz;
        }
    }

    // ---- level schedule: one group holding every front (multi-GPU callers regroup with set_groups) ----
    P.group.assign(nf, 0);
    assign_arenas(P);
    std::vector<int> tslot;
    build_schedule(P, tslot);

    // ---- device memory ----------------------------------------------------------------------------
    size_t freeb = 0, totalb = 0;
    HIPCHK(hipMemGetInfo(&freeb, &totalb));
    // (the front and contribution-block arenas are allocated by the first factorization, ensure_arenas: a plan that is
    //  regrouped for one rank of a sharded run never holds the whole tree's fronts)
    const double need = 8.0 * 1024.0 * (double)P.tpanels * 1.05 + 64.0 * (double)(v.rjsize + v.anz);
    if (need > 0.92 * (double)freeb)
        return fail(STMMQR_ERR_OUT_OF_MEMORY, "the plan's index arrays do not fit in free HBM");
    hipStream_t st = P.stream;
    auto up32 = [&](DevBuf<int> &d, const std::vector<long> &h) {
        std::vector<int> t(h.begin(), h.end());
        return d.upload(t, st);
    };
    LCHK(P.d_fs.upload(P.fs, st));
    LCHK(P.d_fnum.alloc(std::max(1L, nf)));
    HIPCHK(hipMemsetAsync(P.d_fnum.p, 0, std::max(1L, nf) * sizeof(FrontNum), st));
    LCHK(P.d_T.alloc((size_t)STM_PD_RING * P.tslots * STM_NB * STM_NB));
    LCHK(P.d_Gp.alloc((size_t)P.tslots * (P.gp_slabs + 1) * STM_NB * STM_NB));
    LCHK(P.d_Tall.alloc((size_t)std::max(1LL, P.tpanels) * STM_NB * STM_NB));
    LCHK(P.d_Wp.alloc((size_t)P.wp_doubles));
    LCHK(P.d_Ypend.alloc((size_t)std::max(1LL, P.yp_doubles)));
    LCHK(P.d_ypoff.upload(P.ypoff, st));
    LCHK(P.d_Wp2.alloc((size_t)std::max(1LL, P.wp2_doubles)));
    P.wcnt_n = (size_t)(P.wp_doubles / (STM_NB * 32) + 1);
    LCHK(P.d_wcnt.alloc(P.wcnt_n));
    LCHK(P.d_wcnt2.alloc(P.wcnt_n));
    LCHK(P.d_wflag.alloc(P.wcnt_n));
    if (!P.d_abort.p) LCHK(P.d_abort.alloc(2));
    LCHK(P.d_wflag2.alloc(P.wcnt_n));
    HIPCHK(hipMemset(P.d_wcnt.p, 0, P.wcnt_n * sizeof(int)));
    HIPCHK(hipMemset(P.d_wcnt2.p, 0, P.wcnt_n * sizeof(int)));
    LCHK(P.d_tslot.upload(tslot, st));
    LCHK(P.d_Sx.alloc((size_t)v.anz));
    LCHK(P.d_Ax.alloc((size_t)v.anz));
    LCHK(P.d_smap.alloc((size_t)v.anz));
    LCHK(up32(P.d_Sp, P.Sp));
    LCHK(P.d_Sjrel.upload(Sjrel, st));
    LCHK(P.d_Sj0.upload(Sj0, st));
    LCHK(up32(P.d_Sleft, P.Sleft));
    LCHK(up32(P.d_Child, P.Child));
    LCHK(P.d_Rjrel.upload(Rjrel, st));
    LCHK(P.d_Stair.alloc((size_t)v.rjsize));
    LCHK(P.d_Tau.alloc((size_t)v.rjsize));
    LCHK(P.d_Hii.alloc((size_t)v.hisize));
    LCHK(P.d_Cmap.alloc((size_t)v.rjsize));
    LCHK(P.d_Cursor.alloc((size_t)v.rjsize));
    LCHK(P.d_Rhoff.alloc((size_t)v.rjsize));
    LCHK(P.d_Rboff.alloc((size_t)std::max(1L, nf)));
    LCHK(P.d_total.alloc(1));
    LCHK(P.d_dbg.alloc(2048));                                // ([64] counters + the timeline of a STAMPS build: 16 + 64 b + idx)
    LCHK(P.d_amax.alloc(1));
    LCHK(P.d_sig.alloc(2));
    HIPCHK(hipMemsetAsync(P.d_dbg.p, 0, 2048 * sizeof(unsigned long long), st));
    LCHK(P.d_Rdead.alloc((size_t)std::max(1L, n)));
    LCHK(P.d_lists.upload(P.lists, st));
    LCHK(P.d_wlists.upload(P.wlists, st));
    HIPCHK(hipStreamSynchronize(st));
    LCHK(upload_recycle(P));
    return 0;
}

// qr_stranspose2 as a symbolic map: smap[s] = p such that Sx[s] = Ax[p]  (SparseQR_factorize.c:755-785)
int set_pattern(stmmqr_plan &P, const stm_long *Ap, const stm_long *Ai)
{
    const long m = P.m, n = P.n;
    if (!Ap || !Ai) return fail(STMMQR_ERR_INVALID, "Ap/Ai are required");
    if (Ap[n] != P.anz) return fail(STMMQR_ERR_INVALID, "nnz(A) differs from the symbolic analysis");
    std::vector<long> W(P.Sp.begin(), P.Sp.begin() + m);
    std::vector<int> smap(std::max(1L, P.anz), 0);
    for (long col = 0; col < n; col++) {
        const long j = P.has_qfill ? P.Qfill[col] : col;
        for (long p = Ap[j]; p < Ap[j + 1]; p++) {
            const long i = Ai[p];
            if (i < 0 || i >= m) return fail(STMMQR_ERR_INVALID, "row index out of range");
            smap[W[P.PLinv[i]]++] = (int)p;
        }
    }
    if (P.anz > 0)
        HIPCHK(hipMemcpyAsync(P.d_smap.p, smap.data(), (size_t)P.anz * sizeof(int), hipMemcpyHostToDevice, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    P.pattern_set = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// the level-batched schedule (device resident inputs -> device resident factors)
// ------------------------------------------------------------------------------------------------
// State every factorization starts from (issued by stmmqr_factorize_begin, BEFORE any stmmqr_plan_import_front of the
// phased interface: an imported front's fm / rank / cm must survive until its parent assembles it).
int reset_factorization(stmmqr_plan &P)
{
    hipStream_t st = P.stream;
    LCHK(ensure_arenas(P));
    // (with slab recycling every front's slab is zeroed when the front starts: k_zero_slabs in prep)
    if (!P.recycle) HIPCHK(hipMemsetAsync(P.d_F.p, 0, (size_t)P.farena * sizeof(double), st));
    HIPCHK(hipMemsetAsync(P.d_rhtop.p, 0, 2 * sizeof(long long), st));
    HIPCHK(hipMemsetAsync(P.d_Rdead.p, 0, (size_t)std::max(1L, P.n), st));
    HIPCHK(hipMemsetAsync(P.d_fnum.p, 0, (size_t)std::max(1L, P.nf) * sizeof(FrontNum), st));
    // (tickets are back at zero after every launch unless a wait ran out; the flags carry step numbers)
    HIPCHK(hipMemsetAsync(P.d_wcnt.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_wcnt2.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_wflag.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_wflag2.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_abort.p, 0, 2 * sizeof(int), st));
    LCHK(stm_launch_sigma(P.d_Ax.p, (int)P.anz, P.d_amax.p, P.d_sig.p, st));
    LCHK(stm_launch_gather_sx(P.d_Ax.p, P.d_smap.p, P.d_Sx.p, (int)P.anz, st));
    P.stats.nlaunch += 6;
    return 0;
}

// Recovery of ONE group of the phased interface after a bounded panel wait ran out in it: everything its fronts wrote is
// put back to the state reset_factorization left (fronts of other groups and imported fronts are not touched).
int reset_group(stmmqr_plan &P, int grp)
{
    hipStream_t st = P.stream;
    const FrontNum zero = FrontNum();
    for (long f = 0; f < P.nf; f++) {
        if (P.group[f] != grp) continue;
        const FrontSym &s = P.fs[f];
        HIPCHK(hipMemsetAsync(P.d_F.p + s.foff, 0, (size_t)s.ld * (size_t)s.fn * sizeof(double), st));
        HIPCHK(hipMemcpyAsync(P.d_fnum.p + f, &zero, sizeof zero, hipMemcpyHostToDevice, st));
        if (s.fp > 0) HIPCHK(hipMemsetAsync(P.d_Rdead.p + s.col1, 0, (size_t)s.fp, st));
    }
    HIPCHK(hipMemsetAsync(P.d_wcnt.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_wcnt2.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_wflag.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_wflag2.p, 0, P.wcnt_n * sizeof(int), st));
    HIPCHK(hipMemsetAsync(P.d_abort.p, 0, 2 * sizeof(int), st));
    // slab recycling: a recycling plan holds the whole tree in this one group, and the aborted attempt has staged packed blocks behind
    // the arena's bump pointer; the rerun stages every block again, so the pointer (and the overflow word) go back to zero -- otherwise
    // the second set lands behind the first and overflows an arena that holds the estimate + 12.5 %
    if (P.recycle) HIPCHK(hipMemsetAsync(P.d_rhtop.p, 0, 2 * sizeof(long long), st));
    HIPCHK(hipStreamSynchronize(st));                       // (`zero` lives on this stack frame)
    return 0;
}

// One piece of one timeline step (stmmqr_factorize_step): what = STMMQR_STEP_* bits; the update takes the column blocks
// cb_first, cb_first + cb_stride, ... (at most cb_count of them when cb_count >= 0) of the step's fronts.
struct StepReq { int step, what, cb_first, cb_stride, cb_count; };

// Events of the look-ahead schedule order two streams of ONE device: no timing, and no system-scope fence when they are recorded
// (the host never inspects them; the kernels' own agent-scope release / acquire at their boundaries is what the other stream needs).
// STMMQR_LA_SYSFENCE=1 brings the default (system-scope) events back.
static unsigned la_event_flags()
{
    const bool sysfence = getenv("STMMQR_LA_SYSFENCE") && atoi(getenv("STMMQR_LA_SYSFENCE")) != 0;
    return hipEventDisableTiming | (sysfence ? 0u : (unsigned)hipEventDisableSystemFence);
}

int run_schedule(stmmqr_plan &P, bool detail, int grp, const StepReq *req = nullptr)
{
    hipStream_t st = P.stream;
    DevCtx c = P.ctx();
    if (req) c.cbskip = std::max(1, req->cb_stride) - 1;
    const int *L0 = P.d_lists.p;
    long nlaunch = 0;
    // detail timing: bracket each category with an event pair from the pool; no host synchronisation here
    enum { CAT_ASM = 0, CAT_SMALL = 1, CAT_PANEL = 2, CAT_UPD = 3, CAT_CPK = 4 };
    int cur_step = 0;
    auto timed = [&](int cat, auto &&fn) -> int {
        if (!detail) return fn();
        if (P.evused == P.evpairs.size()) {
            stmmqr_plan::EvPair q = {nullptr, nullptr, 0, 0};
            HIPCHK(hipEventCreate(&q.a));
            HIPCHK(hipEventCreate(&q.b));
            P.evpairs.push_back(q);
        }
        stmmqr_plan::EvPair &q = P.evpairs[P.evused++];
        q.cat = cat;
        q.step = cur_step;
        HIPCHK(hipEventRecord(q.a, st));
        int e = fn();
        if (e) return e;
        HIPCHK(hipEventRecord(q.b, st));
        return 0;
    };
    const int t_asm = CAT_ASM, t_front = CAT_SMALL, t_panel = CAT_PANEL, t_upd = CAT_UPD, t_cpk = CAT_CPK;

    if (grp < 0 || grp >= (int)P.gsteps.size()) return fail(STMMQR_ERR_INVALID, "no such front group");
    const std::vector<Step> &SV = P.gsteps[grp];
    // prep(t): set up + assemble the fronts that start at step t, factorize the small ones among them (whole, one launch)
    auto prep = [&](const Step &S, hipStream_t q) -> int {
        const int *starting = L0 + S.start_off;
        if (S.n_start > 0) {
            int e = timed(t_asm, [&]() -> int {
                if (P.recycle) { LCHK(stm_launch_zero_slabs(c, starting, S.n_start, S.asm_maxparts, q)); nlaunch++; }
                LCHK(stm_launch_setup(c, starting, S.n_start, q));
                LCHK(stm_launch_assemble(c, starting, L0 + S.asm_parts_off, S.n_start, S.asm_maxparts, q));
                return 0;
            });
            if (e) return e;
            nlaunch += 2;
        }
        if (S.n_small > 0) {
            int e = timed(t_front, [&]() -> int {
                LCHK(stm_launch_front_wg(c, starting, S.n_small, S.lds_small, q));
                return 0;
            });
            if (e) return e;
            nlaunch++;
        }
        return 0;
    };
    // which kernel takes a panel is a property of the front (stm_use_ca); a step with both kinds gets both launches (each
    // kernel skips the other's fronts).  A front with trailing columns leaves T to its update (dev_tall_group /
    // k_panel_ca), one at its last panel to k_cpack.
    auto panels = [&](const Step &S) -> int {
        const int *act = L0 + S.act_off, *pl = L0 + S.plist_off;
        return timed(t_panel, [&]() -> int {
            if (S.nca_use && !P.serial_panels) { LCHK(stm_launch_panel_ca(c, act, pl, S.n_act, S.nca, 1, st)); nlaunch++; }
            if (S.npipe_use || P.serial_panels) {
                const int lds = (P.serial_panels || (c.dbg & (64 | 256))) ? S.lds_big : S.lds_plan;
                LCHK(stm_launch_panel(c, act, pl, S.n_act, S.nsub, 1, lds, st));
                nlaunch++;
            }
            return 0;
        });
    };
    // trailing update of the column blocks [cb0, cb0 + ncb) of every front in flight (block 0 = the columns of its next
    // panel); gram: the T factors the panel kernels left to the update are built in this launch (row-parallel form; the
    // one-workgroup form builds T in every workgroup)
    auto update = [&](const Step &S, int cb0, int ncb, bool gram, double *Wp, hipStream_t q) -> int {
        const bool split = S.split && g_opt.split_update;
        const bool whole = (cb0 == 0 && gram);                 // the step's whole update: the pair-update classes too
        const bool pairs = whole && S.n_sweep() > 0;
        if (S.n_norm <= 0 || (ncb <= 0 && !(gram && split))) {
            if (!pairs) return 0;
        }
        const int *act = L0 + S.act_off, *pl = L0 + S.plist_off;
        const bool main_ws = (Wp == P.d_Wp.p);
        int *wcnt = main_ws ? P.d_wcnt.p : P.d_wcnt2.p;
        const long long *wl = P.d_wlists.p + S.wp_off;
        return timed(t_upd, [&]() -> int {
            if (S.n_norm > 0 && (ncb > 0 || (gram && split))) {
                if (split && g_opt.fused_update && S.maxsl <= 256 && !P.serial_panels && c.cbskip == 0) {
                    // one launch: C is read and written once (k_upd_f); the epoch of its hand-offs is the step number of the group
                    const int epoch = (int)(&S - SV.data()) + 1 + grp * (1 << 20);
                    LCHK(stm_launch_update_fused(c, act, pl, S.n_norm, cb0, ncb, S.maxsl, Wp, wl, wcnt,
                                                 main_ws ? P.d_wflag.p : P.d_wflag2.p, epoch, gram ? 1 : 0, q));
                    nlaunch++;
                } else if (split) {
                    LCHK(stm_launch_update_split(c, act, pl, S.n_norm, cb0, ncb, S.maxsl, Wp, wl, wcnt, gram ? 1 : 0, q));
                    nlaunch += 2;
                } else {
                    LCHK(stm_launch_update(c, act, pl, S.n_norm, cb0, ncb, q));
                    nlaunch++;
                }
            }
            if (pairs) {
                // panel r of a sweep of w: column blocks 0 .. w-1-r (the columns of the next panels), T by the Gram block; after the
                // last one the w panels at once on everything beyond
                int o = S.n_norm;
                for (int r = 0; r < P.sweep; r++) {
                    const int n = S.n_pk[r];
                    if (n <= 0) continue;
                    LCHK(stm_launch_update_split(c, act + o, pl + o, n, 0, P.sweep - r, S.maxsl_pk[r], Wp, wl + o, wcnt, 1, q));
                    nlaunch += 2;
                    if (r == P.sweep - 1) {
                        if (P.sweep == 4) LCHK(stm_launch_update_quad(c, act + o, pl + o, n, S.maxcbp_po, S.maxsl_pk[r], Wp, wl + o, wcnt, q));
                        else LCHK(stm_launch_update_pair(c, act + o, pl + o, n, S.maxcbp_po, S.maxsl_pk[r], Wp, wl + o, wcnt, q));
                        nlaunch += 3;
                    }
                    o += n;
                }
            }
            return 0;
        });
    };
    // T and block 0 of a step's update in ONE launch (k_upd_f: the slab workgroups of the block meet through global memory; the
    // Gram block of the same launch builds T) -- the look-ahead chain panel(t) -> block 0 -> panel(t+1) then has one launch between
    // two panels instead of three (T, k_upd_w, k_upd_c).  Same bits as the other forms.
    auto update_b0_fused = [&](const Step &S, hipStream_t q) -> int {
        const int *act = L0 + S.act_off, *pl = L0 + S.plist_off;
        const long long *wl = P.d_wlists.p + S.wp_off;
        const int epoch = (int)(&S - SV.data()) + 1 + grp * (1 << 20);
        nlaunch++;
        return timed(t_upd, [&]() -> int {
            LCHK(stm_launch_update_fused(c, act, pl, S.n_norm, 0, 1, S.maxsl, P.d_Wp.p, wl, P.d_wcnt.p, P.d_wflag.p, epoch, 1, q));
            return 0;
        });
    };
    auto post = [&](const Step &S, hipStream_t q) -> int {
        if (S.n_cpk <= 0 && !(P.recycle && S.n_rhp > 0)) return 0;
        return timed(t_cpk, [&]() -> int {
            if (S.n_cpk > 0) { LCHK(stm_launch_cpack(c, L0 + S.cpk_off, L0 + S.cpk_parts_off, S.n_cpk, S.cpk_maxparts, q)); nlaunch++; }
            if (P.recycle && S.n_rhp > 0) {
                // slab recycling: the packed R+H blocks of the fronts that are finished now go to the arena (sizes and places on the
                // device: k_rh_count bumps the arena's pointer); their slabs are free from the next step on
                LCHK(stm_launch_rh_count(c, L0 + S.rhp_off, S.n_rhp, q));
                LCHK(stm_launch_rh_copy(c, L0 + S.rhp_off, L0 + S.rhp_parts_off, S.n_rhp, S.rhp_maxparts, P.d_RH.p, q));
                nlaunch += 2;
            }
            return 0;
        });
    };
    if (req) {
        if (req->step < 0 || req->step >= (int)SV.size()) return fail(STMMQR_ERR_INVALID, "no such step in the group");
        const Step &S = SV[(size_t)req->step];
        cur_step = req->step;
        int e = 0;
        if (req->what & STMMQR_STEP_PREP) e = prep(S, st);
        if (!e && (req->what & STMMQR_STEP_PANEL) && S.n_act > 0) e = panels(S);
        if (!e && (req->what & (STMMQR_STEP_UPDATE | STMMQR_STEP_GRAM)) && S.n_act > 0) {
            if (S.n_sweep() > 0) return fail(STMMQR_ERR_INVALID, "pair-update fronts cannot be stepped (mark the front STMMQR_GROUP_SHARED)");
            int nmine = 0;
            if ((req->what & STMMQR_STEP_UPDATE) && req->cb_first >= 0 && req->cb_first < S.maxcb)
                nmine = (S.maxcb - req->cb_first + c.cbskip) / (1 + c.cbskip);
            if (req->cb_count >= 0) nmine = std::min(nmine, req->cb_count);
            e = update(S, std::max(0, req->cb_first), nmine, (req->what & STMMQR_STEP_GRAM) != 0, P.d_Wp.p, st);
        }
        if (!e && (req->what & STMMQR_STEP_POST)) e = post(S, st);
        P.stats.nlaunch += nlaunch;
        return e;
    }
    // a step goes to the side stream when the update beyond block 0 is worth two cross-stream hand-offs (~25 us):
    // STMMQR_LA_MIN tiles of 256 x 32 (default 2500)
    const long la_min = getenv("STMMQR_LA_MIN") ? atol(getenv("STMMQR_LA_MIN")) : 2500;
    // ... and when the panel workgroups of the step fit the compute units the side stream leaves alone (a wide step
    // fills the GPU with panel workgroups by itself): STMMQR_LA_MAXPWG (default 48)
    // T + block 0 of an offloaded step in ONE launch (update_b0_fused) where the step's panels are expected to reach at most
    // STMMQR_LA_FUSED_ROWS rows (default 5120; 0: never): one launch between two panels of the chain instead of three, and look-ahead then
    // pays from STMMQR_LA_MIN_FUSED tiles (default 1500).  Measured: sme3Dc stand-in 87.2 -> 84.3 ms, default workload 121.4 ->
    // 119.4-120.9; on the 7818-row fronts of c5mini (31 slabs) the fused launch is the slower one (43.5 -> 48.9 ms).
    const int la_fused_rows = getenv("STMMQR_LA_FUSED_ROWS") ? atoi(getenv("STMMQR_LA_FUSED_ROWS")) : 5120;
    const long la_min_fused = getenv("STMMQR_LA_MIN_FUSED") ? atol(getenv("STMMQR_LA_MIN_FUSED")) : 1500;
    // Forward progress of that launch: the slab workgroups of block 0 and of the Gram block wait for each other (bounded), so all
    // of them must be resident at once.  What is guaranteed while the side stream fills the rest of the GPU are the reserved
    // compute units, two k_upd_f workgroups each (248 VGPRs) -- counted with the row BOUND of every front of the step (a rank-
    // deficient front can have more rows than its estimate), not with the estimate the 5120-row rule uses.
    const long la_slots = 2 * (getenv("STMMQR_SIDE_RESERVE") ? std::max(0L, atol(getenv("STMMQR_SIDE_RESERVE"))) : 32L);
    auto b0_fused = [&](const Step &S) -> bool {
        if (la_fused_rows <= 0 || !S.split || !g_opt.split_update || S.maxsl > 256 || S.n_sweep() > 0 || c.cbskip != 0) return false;
        long wgs = 0;
        for (int i = 0; i < S.n_norm; i++) wgs += 2L * stm_upd_nsl(P.fs[P.lists[S.act_off + i]]);
        if (wgs > la_slots) return false;
        for (int i = 0; i < S.n_act; i++)                          // (the rows the panels are expected to reach, not the bound)
            if (stm_panel_rows_est(P.fs[P.lists[S.act_off + i]], P.lists[S.plist_off + i]) > la_fused_rows) return false;
        return true;
    };
    const long la_maxpwg = getenv("STMMQR_LA_MAXPWG") ? atol(getenv("STMMQR_LA_MAXPWG")) : 48;
    auto worth_it = [&](const Step &S) -> bool {
        long tiles = 0, pwg = 0;
        for (int i = 0; i < S.n_act; i++) {
            const FrontSym &fsym = P.fs[P.lists[S.act_off + i]];
            const int p = P.lists[S.plist_off + i];
            const int ncb = stm_upd_ncb(fsym, p);
            if (ncb > 1) tiles += (long)(ncb - 1) * ((stm_panel_rows_est(fsym, p) + STM_UPD_SLAB - 1) / STM_UPD_SLAB);   // (expected rows, not the bound)
            pwg += stm_use_ca(fsym, p, P.plan_algo, P.ca_min) ? stm_ca_slabs(fsym) : stm_tall_launches(fsym, p, P.tall_min);
        }
        return tiles >= (b0_fused(S) ? std::min(la_min, la_min_fused) : la_min) && pwg <= la_maxpwg && S.n_sweep() == 0;      // (pair-update steps stay on one stream)
    };
    // Look-ahead needs the device's side stream (a CU-masked stream, created once per process and released by an atexit
    // handler): it is only created when some step of this group really goes there -- small matrices never touch it.
    // Passenger launches (options.lookahead = 2, the default): ONE stream, no events.  The chain stays  panel(t) -> T(t) + block 0 (one fused
    // launch) -> panel(t+1); the two launches of the update beyond block 0 ride on the chain's launches as extra workgroups behind the
    // chain's own (k_upd_w of step t behind T + block 0 of step t, k_upd_c of step t behind the panels of step t+1: stmmqr_*.hip,
    // "Passenger launches").  Every workgroup does what it does in the serial order: same bits.  A step takes part when its update is the
    // row-parallel form and every workgroup of its fused launch is resident at once (they wait for each other, bounded: two per CU);
    // any other step runs in the serial order after the riders of the step before it.
    const bool pass = g_opt.lookahead >= 2 && !detail && !P.serial_panels && c.cbskip == 0 && g_opt.split_update && !(c.dbg & 512) &&
                      !(getenv("STMMQR_PASSENGERS") && atoi(getenv("STMMQR_PASSENGERS")) == 0);
    bool la = g_opt.lookahead && !pass && !detail && !P.serial_panels && SV.size() > 1;
    if (la) {
        bool any = false;
        for (const Step &S : SV)
            if (S.n_act > 0 && S.maxcb > 1 && worth_it(S)) { any = true; break; }
        la = any;
    }
    if (pass) {
        const long fw_max = getenv("STMMQR_PASS_MAXWG") ? atol(getenv("STMMQR_PASS_MAXWG")) : 384;
        const int pass_rows = getenv("STMMQR_PASS_ROWS") ? atoi(getenv("STMMQR_PASS_ROWS")) : 5120;
        const double pass_k = getenv("STMMQR_PASS_K") ? atof(getenv("STMMQR_PASS_K")) : 1e30;
        const long pass_tiles = getenv("STMMQR_PASS_TILES") ? atol(getenv("STMMQR_PASS_TILES")) : (1L << 40);   // (measured: riding always wins -- 2000: 123 ms, 3000: 117, never: 109.9 on the default workload)
        const int abl = getenv("STMMQR_PASS_ABL") ? atoi(getenv("STMMQR_PASS_ABL")) : 0;   // timing-only ablations (WRONG results): 1 no k_upd_w riders, 2 no k_upd_c riders
        const Step *pend = nullptr;                            // the step whose k_upd_c beyond block 0 is still due
        auto flush_alone = [&]() -> int {
            if (!pend) return 0;
            const Step &Q = *pend;
            pend = nullptr;
            LCHK(stm_launch_update_c(c, L0 + Q.act_off, L0 + Q.plist_off, Q.n_norm, 1, Q.maxcb - 1, Q.maxsl, P.d_Wp2.p,
                                     P.d_wlists.p + Q.wp_off, st));
            nlaunch++;
            return 0;
        };
        for (const Step &S : SV) {
            cur_step = (int)(&S - SV.data());
            int e = prep(S, st);
            if (e) return e;
            if (S.n_act > 0) {
                const int *act = L0 + S.act_off, *pl = L0 + S.plist_off;
                if (S.nca_use) { LCHK(stm_launch_panel_ca(c, act, pl, S.n_act, S.nca, 1, st)); nlaunch++; }
                if (S.npipe_use) {
                    const int lds = (c.dbg & (64 | 256)) ? S.lds_big : S.lds_plan;
                    if (pend && (abl & 2)) pend = nullptr;
                    if (pend) {
                        // a rider has its CU to itself (the panel launch's registers and LDS): 2-3 x the time of k_upd_c's own launch
                        // per tile.  Beyond pass_tiles tiles (a wide step of the lower tree levels) the riders would outlast any panel.
                        long tiles = 0;
                        for (int i = 0; i < pend->n_norm; i++) {
                            const FrontSym &fsym = P.fs[P.lists[pend->act_off + i]];
                            const int pp = P.lists[pend->plist_off + i];
                            const int ncb = stm_upd_ncb(fsym, pp);
                            if (ncb > 1) tiles += (long)(ncb - 1) * ((stm_panel_rows_est(fsym, pp) + STM_UPD_SLAB - 1) / STM_UPD_SLAB);
                        }
                        // (the panel launch they would ride on: ~25 us up to 512 rows, ~48 us up to 4096, ~72 us beyond)
                        int prow = 0;
                        for (int i = 0; i < S.n_act; i++) prow = std::max(prow, stm_panel_rows_est(P.fs[P.lists[S.act_off + i]], P.lists[S.plist_off + i]));
                        const double panel_us = prow <= 512 ? 25.0 : prow <= 4096 ? 48.0 : 72.0;
                        if ((tiles > pass_tiles || (double)tiles > pass_k * panel_us) && (e = flush_alone())) return e;
                    }
                    if (pend && (abl & 4)) {                     // (measurement: the riders as launches of their own, same order)
                        LCHK(stm_launch_panel(c, act, pl, S.n_act, S.nsub, 1, lds, st));
                        if (abl & 8) {
                            const Step &Q = *pend;
                            pend = nullptr;
                            LCHK(stm_launch_panel_pc(c, act, pl, -1, S.nsub, 1, lds, L0 + Q.act_off, L0 + Q.plist_off, Q.n_norm, 1, Q.maxcb - 1,
                                                     Q.maxsl, P.d_Wp2.p, P.d_wlists.p + Q.wp_off, st));
                        }
                        if ((e = flush_alone())) return e;
                    } else if (pend) {
                        const Step &Q = *pend;
                        pend = nullptr;
                        LCHK(stm_launch_panel_pc(c, act, pl, S.n_act, S.nsub, 1, lds, L0 + Q.act_off, L0 + Q.plist_off, Q.n_norm, 1, Q.maxcb - 1,
                                                 Q.maxsl, P.d_Wp2.p, P.d_wlists.p + Q.wp_off, st));
                    } else
                        LCHK(stm_launch_panel(c, act, pl, S.n_act, S.nsub, 1, lds, st));
                    nlaunch++;
                } else if ((e = flush_alone()))
                    return e;
                long fwg = 0;
                int rows_max = 0;                                  // (the rows the panels are expected to reach, not the bound)
                for (int i = 0; i < S.n_norm; i++) {
                    fwg += 2L * stm_upd_nsl(P.fs[P.lists[S.act_off + i]]);
                    rows_max = std::max(rows_max, stm_panel_rows_est(P.fs[P.lists[S.act_off + i]], P.lists[S.plist_off + i]));
                }
                // (beyond ~5000 rows -- 20+ slabs -- the one-launch block 0 is the slower form: its slab workgroups idle while the
                //  partials are added; measured on the 7818-row fronts of c5mini)
                const bool ride = S.split && S.n_sweep() == 0 && S.n_norm > 0 && S.maxcb > 1 && S.maxsl <= 256 && fwg <= fw_max &&
                                  rows_max <= pass_rows && !g_opt.fused_update;
                if (ride) {
                    const int epoch = cur_step + 1 + grp * (1 << 20);
                    if (abl & 4) {
                        LCHK(stm_launch_update_fw(c, act, pl, S.n_norm, 1, S.maxsl, P.d_Wp.p, P.d_wlists.p + S.wp_off, P.d_wcnt.p, P.d_wflag.p,
                                                  epoch, P.d_Wp2.p, P.d_wcnt2.p, st));
                        LCHK(stm_launch_update_w(c, act, pl, S.n_norm, 1, S.maxcb - 1, S.maxsl, P.d_Wp2.p, P.d_wlists.p + S.wp_off, P.d_wcnt2.p, st));
                    } else
                    LCHK(stm_launch_update_fw(c, act, pl, S.n_norm, (abl & 1) ? 1 : S.maxcb, S.maxsl, P.d_Wp.p, P.d_wlists.p + S.wp_off, P.d_wcnt.p,
                                              P.d_wflag.p, epoch, P.d_Wp2.p, P.d_wcnt2.p, st));
                    nlaunch++;
                    pend = &S;
                } else if ((e = update(S, 0, S.maxcb, true, P.d_Wp.p, st)))
                    return e;
            }
            if ((e = post(S, st))) return e;
        }
        int e = flush_alone();
        if (e) return e;
    } else if (!la) {
        for (const Step &S : SV) {
            cur_step = (int)(&S - SV.data());
            int e = prep(S, st);
            if (!e && S.n_act > 0) e = panels(S);
            if (!e && S.n_act > 0) e = update(S, 0, S.maxcb, true, P.d_Wp.p, st);
            if (!e) e = post(S, st);
            if (e) return e;
        }
    } else {
        // Look-ahead of depth one.  The chain  panel(t) -> update of block 0 (the next panel's columns) -> panel(t+1)
        // stays on the plan's stream; the rest of the update of step t, the packing of the fronts that ended at t and the
        // preparation of those that start at t+1 run beside it on the side stream:
        //   main:  [wait prep(t)]  panel(t)  T(t)  record main(t)  [wait side(t-1)]  update block 0
        //   side:  wait main(t)    pack(t)   prep(t+1)  record prep(t+1)   update blocks 1..  record side(t)
        // (side(t) needs the panel and its T only, not block 0: the side stream runs its updates back to back while the
        //  main stream alternates block 0 and the next panel.)  Every kernel does exactly what it does in the serial
        // order (same bits).
        const size_t ns = SV.size();
        for (auto *v : {&P.ev_main, &P.ev_prep, &P.ev_side})
            while (v->size() < ns + 1) {
                hipEvent_t ev = nullptr;
                HIPCHK(hipEventCreateWithFlags(&ev, la_event_flags()));
                v->push_back(ev);
            }
        hipStream_t sd = P.side;
        long side_ev = -1;                                     // last side event the main stream has not waited for
        bool prep_on_side = false;                             // prep(t) was issued on the side stream during step t-1
        int e = 0;
        for (size_t t = 0; t < ns; t++) {
            const Step &S = SV[t];
            if (prep_on_side) HIPCHK(hipStreamWaitEvent(st, P.ev_prep[t], 0));
            else {
                // (what a starting front assembles may have been packed on the side stream)
                if (S.n_start > 0 && side_ev >= 0) { HIPCHK(hipStreamWaitEvent(st, P.ev_side[side_ev], 0)); side_ev = -1; }
                if ((e = prep(S, st))) return e;
            }
            prep_on_side = false;
            if (S.n_act > 0 && (e = panels(S))) return e;
            const bool offload = S.n_act > 0 && S.maxcb > 1 && worth_it(S);
            if (!offload) {
                if (S.n_act > 0) {
                    if (side_ev >= 0) { HIPCHK(hipStreamWaitEvent(st, P.ev_side[side_ev], 0)); side_ev = -1; }
                    if ((e = update(S, 0, S.maxcb, true, P.d_Wp.p, st))) return e;
                }
                if ((e = post(S, st))) return e;
                continue;
            }
            if (b0_fused(S)) {
                if (side_ev >= 0) { HIPCHK(hipStreamWaitEvent(st, P.ev_side[side_ev], 0)); side_ev = -1; }
                if ((e = update_b0_fused(S, st))) return e;                                            // T + block 0
                HIPCHK(hipEventRecord(P.ev_main[t], st));
            } else {
                if ((e = update(S, 0, 0, true, P.d_Wp.p, st))) return e;                               // T
                HIPCHK(hipEventRecord(P.ev_main[t], st));
                if (side_ev >= 0) { HIPCHK(hipStreamWaitEvent(st, P.ev_side[side_ev], 0)); side_ev = -1; }
                if ((e = update(S, 0, 1, false, P.d_Wp.p, st))) return e;                              // block 0
            }
            HIPCHK(hipStreamWaitEvent(sd, P.ev_main[t], 0));
            if ((e = post(S, sd))) return e;
            if (t + 1 < ns && SV[t + 1].n_start > 0) {
                if ((e = prep(SV[t + 1], sd))) return e;
                HIPCHK(hipEventRecord(P.ev_prep[t + 1], sd));
                prep_on_side = true;
            }
            if ((e = update(S, 1, S.maxcb - 1, false, P.d_Wp2.p, sd))) return e;
            HIPCHK(hipEventRecord(P.ev_side[t], sd));
            side_ev = (long)t;
        }
        if (side_ev >= 0) HIPCHK(hipStreamWaitEvent(st, P.ev_side[side_ev], 0));   // join: the caller continues on `stream`
    }
    P.stats.nlaunch += nlaunch;
    P.stats.nlevels += (long)P.glevels[grp].size();
    P.stats.nsteps += (long)P.gsteps[grp].size();
    return 0;
}

int run_pack(stmmqr_plan &P)
{
    hipStream_t st = P.stream;
    const DevCtx c = P.ctx();
    const int *L0 = P.d_lists.p;
    if (P.recycle) {
        // the blocks were staged front by front (run_schedule: post); what is left are the kept fronts' column offsets, the
        // reference's layout of all blocks (Post order: d_fin, used by the download) and the check that the arena held everything
        std::vector<int> kl;
        for (long f = 0; f < P.nf; f++) if (P.kept[(size_t)f]) kl.push_back((int)f);
        if (!kl.empty()) {
            DevBuf<int> d_kl;
            LCHK(d_kl.upload(kl, st));
            DevCtx ck = c;
            ck.rh_top = nullptr;                                  // (no place in the arena: packed on the fly by the download)
            LCHK(stm_launch_rh_count(ck, d_kl.p, (int)kl.size(), st));
            HIPCHK(hipStreamSynchronize(st));
        }
        LCHK(stm_launch_rh_scan(c, L0 + P.post_off, (int)P.nf, P.d_total.p, P.d_fin.p, st));
        long long total = 0, top[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(&total, P.d_total.p, sizeof(long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(top, P.d_rhtop.p, 2 * sizeof(long long), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        P.rh_total = total;
        P.stats.nlaunch += 2;
        if (top[1] != 0) {
            if (!P.rh_grow) P.rh_grow = 1;                            // next: the hard bound (QRsym->maxstack / all recycled slabs)
            else P.overflowed = true;                                 // that too: no recycling for this plan
            P.arena_overflow = true;
            return fail(STMMQR_ERR_OUT_OF_MEMORY, "the packed factors exceed the R+H arena of the slab recycling");
        }
        return 0;
    }
    LCHK(stm_launch_rh_count(c, L0 + P.own_off, P.n_own, st));
    LCHK(stm_launch_rh_scan(c, L0 + P.post_off, (int)P.nf, P.d_total.p, P.d_Rboff.p, st));
    long long total = 0;
    HIPCHK(hipMemcpyAsync(&total, P.d_total.p, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    P.rh_total = total;
    if ((size_t)total > P.d_RH.n) LCHK(P.d_RH.alloc((size_t)(total + total / 64 + 1024)));   // (a little room: a refactorization with other dead columns)
    LCHK(stm_launch_rh_copy(c, L0 + P.own_off, L0 + P.rh_parts_off, P.n_own, P.rh_maxparts, P.d_RH.p, st));
    P.stats.nlaunch += 3;
    return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char *stmmqr_version(void) { return "stmmqr_hip 0.1 (gfx950)"; }
const char *stmmqr_last_error(void) { return g_err.c_str(); }
void stmmqr_get_options(stmmqr_options *o) { if (o) *o = g_opt; }
void stmmqr_set_options(const stmmqr_options *o)
{
    if (!o) return;
    g_opt = *o;
    if (g_opt.big_front_cols < 1) g_opt.big_front_cols = 1;
}
void stmmqr_set_common_layout(const stm_common_layout *l) { if (l) g_layout = *l; }
void stmmqr_get_common_layout(stm_common_layout *l) { if (l) *l = g_layout; }

/* End of use (optional): waits for the device and releases what the library holds process-wide.  A host program that
 * dlopen()s the library calls it before returning from main; the library has no static object whose destructor calls
 * into the HIP runtime, so nothing else happens at exit / dlclose. */
int stmmqr_device_alloc(size_t bytes, void **ptr)
{
    if (!ptr) return fail(STMMQR_ERR_INVALID, "null argument");
    *ptr = nullptr;
    if (hipMalloc(ptr, bytes ? bytes : 1) != hipSuccess) return fail(STMMQR_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    return 0;
}
/* device-to-device copy, complete on return (after everything `hip_stream` -- may be NULL -- held before it); for callers that
 * implement a stmmqr_transport of their own on one GPU (tests) */
int stmmqr_device_copy(void *dst, const void *src, size_t bytes, void *hip_stream)
{
    if (hip_stream) HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    HIPCHK(hipDeviceSynchronize());
    return 0;
}
int stmmqr_device_free(void *ptr)
{
    if (ptr && hipFree(ptr) != hipSuccess) return fail(STMMQR_ERR_DEVICE, "hipFree failed");
    return 0;
}

void stmmqr_plan_cache_clear(void);
void stmmqr_shutdown(void)
{
    stmmqr_plan_cache_clear();
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) == hipSuccess && cnt > 0) {
        (void)hipDeviceSynchronize();
        int dev = 0;
        (void)hipGetDevice(&dev);
        destroy_side_streams();
        (void)hipSetDevice(dev);
    }
}

int stmmqr_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}
const char *stmmqr_device_name(int device)
{
    static thread_local char name[256];
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return "";
    snprintf(name, sizeof name, "%s", prop.gcnArchName);
    return name;
}

int chunk_getSettings(size_t a, size_t b, size_t c, size_t d)
{
    g_chunk[0] = a; g_chunk[1] = b; g_chunk[2] = c; g_chunk[3] = d;
    return 0;
}

stmmqr_plan *stmmqr_plan_create(const stmmqr_symbolic_view *sym, int device, int *status)
{
    int st = 0;
    stmmqr_plan *P = nullptr;
    if (!sym) st = fail(STMMQR_ERR_INVALID, "null symbolic view");
    if (!st) st = ensure_device(device);
    if (!st) {
        P = new (std::nothrow) stmmqr_plan();
        if (!P) st = fail(STMMQR_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    if (!st) {
        (void)hipGetDevice(&P->device);
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithPriority(&P->stream, hipStreamNonBlocking, prio_hi) != hipSuccess)
            st = fail(STMMQR_ERR_DEVICE, "hipStreamCreate failed");
        for (auto &e : P->ev)
            if (!st && hipEventCreate(&e) != hipSuccess) st = fail(STMMQR_ERR_DEVICE, "hipEventCreate failed");
    }
    if (!st) {
        const double t0 = now_ms();
        st = build_plan(*P, *sym);
        if (!st) P->stats.ms_host = now_ms() - t0;
    }
    if (st && P) { delete P; P = nullptr; }
    if (status) *status = st;
    return P;
}

void stmmqr_plan_release_rings(stmmqr_plan *plan);
void stmmqr_plan_destroy(stmmqr_plan *plan)
{
    if (plan) stmmqr_plan_release_rings(plan);
    delete plan;
}

int stmmqr_plan_set_pattern(stmmqr_plan *plan, const stm_long *Ap, const stm_long *Ai)
{
    if (!plan) return fail(STMMQR_ERR_INVALID, "null plan");
    HIPCHK(hipSetDevice(plan->device));
    return set_pattern(*plan, Ap, Ai);
}

// ---- phased interface: begin -> factorize_group(g) ... -> finish.  stmmqr_factorize_device = all of it ----
int stmmqr_factorize_begin(stmmqr_plan *plan, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                           int ax_on_device, double tol, stm_long ntol)
{
    if (!plan || !Ax) return fail(STMMQR_ERR_INVALID, "null plan / values");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    if (Ap && Ai) {
        int e = set_pattern(P, Ap, Ai);
        if (e) return e;
    }
    if (!P.pattern_set) return fail(STMMQR_ERR_INVALID, "pattern of A was never given");
    if (P.arena_overflow && P.recycle && !P.whole_call) {
        // phased use (begin / group / finish by the caller): the previous factorization of this plan did not fit the R+H arena
        // (finish returned OUT_OF_MEMORY and run_pack noted rh_grow / overflowed).  Those only take effect when the schedule is
        // rebuilt, which stmmqr_factorize_device does in its own retry loop; here the rebuild happens at the next begin, so a
        // caller that simply tries again gets the arena at its hard bound (then no recycling) instead of the same failure.
        P.arena_overflow = false;
        P.begun = false;
        std::vector<int> grp(P.group.begin(), P.group.end());
        for (size_t f = 0; f < grp.size(); f++) if (f < P.shared.size() && P.shared[f]) grp[f] |= STMMQR_GROUP_SHARED;
        int e = stmmqr_plan_set_groups(plan, grp.data());
        if (e) return e;
    }
    const double host_ms_plan = P.stats.ms_host;
    P.stats = stmmqr_stats();
    P.stats.ms_host = host_ms_plan;
    P.factored = false;
    P.rowmap_ready = false;
    P.scr_valid = false;
    P.t4_valid = false;
    P.evused = 0;
    P.begun = true;
    P.first_group = true;
    if (!P.do_rank) tol = -1;                                  // SparseQR_factorize.c:285-289
    P.last_tol = tol; P.last_ntol = ntol;
    hipStream_t st = P.stream;
    HIPCHK(hipEventRecord(P.ev[0], st));
    if (P.anz > 0) {
        HIPCHK(hipMemcpyAsync(P.d_Ax.p, Ax, (size_t)P.anz * sizeof(double),
                              ax_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipEventRecord(P.ev[1], st));
    return reset_factorization(P);
}

int stmmqr_factorize_group(stmmqr_plan *plan, int group, int detail)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    int e = 0;
    const bool graph_ok = g_opt.use_graph && !detail && group == 0 && P.first_group && !getenv("STMMQR_DUMPLV");
    if (graph_ok) {
        // replay the step schedule of group 0 as a hipGraph, captured once per plan and per everything that travels in the
        // kernel arguments or decides what is launched: (tol, ntol, debug mask), the schedule generation (set_groups
        // rebuilds the lists and may move the workspaces) and the run-time options
        const DevCtx c = P.ctx();
        // (everything run_schedule reads at capture time and that decides WHAT is launched or travels in the kernel arguments: the
        //  run-time options incl. pair_update, STMMQR_TUNE, and the look-ahead thresholds of the environment)
        auto envl = [](const char *k, long dflt) { return getenv(k) ? atol(getenv(k)) : dflt; };
        long long optkey = 1469598103934665603LL;
        for (long long v : {(long long)g_opt.lookahead, (long long)g_opt.split_update, (long long)g_opt.fused_update, (long long)g_opt.pair_update,
                            (long long)P.serial_panels, (long long)c.tune, (long long)envl("STMMQR_LA_MIN", 2500), (long long)envl("STMMQR_LA_MIN_FUSED", 1500),
                            (long long)envl("STMMQR_LA_FUSED_ROWS", 5120), (long long)envl("STMMQR_LA_MAXPWG", 48), (long long)envl("STMMQR_LA_SYSFENCE", 0),
                            (long long)envl("STMMQR_SIDE_RESERVE", 32)})
            optkey = (optkey ^ v) * 1099511628211LL;
        if (!P.graph_exec || P.graph_tol != c.tol || P.graph_ntol != c.ntol || P.graph_dbg != c.dbg || P.graph_gen != P.sched_gen ||
            P.graph_opt != optkey) {
            if (P.graph_exec) { (void)hipGraphExecDestroy(P.graph_exec); P.graph_exec = nullptr; }
            P.graph_nlaunch = 0;
            // everything run_schedule creates lazily is created BEFORE the capture: the device's side stream (device
            // properties, a CU-masked stream, an atexit handler) and the per-step events of the look-ahead
            if (g_opt.lookahead && !P.serial_panels && P.gsteps[0].size() > 1) {
                if (!P.side) P.side = side_stream_for(P.device);
                for (auto *v : {&P.ev_main, &P.ev_prep, &P.ev_side})
                    while (v->size() < P.gsteps[0].size() + 1) {
                        hipEvent_t ev = nullptr;
                        HIPCHK(hipEventCreateWithFlags(&ev, la_event_flags()));
                        v->push_back(ev);
                    }
            }
            const long nl0 = P.stats.nlaunch;
            hipGraph_t gph = nullptr;
            // the side stream is shared by the plans of a device: captures are serialised against each other (a plan that
            // factorizes without a graph while another one captures is the caller's to avoid, as for any shared stream)
            static std::mutex capture_mu;
            std::lock_guard<std::mutex> lock(capture_mu);
            HIPCHK(hipStreamBeginCapture(P.stream, hipStreamCaptureModeThreadLocal));
            e = run_schedule(P, false, 0);
            const hipError_t ce = hipStreamEndCapture(P.stream, &gph);
            if (e) { if (gph) (void)hipGraphDestroy(gph); return e; }
            HIPCHK(ce);
            HIPCHK(hipGraphInstantiate(&P.graph_exec, gph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(gph);
            P.graph_tol = c.tol; P.graph_ntol = c.ntol; P.graph_dbg = c.dbg; P.graph_gen = P.sched_gen; P.graph_opt = optkey;
            P.graph_nlaunch = P.stats.nlaunch - nl0;
        } else {
            // (the statistics run_schedule accumulates on the host)
            P.stats.nlaunch += P.graph_nlaunch;
            P.stats.nlevels += (long)P.glevels[0].size();
            P.stats.nsteps += (long)P.gsteps[0].size();
        }
        HIPCHK(hipGraphLaunch(P.graph_exec, P.stream));
    } else
        e = run_schedule(P, detail != 0, group);
    // Phased use (begin / group / finish called by the host: the sharded path): a bounded panel wait that ran out is found
    // HERE and the group is run again with one-workgroup panels (no inter-workgroup waits), exactly as
    // stmmqr_factorize_device does for the whole factorization -- the other groups and the imported fronts are not touched.
    if (!e && !P.whole_call && !P.serial_panels) {
        // (four bytes: the kernels raise abort[1] beside the front's own perr -- not a copy of every FrontNum per phase)
        int failed = 0;
        HIPCHK(hipMemcpyAsync(&failed, P.d_abort.p + 1, sizeof(int), hipMemcpyDeviceToHost, P.stream));
        HIPCHK(hipStreamSynchronize(P.stream));
        if (failed) {
            if (g_opt.verbose) fprintf(stderr, "[stmmqr_hip] a panel wait ran out in group %d: running it again with one-workgroup panels\n", group);
            P.stats.retries++;
            e = reset_group(P, group);
            P.serial_panels = true;
            if (!e) e = run_schedule(P, detail != 0, group);
            P.serial_panels = false;
        }
    }
    P.first_group = false;
    return e;
}

int stmmqr_factorize_finish(stmmqr_plan *plan, stmmqr_stats *stats)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    hipStream_t st = P.stream;
    HIPCHK(hipEventRecord(P.ev[4], st));
    int e = run_pack(P);
    if (e) return e;
    HIPCHK(hipEventRecord(P.ev[5], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, P.ev[0], P.ev[1])); P.stats.ms_h2d = ms;
    HIPCHK(hipEventElapsedTime(&ms, P.ev[1], P.ev[5])); P.stats.ms_total = ms;
    HIPCHK(hipEventElapsedTime(&ms, P.ev[4], P.ev[5])); P.stats.ms_pack += ms;
    // STMMQR_DUMPSTEPS=file (detail runs, diagnosis): one line per timeline step with what ran and how long it took
    FILE *dump = (P.evused && getenv("STMMQR_DUMPSTEPS")) ? fopen(getenv("STMMQR_DUMPSTEPS"), "w") : nullptr;
    std::vector<float> st_panel, st_upd;
    if (dump && !P.gsteps.empty()) { st_panel.assign(P.gsteps[0].size(), 0.f); st_upd.assign(P.gsteps[0].size(), 0.f); }
    for (size_t q = 0; q < P.evused; q++) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, P.evpairs[q].a, P.evpairs[q].b));
        if (dump && (size_t)P.evpairs[q].step < st_panel.size()) {
            if (P.evpairs[q].cat == 2) st_panel[(size_t)P.evpairs[q].step] += t;
            if (P.evpairs[q].cat == 3) st_upd[(size_t)P.evpairs[q].step] += t;
        }
        switch (P.evpairs[q].cat) {
            case 0: P.stats.ms_assemble += t; break;
            case 1: P.stats.ms_front += t; P.stats.ms_small += t; break;
            case 2: P.stats.ms_front += t; P.stats.ms_panel += t; P.stats.npanel_launch++; break;
            case 3: P.stats.ms_front += t; P.stats.ms_update += t; P.stats.nupdate_launch++; break;
            default: P.stats.ms_pack += t; break;
        }
    }
    P.evused = 0;
    if (dump) {
        for (size_t t = 0; t < st_panel.size(); t++) {
            const Step &S = P.gsteps[0][t];
            int maxrows = 0, pwg = 0;
            long tiles = 0;
            for (int i = 0; i < S.n_act; i++) {
                const FrontSym &fsym = P.fs[P.lists[S.act_off + i]];
                const int p = P.lists[S.plist_off + i];
                maxrows = std::max(maxrows, stm_panel_rows_est(fsym, p));
                pwg += stm_use_ca(fsym, p, P.plan_algo, P.ca_min) ? stm_ca_slabs(fsym) : stm_tall_launches(fsym, p, P.tall_min);
                tiles += (long)stm_upd_ncb(fsym, p) * stm_upd_nsl(fsym);
            }
            fprintf(dump, "%zu n_act %d maxrows %d nsub %d pwg %d nca_use %d maxcb %d maxsl %d split %d tiles %ld panel_us %.1f upd_us %.1f\n", t,
                    S.n_act, maxrows, S.nsub, pwg, S.nca_use, S.maxcb, S.maxsl, S.split, tiles, 1e3 * st_panel[t], 1e3 * st_upd[t]);
        }
        fclose(dump);
    }

    // per-front numeric summary (small): flops, ranks
    P.h_fnum.resize((size_t)std::max(1L, P.nf));
    if (P.nf > 0)
        HIPCHK(hipMemcpy(P.h_fnum.data(), P.d_fnum.p, (size_t)P.nf * sizeof(FrontNum), hipMemcpyDeviceToHost));
    double flops = 0, bytes_asm = 0, bytes_pack = 0, fl_upd = 0, fl_upd_pair = 0;
    long rank = 0;
    for (long f = 0; f < P.nf; f++) {
        if (P.group[f] < 0) continue;                  // factorized elsewhere
        const FrontNum &nm = P.h_fnum[f];
        const FrontSym &s = P.fs[f];
        flops += nm.flops;
        fl_upd += nm.flops_upd;
        if ((size_t)f < P.pair_front.size() && P.pair_front[f]) fl_upd_pair += nm.flops_upd;
        rank += nm.rank;
        if (nm.perr) {
            P.panel_wait_failed = true;
            return fail(STMMQR_ERR_DEVICE, "a panel workgroup gave up waiting for its neighbours (device shared with another job?)");
        }
        const double cn = s.fn - s.fp, cm = nm.cm;
        const double csize = cm * (cm + 1) / 2 + cm * (cn - cm);
        bytes_asm += 8.0 * ((double)nm.fm * s.fn) + 8.0 * csize;   // F first write + child C read (as a child)
        bytes_pack += 16.0 * (csize + (double)nm.rsize);
    }
    bytes_asm += 8.0 * (double)P.anz + P.bytes_assemble_idx;
    if (getenv("STMMQR_DBG") && (atoi(getenv("STMMQR_DBG")) & 32) && getenv("STMMQR_TIMELINE")) {
        // -DSTMMQR_STAMPS builds: wall-clock (100 MHz) timeline of panel 1 of the LAST front that ran one (the root), per column group
        std::vector<unsigned long long> hb(2048);
        HIPCHK(hipMemcpy(hb.data(), P.d_dbg.p, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 16; b++) if (hb[16 + 64 * b]) t0 = std::min(t0, hb[16 + 64 * b]);
        for (int b = 0; b < 16; b++) {
            const unsigned long long *q = &hb[16 + 64 * b];
            if (!q[0]) continue;
            fprintf(stderr, "[panel timeline] group %2d:", b);
            auto us = [&](int i) { return q[i] ? 0.01 * (double)(q[i] - t0) : -1.0; };
            fprintf(stderr, " start %.2f loaded %.2f |", us(0), us(1));
            for (int h = 0; h < 6 && q[2 + 3 * h]; h++) fprintf(stderr, " wait %.2f vload %.2f applied %.2f |", us(2 + 3 * h), us(3 + 3 * h), us(4 + 3 * h));
            fprintf(stderr, " cols");
            for (int j = 0; j < 8; j++) if (q[20 + j]) fprintf(stderr, " %.2f", us(20 + j));
            fprintf(stderr, " | published %.2f final %.2f\n", us(30), us(31));
        }
    }
    if (getenv("STMMQR_DBG") && (atoi(getenv("STMMQR_DBG")) & 32768)) {
        unsigned long long hb[64];
        HIPCHK(hipMemcpy(hb, P.d_dbg.p, sizeof hb, hipMemcpyDeviceToHost));
        const double n = hb[56] ? 100.0 * (double)hb[56] : 1.0;
        fprintf(stderr, "[block-0 launch, slab 0, us] descriptors+loads issued %.2f  phase 1 %.2f  partials+ticket %.2f  wait %.2f  sums %.2f  T %.2f  W2 %.2f  phase 2 %.2f  (%llu)\n",
                hb[48] / n, hb[49] / n, hb[50] / n, hb[51] / n, hb[52] / n, hb[53] / n, hb[54] / n, hb[55] / n, hb[56]);
        HIPCHK(hipMemset(P.d_dbg.p, 0, sizeof hb));
    }
    if (getenv("STMMQR_DBG") && (atoi(getenv("STMMQR_DBG")) & 48)) {
        unsigned long long hb[64];
        HIPCHK(hipMemcpy(hb, P.d_dbg.p, sizeof hb, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pipeline panels by actual rows] <=128 %llu  <=256 %llu  <=512 %llu  <=1024 %llu  <=2048 %llu  <=4096 %llu  more %llu;  of the <=512: %llu with a row estimate > 512\n",
                hb[32], hb[33], hb[34], hb[35], hb[36], hb[37], hb[38], hb[39]);
        fprintf(stderr, "[k_upd_w workgroups] launched %llu  with work %llu\n", hb[40], hb[41]);
        fprintf(stderr, "[panel cycles, summed over workgroups] stage-in+apply %llu  columns %llu  write-back %llu  - %llu  gram %llu\n",
                hb[0], hb[1], hb[2], hb[3], hb[4]);
        fprintf(stderr, "[last group of the panel pipeline, cycles] load %llu  waits %llu  apply-loads %llu  applies %llu  factor %llu  gram %llu\n", hb[6],
                hb[7], hb[12], hb[8] + hb[11], hb[9], hb[10]);
        fprintf(stderr, "[panel workgroups] %llu  cycles %llu  100 MHz ticks %llu  => %.3f GHz, %.2f us per workgroup\n", hb[46], hb[44], hb[45],
                hb[45] ? 0.1 * (double)hb[44] / (double)hb[45] : 0.0, hb[46] ? 0.01 * (double)hb[45] / (double)hb[46] : 0.0);
        fprintf(stderr, "[Gram-based panels] panels %llu  refresh rounds %llu  slab workgroups %llu\n", hb[13], hb[14], hb[15]);
        HIPCHK(hipMemset(P.d_dbg.p, 0, sizeof hb));
    }
    P.rank = rank;
    P.stats.flops = flops;
    P.stats.flops_update = fl_upd;
    P.stats.flops_update_pair = fl_upd_pair;
    P.stats.bytes_assemble = bytes_asm;
    P.stats.bytes_pack = bytes_pack;
    P.stats.device_bytes = P.device_bytes();
    if (getenv("STMMQR_MEMDUMP")) {
        auto gb = [](const auto &b) { return (double)b.n * sizeof(*b.p) * 1e-9; };
        fprintf(stderr, "[stmmqr_hip] device memory (GB): fronts %.3f  contribution blocks %.3f  R+H %.3f  update workspaces %.3f + %.3f  "
                        "kept T %.3f  pair -Y %.3f  Sx/Ax/smap %.3f  per-column arrays %.3f  total %.3f  (recycle %d, kept fronts %d, maxstack %.3f)\n",
                gb(P.d_F), gb(P.d_C), gb(P.d_RH), gb(P.d_Wp), gb(P.d_Wp2), gb(P.d_Tall), gb(P.d_Ypend), gb(P.d_Sx) + gb(P.d_Ax) + gb(P.d_smap),
                gb(P.d_Stair) + gb(P.d_Tau) + gb(P.d_Hii) + gb(P.d_Cmap) + gb(P.d_Cursor) + gb(P.d_Rhoff) + gb(P.d_Rjrel) + gb(P.d_Sjrel),
                P.stats.device_bytes * 1e-9, (int)P.recycle, (int)std::count(P.kept.begin(), P.kept.end(), (char)1), 8e-9 * (double)P.maxstack);
    }
    P.factored = true;
    P.begun = false;
    if (stats) *stats = P.stats;
    return 0;
}

int stmmqr_factorize_device(stmmqr_plan *plan, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                            int ax_on_device, double tol, stm_long ntol, stmmqr_stats *stats)
{
    if (!plan) return fail(STMMQR_ERR_INVALID, "null plan");
    const bool detail = g_opt.verbose >= 2 || (stats && stats->nlaunch == -1);
    // The multi-workgroup panel kernels wait for each other inside a launch (bounded).  Should such a wait ever run out
    // (the workgroups of a launch are not guaranteed to run together: a GPU shared with another job), the factorization is
    // not lost: it is run once more with every panel factorized by ONE workgroup (dev_panel: no inter-workgroup wait
    // anywhere), slower but independent of co-residency.
    int arena_retries = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        plan->serial_panels = (attempt == 1);
        plan->panel_wait_failed = false;
        plan->whole_call = true;
        int e = stmmqr_factorize_begin(plan, Ap, Ai, Ax, ax_on_device, tol, ntol);
        if (!e) plan->stats.retries = (attempt == 1 ? 1 : 0) + arena_retries;   // (visible in stmmqr_stats: bench.py asserts 0)
        for (int g = 0; g < (int)plan->glevels.size() && !e; g++) e = stmmqr_factorize_group(plan, g, detail);
        if (!e) e = stmmqr_factorize_finish(plan, stats);
        plan->whole_call = false;
        if (e && plan->arena_overflow && plan->recycle) {
            // the packed factors did not fit the arena: sized from the full-rank estimate -> once more with the hard bound
            // (QRsym->maxstack); at the hard bound (never seen) -> once more with every front in a slab of its own and the packed
            // blocks placed at the end, as sharded plans always run
            plan->arena_overflow = false;
            if (g_opt.verbose) fprintf(stderr, "[stmmqr_hip] R+H arena overflow: factorizing again %s\n", plan->overflowed ? "without slab recycling" : "with the arena at its hard bound");
            plan->begun = false;
            std::vector<int> grp(plan->group.begin(), plan->group.end());
            for (size_t f = 0; f < grp.size(); f++) if ((size_t)f < plan->shared.size() && plan->shared[f]) grp[f] |= STMMQR_GROUP_SHARED;
            int e2 = stmmqr_plan_set_groups(plan, grp.data());        // (rh_grow / overflowed: build_schedule sizes the arena anew)
            if (e2) return e2;
            arena_retries++;
            attempt = -1;                                             // (both attempts again)
            Ap = nullptr; Ai = nullptr;
            continue;
        }
        const bool retry = e && plan->panel_wait_failed && attempt == 0;
        plan->serial_panels = false;
        if (!retry) return e;
        if (g_opt.verbose) fprintf(stderr, "[stmmqr_hip] a panel wait ran out: factorizing again with one-workgroup panels\n");
        if (stats && detail) stats->nlaunch = -1;
        Ap = nullptr; Ai = nullptr;                  // (the pattern is set; begin uploads / copies the values again)
    }
    return fail(STMMQR_ERR_DEVICE, "panel kernels failed twice");
}

// ---- multi-GPU support: regroup the fronts, move contribution blocks in and out of a plan --------------
int stmmqr_plan_set_groups(stmmqr_plan *plan, const int *group)
{
    if (!plan || !group) return fail(STMMQR_ERR_INVALID, "null plan / groups");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    // validate BEFORE anything of the plan changes: a refused call leaves groups and schedule as they were
    auto gid = [&](long f) { return group[f] < 0 ? -1 : (group[f] & ~STMMQR_GROUP_SHARED); };
    for (long f = 0; f < P.nf; f++)
        for (long q = P.Childp[f]; q < P.Childp[f + 1]; q++)
            if (gid(f) >= 0 && gid(P.Child[q]) > gid(f))
                return fail(STMMQR_ERR_INVALID, "a child is scheduled in a later phase than its parent");
    {
        // a shared front is alone in its group and takes the panel-by-panel path
        std::vector<int> cnt, nsh;
        for (long f = 0; f < P.nf; f++) {
            const int g = gid(f);
            if (g < 0) continue;
            if (g >= (1 << 24)) return fail(STMMQR_ERR_INVALID, "group id out of range");
            if ((size_t)g >= cnt.size()) { cnt.resize((size_t)g + 1, 0); nsh.resize((size_t)g + 1, 0); }
            cnt[(size_t)g]++;
            if (group[f] & STMMQR_GROUP_SHARED) {
                nsh[(size_t)g]++;
                if (!(P.fs[f].fn >= g_opt.big_front_cols && P.fs[f].fm_ub >= 64))
                    return fail(STMMQR_ERR_INVALID, "a shared front must be one of the large fronts (options.big_front_cols)");
            }
        }
        for (size_t g = 0; g < cnt.size(); g++)
            if (nsh[g] > 0 && cnt[g] != 1) return fail(STMMQR_ERR_INVALID, "a shared front must be alone in its group");
    }
    if (P.begun) return fail(STMMQR_ERR_INVALID, "stmmqr_plan_set_groups between factorize_begin and factorize_finish");
    HIPCHK(hipStreamSynchronize(P.stream));
    P.shared.assign((size_t)std::max(1L, P.nf), 0);
    for (long f = 0; f < P.nf; f++) {
        P.group[f] = gid(f);
        P.shared[(size_t)f] = (group[f] >= 0 && (group[f] & STMMQR_GROUP_SHARED)) ? 1 : 0;
    }
    // a captured schedule describes the old step lists and workspaces
    if (P.graph_exec) { (void)hipGraphExecDestroy(P.graph_exec); P.graph_exec = nullptr; }
    P.graph_nlaunch = 0;
    assign_arenas(P);                                        // (the arenas follow at the next stmmqr_factorize_begin)
    P.factored = false;                                      // (the factors of the old grouping live at the old offsets)
    std::vector<int> tslot;
    build_schedule(P, tslot);                                // (a plan that holds the whole tree again recycles its slabs: new offsets)
    LCHK(P.d_fs.upload(P.fs, P.stream));
    LCHK(upload_recycle(P));
    // workspaces only grow (a regrouping of the same tree usually needs what it needed before)
    auto grow = [](auto &buf, size_t n) -> int { return buf.n >= n && buf.p ? 0 : buf.alloc(n); };
    LCHK(grow(P.d_T, (size_t)STM_PD_RING * P.tslots * STM_NB * STM_NB));
    LCHK(grow(P.d_Gp, (size_t)P.tslots * (P.gp_slabs + 1) * STM_NB * STM_NB));
    LCHK(grow(P.d_Wp, (size_t)P.wp_doubles));
    LCHK(grow(P.d_Ypend, (size_t)std::max(1LL, P.yp_doubles)));
    LCHK(P.d_ypoff.upload(P.ypoff, P.stream));
    LCHK(grow(P.d_Wp2, (size_t)std::max(1LL, P.wp2_doubles)));
    P.wcnt_n = std::max(P.wcnt_n, (size_t)(P.wp_doubles / (STM_NB * 32) + 1));
    LCHK(grow(P.d_wcnt, P.wcnt_n));
    LCHK(grow(P.d_wcnt2, P.wcnt_n));
    LCHK(grow(P.d_wflag, P.wcnt_n));
    if (!P.d_abort.p) LCHK(P.d_abort.alloc(2));
    LCHK(grow(P.d_wflag2, P.wcnt_n));
    HIPCHK(hipMemset(P.d_wcnt.p, 0, P.wcnt_n * sizeof(int)));
    HIPCHK(hipMemset(P.d_wcnt2.p, 0, P.wcnt_n * sizeof(int)));
    LCHK(P.d_tslot.upload(tslot, P.stream));
    LCHK(P.d_lists.upload(P.lists, P.stream));
    LCHK(P.d_wlists.upload(P.wlists, P.stream));
    HIPCHK(hipStreamSynchronize(P.stream));
    return 0;
}

// Does front f have a contribution-block slot on this plan, allocated, that holds `csize` doubles?  A front that is neither
// factorized here nor a child of a front that is has NO slot (its coff is 0: the first resident block's), and the arenas exist
// only between the first stmmqr_factorize_begin after a (re)grouping and the next regrouping.
static int check_c_slot(const stmmqr_plan &P, stm_long f, long long csize, const char *what)
{
    if (f < 0 || f >= P.nf) return fail(STMMQR_ERR_INVALID, std::string(what) + ": no such front");
    if (P.recycle)
        return fail(STMMQR_ERR_INVALID, std::string(what) + ": this plan holds the whole tree and recycles its contribution blocks "
                                        "(nothing to exchange; regroup with stmmqr_plan_set_groups first)");
    if ((size_t)f >= P.has_c.size() || !P.has_c[(size_t)f])
        return fail(STMMQR_ERR_INVALID, std::string(what) + ": the front has no contribution-block slot on this plan (it is neither "
                                        "factorized here nor a child of a front that is)");
    if (!P.d_C.p || P.d_C.n != (size_t)P.carena)
        return fail(STMMQR_ERR_INVALID, std::string(what) + ": the arenas of the current grouping are not allocated yet "
                                        "(stmmqr_factorize_begin comes first)");
    if (csize < 0 || csize > P.c_slot[(size_t)f])
        return fail(STMMQR_ERR_INVALID, std::string(what) + ": the block exceeds the front's slot");
    return 0;
}

/* info[0..5] = fm, rank, cm, csize, fn, fp of front f after it has been factorized (or imported) here */
int stmmqr_plan_front_info(stmmqr_plan *plan, stm_long f, stm_long *info)
{
    if (!plan || f < 0 || f >= plan->nf || !info) return fail(STMMQR_ERR_INVALID, "bad front");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    HIPCHK(hipStreamSynchronize(P.stream));
    FrontNum nm;
    HIPCHK(hipMemcpy(&nm, P.d_fnum.p + f, sizeof nm, hipMemcpyDeviceToHost));
    const long cn = P.fs[f].fn - P.fs[f].fp, cm = nm.cm;
    info[0] = nm.fm; info[1] = nm.rank; info[2] = cm; info[3] = cm * (cm + 1) / 2 + cm * (cn - cm);
    info[4] = P.fs[f].fn; info[5] = P.fs[f].fp;
    return 0;
}

/* copy out the packed contribution block (csize doubles) and the cm row ids of front f */
int stmmqr_plan_export_front(stmmqr_plan *plan, stm_long f, double *C, stm_long *rows, int c_on_device)
{
    stm_long info[6];
    int e = stmmqr_plan_front_info(plan, f, info);
    if (e) return e;
    stmmqr_plan &P = *plan;
    if ((e = check_c_slot(P, f, info[3], "stmmqr_plan_export_front"))) return e;
    if (info[3] > 0 && C)
        HIPCHK(hipMemcpy(C, P.d_C.p + P.fs[f].coff, (size_t)info[3] * sizeof(double),
                         c_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    if (info[2] > 0 && rows) {
        std::vector<int> r32((size_t)info[2]);
        HIPCHK(hipMemcpy(r32.data(), P.d_Hii.p + P.fs[f].hip + info[1], (size_t)info[2] * sizeof(int), hipMemcpyDeviceToHost));
        for (long i = 0; i < info[2]; i++) rows[i] = r32[i];
    }
    return 0;
}

/* install the contribution block of a front that was factorized on another device */
int stmmqr_plan_import_front(stmmqr_plan *plan, stm_long f, stm_long fm, stm_long rank, stm_long cm, const double *C,
                             const stm_long *rows, int c_on_device)
{
    if (!plan || f < 0 || f >= plan->nf) return fail(STMMQR_ERR_INVALID, "bad front");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    const long cn = P.fs[f].fn - P.fs[f].fp;
    if (cm < 0 || cm > cn || rank < 0 || rank + cm > P.fs[f].fm_ub)
        return fail(STMMQR_ERR_INVALID, "imported front does not fit the symbolic bounds");
    const long csize = cm * (cm + 1) / 2 + cm * (cn - cm);
    if (!P.begun) return fail(STMMQR_ERR_INVALID, "stmmqr_plan_import_front outside factorize_begin / factorize_finish (begin resets every front's state)");
    if (int e = check_c_slot(P, f, csize, "stmmqr_plan_import_front")) return e;
    if ((cm > 0 && !rows) || (csize > 0 && !C)) return fail(STMMQR_ERR_INVALID, "stmmqr_plan_import_front: null block / row ids");
    HIPCHK(hipStreamSynchronize(P.stream));
    FrontNum nm;
    memset(&nm, 0, sizeof nm);
    nm.fm = (int)fm; nm.rank = (int)rank; nm.cm = (int)cm; nm.done = 1; nm.g = (int)rank;
    HIPCHK(hipMemcpy(P.d_fnum.p + f, &nm, sizeof nm, hipMemcpyHostToDevice));
    if (csize > 0)
        HIPCHK(hipMemcpy(P.d_C.p + P.fs[f].coff, C, (size_t)csize * sizeof(double),
                         c_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    if (cm > 0) {
        std::vector<int> r32((size_t)cm);
        for (long i = 0; i < cm; i++) r32[i] = (int)rows[i];
        HIPCHK(hipMemcpy(P.d_Hii.p + P.fs[f].hip + rank, r32.data(), (size_t)cm * sizeof(int), hipMemcpyHostToDevice));
    }
    return 0;
}

// ---- a front shared between plans (one per device): every plan holds the whole front, panel q is factorized by the plan
// with q % nparts == part and travels to the others, every plan updates the 32-column blocks of the panels it owns.
// The arithmetic of a column block does not depend on who runs it: the bits are those of the unshared front (without the
// pair update).  Unit of exchange in the reference: the contribution block, SparseQR_factorize.c:1228; here, inside one
// front, the factorized panel. ----
int stmmqr_plan_group_steps(stmmqr_plan *plan, int group)
{
    if (!plan || group < 0 || group >= (int)plan->gsteps.size()) { fail(STMMQR_ERR_INVALID, "no such front group"); return -1; }
    return (int)plan->gsteps[(size_t)group].size();
}

int stmmqr_factorize_step(stmmqr_plan *plan, int group, int step, int what, int cb_first, int cb_stride, int cb_count)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    if (cb_stride < 1) return fail(STMMQR_ERR_INVALID, "column-block stride < 1");
    StepReq rq = {step, what, cb_first, cb_stride, cb_count};
    P.first_group = false;
    return run_schedule(P, false, group, &rq);
}

namespace {
struct PanelMsg { long long f_off, t_off, tau_off, stair_off, dead_off, num_off, total; long nb; };
// layout of one panel message (doubles): the panel's columns of F (whole leading dimension), T of the panel's slot, then
// the front's Tau / Stair / Rdead ranges and its FrontNum -- small next to the columns, and sending the whole ranges keeps
// every plan's copy of them identical without tracking which entries a panel touched
PanelMsg panel_msg(const FrontSym &s, long p)
{
    PanelMsg m;
    const long k1 = p * STM_NB;
    m.nb = std::max(0L, std::min((long)STM_NB, (long)s.fn - k1));
    m.f_off = 0;
    m.t_off = m.f_off + (long long)s.ld * STM_NB;
    m.tau_off = m.t_off + STM_NB * STM_NB;
    m.stair_off = m.tau_off + s.fn;
    m.dead_off = m.stair_off + (s.fn + 1) / 2;
    m.num_off = m.dead_off + (s.fp + 7) / 8;
    m.total = m.num_off + (long long)((sizeof(FrontNum) + 7) / 8);
    return m;
}
}  // namespace

int stmmqr_plan_panel_doubles(stmmqr_plan *plan, stm_long f, stm_long *ndoubles)
{
    if (!plan || f < 0 || f >= plan->nf || !ndoubles) return fail(STMMQR_ERR_INVALID, "bad front");
    *ndoubles = (stm_long)panel_msg(plan->fs[f], 0).total;
    return 0;
}

static int panel_copy(stmmqr_plan &P, stm_long f, stm_long p, double *buf, int on_device, bool out, bool nosync = false)
{
    if (f < 0 || f >= P.nf || !buf) return fail(STMMQR_ERR_INVALID, "bad front / buffer");
    const FrontSym &s = P.fs[f];
    if (p < 0 || p >= s.npanels) return fail(STMMQR_ERR_INVALID, "no such panel");
    if (P.group[f] < 0) return fail(STMMQR_ERR_INVALID, "the front is not factorized by this plan");
    if (!P.begun || !P.d_F.p || P.d_F.n != (size_t)P.farena)
        return fail(STMMQR_ERR_INVALID, "panel messages move between factorize_begin and factorize_finish (the front arena of the current grouping must exist)");
    HIPCHK(hipSetDevice(P.device));
    const PanelMsg m = panel_msg(s, p);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : (out ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
    hipStream_t st = P.stream;
    if (on_device) {
        // device buffer: ONE launch packs / unpacks the six ranges
        void *homes[6] = {P.d_F.p + s.foff + (long long)p * STM_NB * s.ld,
                          P.d_T.p + (long long)STM_TSLOT(P.h_tslot[(size_t)f], (int)p) * STM_NB * STM_NB, P.d_Tau.p + s.rp, P.d_Stair.p + s.rp,
                          P.d_Rdead.p + s.col1, P.d_fnum.p + f};
        const long long offs[6] = {m.f_off * 8, m.t_off * 8, m.tau_off * 8, m.stair_off * 8, m.dead_off * 8, m.num_off * 8};
        const long long bytes[6] = {(long long)s.ld * m.nb * 8, 8LL * STM_NB * STM_NB, 8LL * s.fn, 4LL * s.fn, (long long)s.fp, (long long)sizeof(FrontNum)};
        LCHK(stm_launch_panel_msg(homes, offs, bytes, buf, out ? 1 : 0, st));
        if (out && !nosync) HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    auto cp = [&](void *dev, long long off, size_t bytes) -> int {
        if (!bytes) return 0;
        if (out) HIPCHK(hipMemcpyAsync(buf + off, dev, bytes, kind, st));
        else HIPCHK(hipMemcpyAsync(dev, buf + off, bytes, kind, st));
        return 0;
    };
    LCHK(cp(P.d_F.p + s.foff + (long long)p * STM_NB * s.ld, m.f_off, (size_t)s.ld * (size_t)m.nb * sizeof(double)));
    LCHK(cp(P.d_T.p + (long long)STM_TSLOT(P.h_tslot[(size_t)f], (int)p) * STM_NB * STM_NB, m.t_off, sizeof(double) * STM_NB * STM_NB));
    LCHK(cp(P.d_Tau.p + s.rp, m.tau_off, (size_t)s.fn * sizeof(double)));
    LCHK(cp(P.d_Stair.p + s.rp, m.stair_off, (size_t)s.fn * sizeof(int)));
    LCHK(cp(P.d_Rdead.p + s.col1, m.dead_off, (size_t)s.fp));
    LCHK(cp(P.d_fnum.p + f, m.num_off, sizeof(FrontNum)));
    if ((out || !on_device) && !nosync) HIPCHK(hipStreamSynchronize(st));   // the caller sends the buffer / reuses its host memory
    return 0;
}

int stmmqr_plan_export_panel(stmmqr_plan *plan, stm_long f, stm_long p, double *buf, int on_device)
{
    if (!plan) return fail(STMMQR_ERR_INVALID, "null plan");
    return panel_copy(*plan, f, p, buf, on_device, true);
}

int stmmqr_plan_import_panel(stmmqr_plan *plan, stm_long f, stm_long p, const double *buf, int on_device)
{
    if (!plan) return fail(STMMQR_ERR_INVALID, "null plan");
    return panel_copy(*plan, f, p, const_cast<double *>(buf), on_device, false);
}

// The packed contribution block of a shared front is complete only in the columns of the panels a plan owns: the runs of
// owned columns are contiguous in the packed block (column j of C = column fp + j of the front, SparseQR_factorize.c:1228).
static int front_cols_copy(stmmqr_plan &P, stm_long f, int part, int nparts, double *buf, int on_device, stm_long *ndoubles, bool out)
{
    if (f < 0 || f >= P.nf || nparts < 1 || part < 0 || part >= nparts) return fail(STMMQR_ERR_INVALID, "bad front / part");
    HIPCHK(hipSetDevice(P.device));
    if (int e = check_c_slot(P, f, 0, out ? "stmmqr_plan_export_front_cols" : "stmmqr_plan_import_front_cols")) return e;
    HIPCHK(hipStreamSynchronize(P.stream));
    FrontNum nm;
    HIPCHK(hipMemcpy(&nm, P.d_fnum.p + f, sizeof nm, hipMemcpyDeviceToHost));
    const FrontSym &s = P.fs[f];
    const long cn = s.fn - s.fp, cm = nm.cm;
    if (cm < 0 || cm > cn || (long long)cm * (cm + 1) / 2 + (long long)cm * (cn - cm) > P.c_slot[(size_t)f])
        return fail(STMMQR_ERR_INVALID, "front_cols: the front's contribution block does not fit its slot");
    auto coff = [&](long j) -> long long { return j < cm ? (long long)j * (j + 1) / 2 : (long long)cm * (cm + 1) / 2 + (long long)(j - cm) * cm; };
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : (out ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
    long long pos = 0;
    for (long q = s.fp / STM_NB; q * STM_NB < s.fn; q++) {
        if (q % nparts != part) continue;
        const long j0 = std::max(0L, q * STM_NB - (long)s.fp), j1 = std::min(cn, (q + 1) * STM_NB - (long)s.fp);
        if (j1 <= j0 || cm <= 0) continue;
        const long long a = coff(j0), b = coff(j1);
        if (buf && b > a) {
            if (out) HIPCHK(hipMemcpyAsync(buf + pos, P.d_C.p + s.coff + a, (size_t)(b - a) * sizeof(double), kind, P.stream));
            else HIPCHK(hipMemcpyAsync(P.d_C.p + s.coff + a, buf + pos, (size_t)(b - a) * sizeof(double), kind, P.stream));
        }
        pos += b - a;
    }
    HIPCHK(hipStreamSynchronize(P.stream));
    if (ndoubles) *ndoubles = (stm_long)pos;
    return 0;
}

int stmmqr_plan_export_front_cols(stmmqr_plan *plan, stm_long f, int part, int nparts, double *buf, int on_device, stm_long *ndoubles)
{
    if (!plan) return fail(STMMQR_ERR_INVALID, "null plan");
    return front_cols_copy(*plan, f, part, nparts, buf, on_device, ndoubles, true);
}

int stmmqr_plan_import_front_cols(stmmqr_plan *plan, stm_long f, int part, int nparts, const double *buf, int on_device)
{
    if (!plan || !buf) return fail(STMMQR_ERR_INVALID, "null plan / buffer");
    return front_cols_copy(*plan, f, part, nparts, const_cast<double *>(buf), on_device, nullptr, false);
}

// ---- the panel loop of a SHARED front, native (round 4; sharded.run_shared_front is its Python twin and stays the CPU-testable
// form).  One call per rank and shared front: the rank's place i in the group of R ranks [first_rank, first_rank + R), panel q
// belongs to place q mod R.  Everything is enqueued -- compute on the plan's stream, messages on a comm stream of the plan, the
// two ordered by events -- and the host returns without waiting: no per-step stream synchronisation, no interpreter between two
// steps (the Python loop cost 55-74 us of host time per step).  The owner's send of panel t runs beside the rest of
// its update of step t - 1; a receiver posts the receive of panel t before it starts its update of step t - 1.
// The transport is a table of callbacks (stmmqr_transport): RCCL point-to-point (stmmqr_rccl_transport_create) on a node with
// several GPUs; tests play the ranks on one GPU with a transport of their own. ----
namespace {
struct SharedRing {
    stm_long f = -1;
    int R = 0;
    long long nd = 0;
    std::vector<double *> buf;                 // R device buffers of one panel message each
    std::vector<hipEvent_t> ev_free;           // slot q mod R: its last import / send has finished
    hipEvent_t ev_exp = nullptr, ev_rcv = nullptr;
};
std::mutex g_ring_mu;
std::vector<std::pair<stmmqr_plan *, SharedRing *>> g_rings;
hipStream_t g_comm_stream[64] = {};

SharedRing *ring_for(stmmqr_plan &P, stm_long f, int R)
{
    std::lock_guard<std::mutex> lock(g_ring_mu);
    for (auto &pr : g_rings)
        if (pr.first == &P && pr.second->f == f && pr.second->R == R) return pr.second;
    SharedRing *r = new SharedRing();
    r->f = f; r->R = R;
    r->nd = panel_msg(P.fs[f], 0).total;
    r->buf.assign((size_t)R, nullptr);
    r->ev_free.assign((size_t)R, nullptr);
    bool ok = true;
    for (int q = 0; q < R && ok; q++) {
        ok = hipMalloc((void **)&r->buf[(size_t)q], (size_t)r->nd * sizeof(double)) == hipSuccess &&
             hipEventCreateWithFlags(&r->ev_free[(size_t)q], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipEventCreateWithFlags(&r->ev_exp, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&r->ev_rcv, hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete r; return nullptr; }              // (buffers of a failed attempt are few and are left to process exit)
    g_rings.push_back({&P, r});
    return r;
}
}  // namespace

void stmmqr_plan_release_rings(stmmqr_plan *plan)
{
    std::lock_guard<std::mutex> lock(g_ring_mu);
    for (size_t i = 0; i < g_rings.size();) {
        if (g_rings[i].first == plan) {
            SharedRing *r = g_rings[i].second;
            for (double *b : r->buf) if (b) (void)hipFree(b);
            for (hipEvent_t e : r->ev_free) if (e) (void)hipEventDestroy(e);
            if (r->ev_exp) (void)hipEventDestroy(r->ev_exp);
            if (r->ev_rcv) (void)hipEventDestroy(r->ev_rcv);
            delete r;
            g_rings.erase(g_rings.begin() + (long)i);
        } else i++;
    }
}

int stmmqr_factorize_shared_front(stmmqr_plan *plan, int group, stm_long f, int first_rank, int nranks, const stmmqr_transport *tr)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    if (!tr || !tr->send || !tr->recv) return fail(STMMQR_ERR_INVALID, "null transport");
    stmmqr_plan &P = *plan;
    if (f < 0 || f >= P.nf || (size_t)f >= P.shared.size() || !P.shared[(size_t)f] || P.group[f] != group)
        return fail(STMMQR_ERR_INVALID, "not a shared front of this group (stmmqr_plan_set_groups: STMMQR_GROUP_SHARED)");
    const int R = nranks, i = tr->rank - first_rank;
    if (R < 1 || i < 0 || i >= R) return fail(STMMQR_ERR_INVALID, "this rank is not in the front's group");
    HIPCHK(hipSetDevice(P.device));
    if (P.device < 0 || P.device >= 64) return fail(STMMQR_ERR_INVALID, "device index out of range");
    if (!g_comm_stream[P.device]) HIPCHK(hipStreamCreateWithFlags(&g_comm_stream[P.device], hipStreamNonBlocking));
    hipStream_t cs = g_comm_stream[P.device], st = P.stream;
    const int nsteps = (int)P.gsteps[(size_t)group].size();
    SharedRing *ring = ring_for(P, f, R);
    if (!ring) return fail(STMMQR_ERR_OUT_OF_MEMORY, "panel message buffers");
    const size_t bytes = (size_t)ring->nd * sizeof(double);
    auto step = [&](int t, int what, int cb_first, int cb_stride, int cb_count) -> int {
        StepReq rq = {t, what, cb_first, cb_stride, cb_count};
        P.first_group = false;
        return run_schedule(P, false, group, &rq);
    };
    auto mod = [&](int a) { return ((a % R) + R) % R; };
    int e = step(0, STMMQR_STEP_PREP, 0, 1, -1);
    for (int t = 0; t < nsteps && !e; t++) {
        const int o = t % R, first = mod(i - t);               // my first column block of step t - 1
        double *buf = ring->buf[(size_t)o];
        if (i == o) {
            if (t > 0) e = step(t - 1, STMMQR_STEP_UPDATE | STMMQR_STEP_GRAM, first, R, 1);      // block 0: the columns of my panel
            if (!e) e = step(t, STMMQR_STEP_PANEL, 0, 1, -1);
            if (e) break;
            HIPCHK(hipStreamWaitEvent(st, ring->ev_free[(size_t)o], 0));                       // (the sends of panel t - R are out)
            if ((e = panel_copy(P, f, t, buf, 1, true, true))) break;
            HIPCHK(hipEventRecord(ring->ev_exp, st));
            HIPCHK(hipStreamWaitEvent(cs, ring->ev_exp, 0));
            if (tr->group_begin && tr->group_begin(tr->ctx)) { e = fail(STMMQR_ERR_DEVICE, "transport: group begin"); break; }
            for (int j = 0; j < R && !e; j++)
                if (j != i && tr->send(tr->ctx, buf, bytes, first_rank + j, (void *)cs)) e = fail(STMMQR_ERR_DEVICE, "transport: send of a panel failed");
            if (tr->group_end && tr->group_end(tr->ctx) && !e) e = fail(STMMQR_ERR_DEVICE, "transport: group end");
            if (e) break;
            HIPCHK(hipEventRecord(ring->ev_free[(size_t)o], cs));
            if (t > 0) e = step(t - 1, STMMQR_STEP_UPDATE, first + R, R, -1);                   // the rest, beside the sends
        } else {
            // the receive is posted first (its slot is free once the import of panel t - R has run), then the update of step t - 1
            HIPCHK(hipStreamWaitEvent(cs, ring->ev_free[(size_t)o], 0));
            if (tr->group_begin && tr->group_begin(tr->ctx)) { e = fail(STMMQR_ERR_DEVICE, "transport: group begin"); break; }
            if (tr->recv(tr->ctx, buf, bytes, first_rank + o, (void *)cs)) e = fail(STMMQR_ERR_DEVICE, "transport: receive of a panel failed");
            if (tr->group_end && tr->group_end(tr->ctx) && !e) e = fail(STMMQR_ERR_DEVICE, "transport: group end");
            if (e) break;
            HIPCHK(hipEventRecord(ring->ev_rcv, cs));
            if (t > 0) e = step(t - 1, STMMQR_STEP_UPDATE | STMMQR_STEP_GRAM, first, R, -1);
            if (e) break;
            HIPCHK(hipStreamWaitEvent(st, ring->ev_rcv, 0));
            if ((e = panel_copy(P, f, t, buf, 1, false, true))) break;
            HIPCHK(hipEventRecord(ring->ev_free[(size_t)o], st));
        }
    }
    if (!e) e = step(nsteps - 1, STMMQR_STEP_UPDATE | STMMQR_STEP_GRAM, mod(i - nsteps), R, -1);
    if (!e) e = step(nsteps - 1, STMMQR_STEP_POST, 0, 1, -1);
    // the caller gathers the contribution block (stmmqr_plan_export_front_cols) and goes on with the next phase: it needs the
    // device to have finished this one -- ONE synchronisation per shared front, not one per step
    HIPCHK(hipStreamSynchronize(cs));
    HIPCHK(hipStreamSynchronize(st));
    return e;
}

// ---- RCCL point-to-point transport (ncclSend / ncclRecv on the comm stream).  librccl is loaded at run time -- the library
// has no link-time dependency on it: a one-GPU process never needs it, and under PyTorch the copy that torch.distributed has
// already loaded is the one that is found. ----
namespace {
struct NcclId { char internal[128]; };
typedef void *ncclComm_p;
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(ncclComm_p *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_p) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    int (*GroupStart)(void) = nullptr;
    int (*GroupEnd)(void) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} g_rccl;
int load_rccl()
{
    if (g_rccl.h) return 0;
    const char *names[] = {getenv("STMMQR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if (n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(STMMQR_ERR_DEVICE, std::string("librccl not found: ") + (dlerror() ? dlerror() : ""));
    RcclApi a;
    a.h = h;
    a.GetUniqueId = (int (*)(NcclId *))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(ncclComm_p *, int, NcclId, int))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (int (*)(ncclComm_p))dlsym(h, "ncclCommDestroy");
    a.Send = (int (*)(const void *, size_t, int, int, ncclComm_p, hipStream_t))dlsym(h, "ncclSend");
    a.Recv = (int (*)(void *, size_t, int, int, ncclComm_p, hipStream_t))dlsym(h, "ncclRecv");
    a.GroupStart = (int (*)(void))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (int (*)(void))dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.Send || !a.Recv || !a.GroupStart || !a.GroupEnd)
        return fail(STMMQR_ERR_DEVICE, "librccl lacks a point-to-point entry point");
    g_rccl = a;
    return 0;
}
struct RcclTransport { stmmqr_transport pub; ncclComm_p comm = nullptr; };
int rccl_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream)
{
    return g_rccl.Send(buf, bytes, 0 /* ncclInt8 */, peer, ((RcclTransport *)ctx)->comm, (hipStream_t)stream);
}
int rccl_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream)
{
    return g_rccl.Recv(buf, bytes, 0 /* ncclInt8 */, peer, ((RcclTransport *)ctx)->comm, (hipStream_t)stream);
}
int rccl_gbegin(void *) { return g_rccl.GroupStart(); }
int rccl_gend(void *) { return g_rccl.GroupEnd(); }
}  // namespace

int stmmqr_rccl_unique_id(char id[128])
{
    if (!id) return fail(STMMQR_ERR_INVALID, "null id");
    if (int e = load_rccl()) return e;
    NcclId u;
    memset(&u, 0, sizeof u);
    const int rc = g_rccl.GetUniqueId(&u);
    if (rc) return fail(STMMQR_ERR_DEVICE, std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
    memcpy(id, u.internal, 128);
    return 0;
}

int stmmqr_rccl_transport_create(int world, int rank, const char id[128], stmmqr_transport **out)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(STMMQR_ERR_INVALID, "bad transport arguments");
    *out = nullptr;
    if (int e = load_rccl()) return e;
    RcclTransport *t = new (std::nothrow) RcclTransport();
    if (!t) return fail(STMMQR_ERR_OUT_OF_MEMORY, "host allocation failed");
    NcclId u;
    memcpy(u.internal, id, 128);
    const int rc = g_rccl.CommInitRank(&t->comm, world, u, rank);
    if (rc) { delete t; return fail(STMMQR_ERR_DEVICE, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error")); }
    t->pub.ctx = t; t->pub.send = rccl_send; t->pub.recv = rccl_recv; t->pub.group_begin = rccl_gbegin; t->pub.group_end = rccl_gend;
    t->pub.rank = rank; t->pub.size = world;
    *out = &t->pub;
    return 0;
}

void stmmqr_rccl_transport_destroy(stmmqr_transport *tr)
{
    if (!tr) return;
    RcclTransport *t = (RcclTransport *)tr->ctx;
    if (t && t->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(t->comm);
    delete t;
}

/* a transport's own send / receive of one device buffer, ordered on `stream` (a hipStream_t): what the tests and the Python side
 * use to move a buffer with the same object the native loop uses */
int stmmqr_transport_sendrecv(const stmmqr_transport *tr, const void *sendbuf, size_t sendbytes, int dst, void *recvbuf, size_t recvbytes,
                              int src, void *stream)
{
    if (!tr) return fail(STMMQR_ERR_INVALID, "null transport");
    int e = 0;
    if (tr->group_begin) e |= tr->group_begin(tr->ctx);
    if (sendbuf && sendbytes && dst >= 0) e |= tr->send(tr->ctx, sendbuf, sendbytes, dst, stream);
    if (recvbuf && recvbytes && src >= 0) e |= tr->recv(tr->ctx, recvbuf, recvbytes, src, stream);
    if (tr->group_end) e |= tr->group_end(tr->ctx);
    return e ? fail(STMMQR_ERR_DEVICE, "transport: send / receive failed") : 0;
}

/* off[0..fn]: where each column of front f starts inside its packed R+H block (off[fn] = the block's size) */
int stmmqr_plan_front_rhoff(stmmqr_plan *plan, stm_long f, stm_long *off)
{
    if (!plan || !plan->factored || f < 0 || f >= plan->nf || !off) return fail(STMMQR_ERR_INVALID, "bad front / no factorization held");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    const FrontSym &s = P.fs[f];
    std::vector<long long> h((size_t)std::max(1, s.fn));
    if (s.fn > 0) HIPCHK(hipMemcpy(h.data(), P.d_Rhoff.p + s.rp, (size_t)s.fn * sizeof(long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < s.fn; k++) off[k] = (stm_long)h[(size_t)k];
    off[s.fn] = (stm_long)P.h_fnum[(size_t)f].rsize;
    return 0;
}

/* device memory held by the plan right now (bytes): arenas, factors, workspaces, index arrays */
double stmmqr_plan_device_bytes(const stmmqr_plan *plan) { return plan ? plan->device_bytes() : 0.0; }

/* out[0..1] = flops, flops of the trailing updates of front f (read from the device: valid once its panels are done) */
int stmmqr_plan_front_flops(stmmqr_plan *plan, stm_long f, double *out)
{
    if (!plan || f < 0 || f >= plan->nf || !out) return fail(STMMQR_ERR_INVALID, "bad front");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    HIPCHK(hipStreamSynchronize(P.stream));
    FrontNum nm;
    HIPCHK(hipMemcpy(&nm, P.d_fnum.p + f, sizeof nm, hipMemcpyDeviceToHost));
    out[0] = nm.flops; out[1] = nm.flops_upd;
    return 0;
}

int stmmqr_plan_result_sizes(const stmmqr_plan *plan, stm_long *rh_total, stm_long *rank)
{
    if (!plan || !plan->factored) return fail(STMMQR_ERR_INVALID, "no factorization held by the plan");
    if (rh_total) *rh_total = (stm_long)plan->rh_total;
    if (rank) *rank = plan->rank;
    return 0;
}

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
                    d.f = f; d.xoff = (int)xo; d.dqoff = (int)dqo; d.wqoff = (int)wo; d.nslab = (s.fm_ub + STM_QB_ROWS - 1) / STM_QB_ROWS; d.pad = 0;
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
                LCHK(stm_launch_qapply_big4(c, P.d_qb.p + Q.off, P.d_qbt4off.p + Q.off, Q.n, Q.max_np, Q.max_nslab, Q.max_fm, m, P.d_W.p, P.d_Xf.p,
                                            P.d_Dq.p, P.d_Wq4.p, P.d_T4.p, P.stream, nb, B));
                return 0;
            }
            LCHK(stm_launch_qapply_big(c, P.d_qb.p + Q.off, Q.n, Q.max_np, Q.max_nslab, Q.max_fm, m, P.d_W.p, P.d_Xf.p, P.d_Dq.p,
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
        LCHK(stm_launch_rsolve_big(c, P.d_qb.p + Q.off, Q.n, Q.max_rsteps, Q.max_nslab, P.d_Rj.p, P.d_W.p, P.d_Xs.p, P.d_Xf.p,
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

int stmmqr_plan_download(stmmqr_plan *plan, double *Stack, stm_long *Rblock_off, char *Rdead, stm_long *HStair,
                         double *HTau, stm_long *Hii, stm_long *HPinv, stm_long *Hm, stm_long *Hr,
                         stm_long *scalars, stmmqr_stats *stats)
{
    if (!plan || !plan->factored) return fail(STMMQR_ERR_INVALID, "no factorization held by the plan");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    hipStream_t st = P.stream;
    const long nf = P.nf, m = P.m, n = P.n;
    HIPCHK(hipEventRecord(P.ev[6], st));
    if (Stack && P.rh_total > 0 && P.recycle) {
        // slab recycling: the blocks lie in the arena in the order the fronts finished (kept fronts: still in front form); the
        // reference's layout is produced window by window in a bounce buffer (k_rh_window) and copied out, two windows in flight
        const long long win = std::min<long long>(P.rh_total, 32LL << 20);            // 256 MB windows
        if (P.d_bounce.n < (size_t)(2 * win)) LCHK(P.d_bounce.alloc((size_t)(2 * win)));
        const DevCtx c = P.ctx();
        const int parts = (int)std::min<long long>(64, std::max<long long>(1, win / (64 * 1024)));
        hipEvent_t evw[2] = {nullptr, nullptr};
        for (auto &e : evw) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        int rc = 0;
        for (long long w0 = 0, i = 0; w0 < P.rh_total && !rc; w0 += win, i++) {
            const long long w1 = std::min(P.rh_total, w0 + win);
            double *out = P.d_bounce.p + (i & 1) * win;
            if (i >= 2 && hipEventSynchronize(evw[i & 1]) != hipSuccess) rc = 1;       // (the copy that last read this half is done)
            if (!rc && stm_launch_rh_window(c, P.d_lists.p + P.own_off, P.n_own, parts, P.d_fin.p, P.d_kept.p, P.d_RH.p, w0, w1, out, st)) rc = 1;
            if (!rc && hipMemcpyAsync(Stack + w0, out, (size_t)(w1 - w0) * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) rc = 1;
            if (!rc && hipEventRecord(evw[i & 1], st) != hipSuccess) rc = 1;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = 1;
        for (auto &e : evw) (void)hipEventDestroy(e);
        if (rc) return fail(STMMQR_ERR_DEVICE, "download of the packed factors failed");
    } else if (Stack && P.rh_total > 0)
        HIPCHK(hipMemcpyAsync(Stack, P.d_RH.p, (size_t)P.rh_total * sizeof(double), hipMemcpyDeviceToHost, st));
    std::vector<int> stair32((size_t)std::max(1L, P.rjsize)), hii32((size_t)std::max(1L, P.hisize));
    std::vector<long long> rboff((size_t)std::max(1L, nf));
    if (P.rjsize > 0)
        HIPCHK(hipMemcpyAsync(stair32.data(), P.d_Stair.p, (size_t)P.rjsize * sizeof(int), hipMemcpyDeviceToHost, st));
    if (P.hisize > 0)
        HIPCHK(hipMemcpyAsync(hii32.data(), P.d_Hii.p, (size_t)P.hisize * sizeof(int), hipMemcpyDeviceToHost, st));
    if (HTau && P.rjsize > 0)
        HIPCHK(hipMemcpyAsync(HTau, P.d_Tau.p, (size_t)P.rjsize * sizeof(double), hipMemcpyDeviceToHost, st));
    if (Rdead && n > 0) HIPCHK(hipMemcpyAsync(Rdead, P.d_Rdead.p, (size_t)n, hipMemcpyDeviceToHost, st));
    if (nf > 0)
        HIPCHK(hipMemcpyAsync(rboff.data(), P.recycle ? P.d_fin.p : P.d_Rboff.p, (size_t)nf * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(P.ev[7], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, P.ev[6], P.ev[7]));
    P.stats.ms_d2h = ms;

    const double t0 = now_ms();
    if (HStair) for (long i = 0; i < P.rjsize; i++) HStair[i] = stair32[i];
    if (Rblock_off) for (long f = 0; f < nf; f++) Rblock_off[f] = (stm_long)rboff[f];
    long maxfrank = 1, maxfm = 0, rank = 0;
    bool all_here = true;
    for (long f = 0; f < nf; f++) {
        const FrontNum &nm = P.h_fnum[f];
        const bool own = P.group[f] >= 0;
        all_here = all_here && own;
        if (Hm) Hm[f] = own ? nm.fm : 0;
        if (Hr) Hr[f] = own ? nm.rank : 0;
        if (!own) continue;
        maxfrank = std::max(maxfrank, (long)nm.rank);
        maxfm = std::max(maxfm, (long)nm.fm);
        rank += nm.rank;
    }
    if (!all_here) {
        // sharded run: the caller merges the shards and runs qr_hpinv on the union; Hii stays in S-row ids
        if (Hii)
            for (long f = 0; f < nf; f++)
                if (P.group[f] >= 0)
                    for (long i = 0; i < P.h_fnum[f].fm; i++) Hii[P.Hip[f] + i] = hii32[P.Hip[f] + i];
        Hii = nullptr; HPinv = nullptr;
    }
    // qr_hpinv (SparseQR_factorize.c:991-1060): global row permutation, Hii rewritten in place
    if (Hii || HPinv) {
        std::vector<long> W((size_t)std::max(1L, m), 0);
        long row1 = 0, row2 = m;
        for (long i = P.Sleft[n]; i < m; i++) W[i] = --row2;
        for (long f = 0; f < nf; f++) {
            const int *Hi = hii32.data() + P.Hip[f];
            const FrontNum &nm = P.h_fnum[f];
            const long rm = nm.rank, fm = nm.fm;
            for (long i = 0; i < rm; i++) W[Hi[i]] = row1++;
            const long cn = P.fs[f].fn - P.fs[f].fp;
            const long cm = std::min(fm - rm, cn);
            for (long i = fm - 1; i >= rm + cm; i--) W[Hi[i]] = --row2;
        }
        if (HPinv) for (long i = 0; i < m; i++) HPinv[i] = W[P.PLinv[i]];
        if (Hii) {
            for (long f = 0; f < nf; f++) {
                const int *Hi = hii32.data() + P.Hip[f];
                stm_long *Ho = Hii + P.Hip[f];
                const long fm = P.h_fnum[f].fm;
                for (long i = 0; i < fm; i++) Ho[i] = W[Hi[i]];
            }
        }
    }
    if (scalars) {
        scalars[0] = rank;
        long rank1 = rank;
        if (P.last_ntol < n) {
            std::vector<char> rd((size_t)std::max(1L, n));
            if (Rdead) memcpy(rd.data(), Rdead, (size_t)n);
            else HIPCHK(hipMemcpy(rd.data(), P.d_Rdead.p, (size_t)n, hipMemcpyDeviceToHost));
            rank1 = 0;
            for (long j = 0; j < P.last_ntol; j++) rank1 += !rd[j];
        }
        scalars[1] = rank1; scalars[2] = maxfrank; scalars[3] = maxfm;
    }
    P.stats.ms_host += now_ms() - t0;
    if (stats) *stats = P.stats;
    return 0;
}

int stmmqr_factorize_arrays(const stmmqr_symbolic_view *sym, const stm_long *Ap, const stm_long *Ai,
                            const double *Ax, double tol, stm_long ntol, double *Stack, stm_long stack_cap,
                            stm_long *Rblock_off, char *Rdead, stm_long *HStair, double *HTau, stm_long *Hii,
                            stm_long *HPinv, stm_long *Hm, stm_long *Hr, stm_long *scalars, stmmqr_stats *stats)
{
    int st = 0;
    stmmqr_plan *P = stmmqr_plan_create(sym, -1, &st);
    if (!P) return st;
    st = stmmqr_factorize_device(P, Ap, Ai, Ax, 0, tol, ntol, stats);
    if (!st && Stack && P->rh_total > stack_cap) st = fail(STMMQR_ERR_INVALID, "Stack buffer too small");
    if (!st) st = stmmqr_plan_download(P, Stack, Rblock_off, Rdead, HStair, HTau, Hii, HPinv, Hm, Hr, scalars, stats);
    stmmqr_plan_destroy(P);
    return st;
}

// -------------------------------------------------------------------------------------------------
// drop-in seam: qr_factorize with the reference's structs
// -------------------------------------------------------------------------------------------------
static inline int &cc_int(stm_sparse_common *cc, size_t off) { return *(int *)((char *)cc + off); }
static inline size_t &cc_size(stm_sparse_common *cc, size_t off) { return *(size_t *)((char *)cc + off); }
static inline double &cc_dbl(stm_sparse_common *cc, size_t off) { return *(double *)((char *)cc + off); }

// (stmmqr_internal.h: shared with stmmqr_seams.cpp / stmmqr_symbolic.cpp; local to the library)
int stm_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }
void stm_cc_set_status(stm_sparse_common *cc, int code) { if (cc) cc_int(cc, g_layout.status) = code; }
static void *cc_malloc(size_t n, size_t size, stm_sparse_common *cc, bool zero = false);
static void cc_free(size_t n, size_t size, void *p, stm_sparse_common *cc);
void *stm_cc_malloc(size_t n, size_t size, stm_sparse_common *cc) { return cc_malloc(n, size, cc, false); }
void stm_cc_free(size_t n, size_t size, void *p, stm_sparse_common *cc) { cc_free(n, size, p, cc); }
// ... and exported for libstmmqr_hip_api.so (csrc/stmmqr_api.cpp): what it returns is released by the reference's own
// SparseCore_free_dense / SparseCore_free and must be counted the same way
void *stmmqr_cc_malloc(size_t n, size_t size, stm_sparse_common *cc) { return cc_malloc(n, size, cc, false); }
void stmmqr_cc_free(size_t n, size_t size, void *p, stm_sparse_common *cc) { cc_free(n, size, p, cc); }
void stmmqr_cc_set_status(stm_sparse_common *cc, int code) { stm_cc_set_status(cc, code); }

// SparseCore_malloc semantics (src/core/SparseCore_common.c:603-655): malloc(max(1,n)*size) + counters
static void *cc_malloc(size_t n, size_t size, stm_sparse_common *cc, bool zero)
{
    void *p = zero ? calloc(std::max<size_t>(1, n), size) : malloc(std::max<size_t>(1, n) * size);
    if (!p) {
        if (cc) cc_int(cc, g_layout.status) = STMMQR_ERR_OUT_OF_MEMORY;
        return nullptr;
    }
    if (cc) {
        cc_size(cc, g_layout.malloc_count)++;
        cc_size(cc, g_layout.memory_inuse) += n * size;
        cc_size(cc, g_layout.memory_usage) =
            std::max(cc_size(cc, g_layout.memory_usage), cc_size(cc, g_layout.memory_inuse));
    }
    return p;
}
static void cc_free(size_t n, size_t size, void *p, stm_sparse_common *cc)
{
    if (!p) return;
    free(p);
    if (cc) {
        cc_size(cc, g_layout.malloc_count)--;
        cc_size(cc, g_layout.memory_inuse) -= n * size;
    }
}
// SparseCore_free_sparse (src/core/SparseCore_matrix_type.c:146-180)
static void cc_free_sparse(stm_sparse_csc **Ah, stm_sparse_common *cc)
{
    if (!Ah || !*Ah) return;
    stm_sparse_csc *A = *Ah;
    cc_free(A->ncol + 1, sizeof(stm_long), A->p, cc);
    cc_free(A->nzmax, sizeof(stm_long), A->i, cc);
    cc_free(A->ncol, sizeof(stm_long), A->nz, cc);
    cc_free(A->nzmax, sizeof(double), A->x, cc);
    cc_free(1, sizeof(stm_sparse_csc), A, cc);
    *Ah = nullptr;
}
static void free_numeric(stm_qr_numeric *N, stm_sparse_common *cc)
{
    if (!N) return;
    cc_free(N->nf, sizeof(double *), N->Rblock, cc);
    cc_free(N->n, 1, N->Rdead, cc);
    cc_free(N->rjsize, sizeof(stm_long), N->HStair, cc);
    cc_free(N->rjsize, sizeof(double), N->HTau, cc);
    cc_free(N->nf, sizeof(stm_long), N->Hm, cc);
    cc_free(N->nf, sizeof(stm_long), N->Hr, cc);
    cc_free(N->hisize, sizeof(stm_long), N->Hii, cc);
    cc_free(N->m, sizeof(stm_long), N->HPinv, cc);
    if (N->Stacks)
        for (stm_long s = 0; s < N->ns; s++)
            cc_free(N->Stack_size ? N->Stack_size[s] : N->maxstack, sizeof(double), N->Stacks[s], cc);
    cc_free(N->ns, sizeof(double *), N->Stacks, cc);
    cc_free(N->ns, sizeof(stm_long), N->Stack_size, cc);
    cc_free(1, sizeof(stm_qr_numeric), N, cc);
}

// ---- plan cache of the drop-in seam ------------------------------------------------------------------------------------
// The reference's driver calls qr_factorize once per SparseQR(); an application that refactorizes (new values, same pattern)
// calls it again with an equal qr_symbolic.  Building the plan (symbolic upload, schedule, workspaces, arena allocation) costs
// about as much as the factorization itself on the BASELINE matrices, so the seam keeps the last plans: the key is a hash of
// everything the plan is derived from (the qr_symbolic's scalars and arrays, the options and the environment knobs read at plan
// time), a second hash of A's pattern tells whether the value map (qr_stranspose2) is still valid.  A cached plan keeps its
// device memory: STMMQR_PLAN_CACHE=0 turns the cache off, STMMQR_PLAN_CACHE=n keeps n plans (default 1),
// stmmqr_plan_cache_clear() / stmmqr_shutdown() release them.
namespace {
inline unsigned long long hash_bytes(const void *p, size_t bytes, unsigned long long h)
{
    const unsigned long long *w = (const unsigned long long *)p;
    const size_t nw = bytes / 8;
    unsigned long long h0 = h, h1 = h ^ 0x9e3779b97f4a7c15ULL, h2 = h + 0x632be59bd9b4e019ULL, h3 = ~h;
    size_t i = 0;
    for (; i + 4 <= nw; i += 4) {                          // four independent lanes: ~8 GB/s on one host core
        h0 = (h0 ^ w[i]) * 0x100000001b3ULL; h0 ^= h0 >> 29;
        h1 = (h1 ^ w[i + 1]) * 0x100000001b3ULL; h1 ^= h1 >> 31;
        h2 = (h2 ^ w[i + 2]) * 0x100000001b3ULL; h2 ^= h2 >> 27;
        h3 = (h3 ^ w[i + 3]) * 0x100000001b3ULL; h3 ^= h3 >> 30;
    }
    for (; i < nw; i++) { h0 = (h0 ^ w[i]) * 0x100000001b3ULL; h0 ^= h0 >> 29; }
    const unsigned char *c = (const unsigned char *)p + nw * 8;
    for (size_t k = 0; k < bytes % 8; k++) h1 = (h1 ^ c[k]) * 0x100000001b3ULL;
    return ((h0 * 31 + h1) * 31 + h2) * 31 + h3;
}
unsigned long long symbolic_key(const stm_qr_symbolic *S)
{
    unsigned long long h = 0xcbf29ce484222325ULL;
    const stm_long sc[] = {S->m, S->n, S->anz, S->nf, S->maxfn, S->rjsize, S->hisize, S->do_rank_detection, S->keepH,
                           (stm_long)(S->Qfill != nullptr), (stm_long)(S->Fm != nullptr), S->maxstack};   // (maxstack sizes the R+H arena)
    h = hash_bytes(sc, sizeof sc, h);
    auto add = [&](const stm_long *a, stm_long cnt) { if (a && cnt > 0) h = hash_bytes(a, (size_t)cnt * sizeof(stm_long), h); };
    add(S->Sp, S->m + 1); add(S->Sj, S->anz); add(S->Qfill, S->n); add(S->PLinv, S->m); add(S->Sleft, S->n + 2);
    add(S->Child, S->nf + 1); add(S->Childp, S->nf + 2); add(S->Super, S->nf + 1); add(S->Rp, S->nf + 1); add(S->Rj, S->rjsize);
    add(S->Post, S->nf); add(S->Hip, S->nf + 1); add(S->Fm, S->nf);
    h = hash_bytes(&g_opt, sizeof g_opt, h);
    // (every knob of the environment that is read when the plan / its schedule / its arenas are built)
    for (const char *k : {"STMMQR_CA_MIN", "STMMQR_PAIR_MIN", "STMMQR_SCHED", "STMMQR_RIDE", "STMMQR_QBIG_MIN", "STMMQR_RECYCLE", "STMMQR_TUNE",
                          "STMMQR_RH_EST_SCALE"}) {
        const char *v = getenv(k);
        if (v) h = hash_bytes(v, strlen(v), h ^ 0x51ed);
    }
    return h;
}
struct CachedPlan { unsigned long long key = 0, pat = 0; stmmqr_plan *plan = nullptr; int device = -1; unsigned long tick = 0; };
std::mutex g_cache_mu;
std::vector<CachedPlan> g_cache;
unsigned long g_cache_tick = 0;
int cache_capacity()
{
    const char *v = getenv("STMMQR_PLAN_CACHE");
    return v ? std::max(0, atoi(v)) : 1;
}
// take a plan for this key out of the cache (nullptr: none); the caller owns it until cache_put
stmmqr_plan *cache_take(unsigned long long key, unsigned long long *pat)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (size_t i = 0; i < g_cache.size(); i++)
        if (g_cache[i].plan && g_cache[i].key == key && g_cache[i].device == dev) {
            stmmqr_plan *P = g_cache[i].plan;
            *pat = g_cache[i].pat;
            g_cache.erase(g_cache.begin() + (long)i);
            return P;
        }
    return nullptr;
}
void cache_put(unsigned long long key, unsigned long long pat, stmmqr_plan *P)
{
    const int cap = cache_capacity();
    std::vector<stmmqr_plan *> drop;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        if (cap <= 0) drop.push_back(P);
        else {
            CachedPlan e; e.key = key; e.pat = pat; e.plan = P; e.device = P->device; e.tick = ++g_cache_tick;
            g_cache.push_back(e);
            while ((int)g_cache.size() > cap) {
                size_t old = 0;
                for (size_t i = 1; i < g_cache.size(); i++) if (g_cache[i].tick < g_cache[old].tick) old = i;
                drop.push_back(g_cache[old].plan);
                g_cache.erase(g_cache.begin() + (long)old);
            }
        }
    }
    for (stmmqr_plan *q : drop) stmmqr_plan_destroy(q);
}
}  // namespace

void stmmqr_plan_cache_clear(void)
{
    std::vector<CachedPlan> old;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        old.swap(g_cache);
    }
    for (auto &e : old) if (e.plan) stmmqr_plan_destroy(e.plan);
}

/* release a qr_numeric returned by qr_factorize (for hosts WITHOUT the reference's qr_freenum; same accounting) */
void stmmqr_free_numeric(stm_qr_numeric **Nh, stm_sparse_common *cc);

// a large result array of the seam: plain malloc (the reference's qr_freenum releases it with free()), but asked to come in
// huge pages and populated NOW by the kernel in one call instead of page fault by page fault under the copy that fills it
static void prefault(void *p, size_t bytes)
{
#ifdef __linux__
    if (!p || bytes < (8u << 20)) return;
    const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
    if (b <= a) return;
#ifdef MADV_HUGEPAGE
    (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
#endif
#ifdef MADV_POPULATE_WRITE
    // (in pieces: one call for a gigabyte holds the address-space lock long enough to stall the thread that launches kernels)
    for (uintptr_t q = a; q < b; q += (uintptr_t)32 << 20)
        (void)madvise((void *)q, std::min<uintptr_t>(b - q, (uintptr_t)32 << 20), MADV_POPULATE_WRITE);
#endif
#endif
}

stm_qr_numeric *qr_factorize(stm_sparse_csc **Ahandle, stm_long freeA, double tol, stm_long ntol,
                             stm_qr_symbolic *S, stm_sparse_common *cc)
{
    const bool timing = getenv("STMMQR_SEAM_TIMING") != nullptr;
    const double t_in = now_ms();
    if (!S) {                                                  // SparseQR_factorize.c:247-254
        if (freeA) cc_free_sparse(Ahandle, cc);
        return nullptr;
    }
    auto set_status = [&](int st) { if (cc) cc_int(cc, g_layout.status) = st; };
    stm_sparse_csc *A = Ahandle ? *Ahandle : nullptr;
    if (!A) { set_status(STMMQR_ERR_INVALID); return nullptr; }
    // A must be the matrix QRsym was made for BEFORE anything walks its arrays with QRsym's sizes (the cache key below hashes
    // A->p over n + 1 and A->i over anz entries: a mismatched pair would be read past its end instead of being refused)
    if ((stm_long)A->nrow != S->m || (stm_long)A->ncol != S->n || !A->p || (S->anz > 0 && (!A->i || !A->x)) ||
        ((const stm_long *)A->p)[S->n] != S->anz || (stm_long)A->nzmax < S->anz) {
        if (freeA) cc_free_sparse(Ahandle, cc);
        set_status(STMMQR_ERR_INVALID);
        return nullptr;
    }

    stmmqr_symbolic_view v;
    v.m = S->m; v.n = S->n; v.anz = S->anz; v.nf = S->nf; v.maxfn = S->maxfn; v.rjsize = S->rjsize;
    v.hisize = S->hisize; v.do_rank_detection = S->do_rank_detection;
    v.Sp = S->Sp; v.Sj = S->Sj; v.Qfill = S->Qfill; v.PLinv = S->PLinv; v.Sleft = S->Sleft;
    v.Child = S->Child; v.Childp = S->Childp; v.Super = S->Super; v.Rp = S->Rp; v.Rj = S->Rj; v.Post = S->Post;
    v.Hip = S->Hip; v.Fm = S->Fm; v.maxstack = S->maxstack;

    int st = 0;
    // the plan: from the cache when an equal qr_symbolic was factorized before (same options), else built now
    unsigned long long key = 0, pat = 0, pat_cached = 0;
    const bool use_cache = cache_capacity() > 0;
    stmmqr_plan *P = nullptr;
    bool cached = false;
    if (use_cache) {
        st = ensure_device(-1);
        if (!st) {
            key = symbolic_key(S);
            pat = hash_bytes(A->p, (size_t)(S->n + 1) * sizeof(stm_long), 0x1234567);
            pat = hash_bytes(A->i, (size_t)std::max<stm_long>(0, S->anz) * sizeof(stm_long), pat);
            P = cache_take(key, &pat_cached);
            cached = P != nullptr;
        }
    }
    if (!st && !P) P = stmmqr_plan_create(&v, -1, &st);
    const double t_plan = now_ms();
    stmmqr_stats stats;
    const bool same_pattern = cached && pat_cached == pat && P->pattern_set;
    // The returned stack (the packed R+H: 1.2 GB for the xenon1 stand-in) is malloc'ed -- the reference's qr_freenum free()s it --
    // and populating its pages costs the host 40 ms per GB: that runs in a helper thread BESIDE the factorization.  Its exact size
    // is only known at the end (it depends on the numerical rank), so the thread takes the size of the last factorization with
    // this plan (+ 2 %) or, the first time, the reference's own first allocation QRsym->maxstack (SparseQR_factorize.c:405-422),
    // and the block is shrunk to the exact size afterwards, as the reference shrinks its stack (:597-663).
    double *early_stack = nullptr;
    size_t early_doubles = 0;
    std::thread early;
    if (!st && !(getenv("STMMQR_SEAM_EARLY_ALLOC") && atoi(getenv("STMMQR_SEAM_EARLY_ALLOC")) == 0)) {
        // (a first call has only QRsym->maxstack to go by -- about twice the packed factors on the BASELINE matrices -- and
        //  populating that much beside the factorization costs more than it saves: the helper runs for cached plans only,
        //  STMMQR_SEAM_EARLY_ALLOC=2 forces it for first calls too)
        const bool force = getenv("STMMQR_SEAM_EARLY_ALLOC") && atoi(getenv("STMMQR_SEAM_EARLY_ALLOC")) == 2;
        early_doubles = (cached && P->rh_total > 0) ? (size_t)((double)P->rh_total * 1.02) + 1024
                                                    : (force ? (size_t)std::max<stm_long>(S->maxstack, 1) : 0);
        if (early_doubles * sizeof(double) >= (64u << 20)) {
            try {                                                  // (nothing may be thrown across the C ABI: no helper, plain allocation later)
                early = std::thread([&early_stack, early_doubles]() {
                    early_stack = (double *)malloc(early_doubles * sizeof(double));
                    prefault(early_stack, early_doubles * sizeof(double));
                });
            } catch (...) {
                early_doubles = 0;
            }
        } else early_doubles = 0;
    }
    if (!st) st = stmmqr_factorize_device(P, same_pattern ? nullptr : (const stm_long *)A->p, same_pattern ? nullptr : (const stm_long *)A->i,
                                          (const double *)A->x, 0, tol, ntol, &stats);
    if (st == STMMQR_ERR_OUT_OF_MEMORY && use_cache && !cached) {
        // The cache keeps the device memory of the plans it holds (3 GB on the xenon1 stand-in, 25 GB on the configs[4] stand-in) after
        // qr_factorize returns; the reference frees everything.  A new matrix that does not fit BESIDE a cached plan must not fail
        // where the reference would succeed: the cache is emptied and the call tried once more.
        bool any;
        { std::lock_guard<std::mutex> lock(g_cache_mu); any = !g_cache.empty(); }
        if (any) {
            if (g_opt.verbose) fprintf(stderr, "[stmmqr_hip] out of device memory beside cached plans: cache emptied, trying again\n");
            if (P) { stmmqr_plan_destroy(P); P = nullptr; }
            stmmqr_plan_cache_clear();
            st = 0;
            P = stmmqr_plan_create(&v, -1, &st);
            if (!st) st = stmmqr_factorize_device(P, (const stm_long *)A->p, (const stm_long *)A->i, (const double *)A->x, 0, tol, ntol, &stats);
        }
    }
    const double t_fact = now_ms();
    if (freeA) cc_free_sparse(Ahandle, cc);                    // :324-327
    if (early.joinable()) early.join();
    if (st) {
        free(early_stack);
        if (P) stmmqr_plan_destroy(P);
        set_status(st);
        return nullptr;
    }
    const stm_long nf = S->nf, n = S->n, m = S->m;
    stm_qr_numeric *N = (stm_qr_numeric *)cc_malloc(1, sizeof(stm_qr_numeric), cc, true);
    if (!N) { free(early_stack); stmmqr_plan_destroy(P); return nullptr; }
    N->n = n; N->m = m; N->nf = nf; N->rjsize = S->rjsize; N->hisize = S->hisize; N->keepH = S->keepH;
    N->maxstack = S->maxstack; N->ns = 1; N->ntasks = 1; N->maxfm = -1; N->norm_E_fro = 0;
    N->Rblock = (double **)cc_malloc(nf, sizeof(double *), cc);
    N->Rdead = (char *)cc_malloc(n, 1, cc, true);
    N->Stacks = (double **)cc_malloc(1, sizeof(double *), cc, true);
    N->Stack_size = (stm_long *)cc_malloc(1, sizeof(stm_long), cc, true);
    N->HStair = (stm_long *)cc_malloc(S->rjsize, sizeof(stm_long), cc);
    N->HTau = (double *)cc_malloc(S->rjsize, sizeof(double), cc);
    N->Hii = (stm_long *)cc_malloc(S->hisize, sizeof(stm_long), cc);
    N->Hm = (stm_long *)cc_malloc(nf, sizeof(stm_long), cc);
    N->Hr = (stm_long *)cc_malloc(nf, sizeof(stm_long), cc);
    N->HPinv = (stm_long *)cc_malloc(m, sizeof(stm_long), cc);
    std::vector<stm_long> roff((size_t)std::max<stm_long>(1, nf));
    stm_long scal[4] = {0, 0, 0, 0};
    bool ok = N->Rblock && N->Rdead && N->Stacks && N->Stack_size && N->HStair && N->HTau && N->Hii && N->Hm &&
              N->Hr && N->HPinv;
    if (ok) {
        // the reference shrinks its stack to exactly the packed R+H (:597-663): allocate that size directly
        N->Stack_size[0] = (stm_long)P->rh_total;
        if (early_stack && (size_t)P->rh_total <= early_doubles) {
            // shrink to the exact size (an mmap'ed block shrinks in place); counted as ONE allocation of that size
            double *q = (double *)realloc(early_stack, std::max<size_t>(1, (size_t)P->rh_total) * sizeof(double));
            N->Stacks[0] = q ? q : early_stack;
            early_stack = nullptr;
            if (cc) {
                cc_size(cc, g_layout.malloc_count)++;
                cc_size(cc, g_layout.memory_inuse) += (size_t)P->rh_total * sizeof(double);
                cc_size(cc, g_layout.memory_usage) = std::max(cc_size(cc, g_layout.memory_usage), cc_size(cc, g_layout.memory_inuse));
            }
        } else {
            free(early_stack);                                  // (too small: more live rows than last time)
            early_stack = nullptr;
            N->Stacks[0] = (double *)cc_malloc((size_t)P->rh_total, sizeof(double), cc);
            if (N->Stacks[0]) prefault(N->Stacks[0], (size_t)P->rh_total * sizeof(double));
        }
        ok = N->Stacks[0] != nullptr;
    }
    free(early_stack);
    const double t_alloc = now_ms();
    const double P_rh_bytes = 8.0 * (double)P->rh_total;
    if (ok) {
        st = stmmqr_plan_download(P, N->Stacks[0], roff.data(), N->Rdead, N->HStair, N->HTau, N->Hii, N->HPinv, N->Hm,
                                  N->Hr, scal, &stats);
        ok = st == 0;
    }
    const double t_down = now_ms();
    if (use_cache && ok) cache_put(key, pat, P);               // (keeps its device memory for the next call with this qr_symbolic)
    else stmmqr_plan_destroy(P);
    if (!ok) {
        free_numeric(N, cc);
        set_status(st ? st : STMMQR_ERR_OUT_OF_MEMORY);
        return nullptr;
    }
    for (stm_long f = 0; f < nf; f++) N->Rblock[f] = N->Stacks[0] + roff[f];
    N->rank = scal[0]; N->rank1 = scal[1]; N->maxfrank = scal[2]; N->maxfm = scal[3];
    if (cc) cc_dbl(cc, g_layout.SPQR_flopcount) = stats.flops;
    if (timing)
        fprintf(stderr, "[stmmqr_hip] qr_factorize seam: plan %s %.1f ms, factorization %.1f ms (device %.1f), host arrays %.1f ms, "
                        "download of %.0f MB %.1f ms, total %.1f ms\n", cached ? (same_pattern ? "cached" : "cached (new pattern)") : "built",
                t_plan - t_in, t_fact - t_plan, stats.ms_total, t_alloc - t_fact, (double)P_rh_bytes * 1e-6, t_down - t_alloc, now_ms() - t_in);
    return N;
}

void stmmqr_free_numeric(stm_qr_numeric **Nh, stm_sparse_common *cc)
{
    if (!Nh || !*Nh) return;
    free_numeric(*Nh, cc);
    *Nh = nullptr;
}

}  // extern "C"
