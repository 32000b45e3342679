// stmmqr_multi.cpp -- multi-GPU support of the numeric phase: contribution blocks and panels in and out of a plan,
// the step interface, the native panel loop of a front SHARED between plans, the RCCL point-to-point transport.
#include "stmmqr_plan.h"
#include <algorithm>

extern "C" {

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
    if ((e = stm_check_c_slot(P, f, info[3], "stmmqr_plan_export_front"))) return e;
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
    if (int e = stm_check_c_slot(P, f, csize, "stmmqr_plan_import_front")) return e;
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
    return stm_run_schedule(P, false, group, &rq);
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
    if (int e = stm_check_c_slot(P, f, 0, out ? "stmmqr_plan_export_front_cols" : "stmmqr_plan_import_front_cols")) return e;
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

static int shared_front_loop(stmmqr_plan *plan, int group, stm_long f, int first_rank, int nranks, const stmmqr_transport *tr, bool host_sync)
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
        return stm_run_schedule(P, false, group, &rq);
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
    if (!host_sync) {
        // (stmmqr_factorize_phases: whatever follows on the plan's stream is ordered after the last send -- no host wait at all)
        HIPCHK(hipEventRecord(ring->ev_exp, cs));
        HIPCHK(hipStreamWaitEvent(st, ring->ev_exp, 0));
        return e;
    }
    // the caller gathers the contribution block (stmmqr_plan_export_front_cols) and goes on with the next phase: it needs the
    // device to have finished this one -- ONE synchronisation per shared front, not one per step
    HIPCHK(hipStreamSynchronize(cs));
    HIPCHK(hipStreamSynchronize(st));
    return e;
}

int stmmqr_factorize_shared_front(stmmqr_plan *plan, int group, stm_long f, int first_rank, int nranks, const stmmqr_transport *tr)
{
    return shared_front_loop(plan, group, f, first_rank, nranks, tr, true);
}

// ---- the subtree exchange, native (round 5; sharded.factorize_sharded's Python exchange stays the CPU-testable form).  Where
// subtrees join, the contribution block of a front factorized on one rank is assembled by a front of another (the unit of
// exchange of the reference's task tree: the child's packed C block, SparseQR_factorize.c:1228).  The Python form asks the device
// for (fm, rank, cm, csize) of every block, trades these as a message of their own, allocates buffers of the sizes received and
// then trades blocks and row ids: two host synchronisations, two rounds of messages and a dozen interpreter calls per block.
// Here a block is ONE message of a size the plan knows (its symbolic slot: stm_launch_front_msg), packed on the device, sent and
// received on the plan's stream, unpacked on the device: nothing waits on the host. ----
static long long front_msg_doubles(const stmmqr_plan &P, stm_long f) { return 8 + P.c_slot[(size_t)f] + (P.fs[f].fn - P.fs[f].fp); }

static int msg_arena(stmmqr_plan &P, long long doubles)
{
    if ((long long)P.d_msg.n >= doubles) return 0;
    HIPCHK(hipStreamSynchronize(P.stream));                  // (growing: first exchange of a grouping only)
    if (P.d_msg.alloc((size_t)doubles)) return fail(STMMQR_ERR_OUT_OF_MEMORY, "message buffers of the subtree exchange");
    return 0;
}

int stmmqr_factorize_exchange(stmmqr_plan *plan, stm_long nout, const stm_long *out_front, const int *out_peer, stm_long nin,
                              const stm_long *in_front, const int *in_peer, const stmmqr_transport *tr)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    if (!tr || !tr->send || !tr->recv) return fail(STMMQR_ERR_INVALID, "null transport");
    if (nout < 0 || nin < 0 || (nout > 0 && (!out_front || !out_peer)) || (nin > 0 && (!in_front || !in_peer)))
        return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_exchange: null lists");
    stmmqr_plan &P = *plan;
    HIPCHK(hipSetDevice(P.device));
    // every front packed once (it may go to several ranks: the ranks that share its parent), every incoming block its own buffer
    std::vector<stm_long> uo;
    std::vector<long long> off_o((size_t)nout, 0), off_i((size_t)nin, 0);
    std::vector<long long> uoff;
    long long top = 0;
    for (stm_long q = 0; q < nout; q++) {
        const stm_long f = out_front[q];
        if (int e = stm_check_c_slot(P, f, 0, "stmmqr_factorize_exchange")) return e;
        if (out_peer[q] < 0 || out_peer[q] >= tr->size || out_peer[q] == tr->rank) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_exchange: bad peer");
        size_t u = 0;
        while (u < uo.size() && uo[u] != f) u++;
        if (u == uo.size()) { uo.push_back(f); uoff.push_back(top); top += front_msg_doubles(P, f); }
        off_o[(size_t)q] = uoff[u];
    }
    for (stm_long q = 0; q < nin; q++) {
        const stm_long f = in_front[q];
        if (int e = stm_check_c_slot(P, f, 0, "stmmqr_factorize_exchange")) return e;
        if (in_peer[q] < 0 || in_peer[q] >= tr->size || in_peer[q] == tr->rank) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_exchange: bad peer");
        off_i[(size_t)q] = top;
        top += front_msg_doubles(P, f);
    }
    if (top == 0) return 0;
    if (int e = msg_arena(P, top)) return e;
    const DevCtx c = P.ctx();
    hipStream_t st = P.stream;
    auto launch = [&](const std::vector<stm_long> &fr, const long long *off, int out) -> int {
        for (size_t q0 = 0; q0 < fr.size(); q0 += 16) {
            StmFrontMsgs g;
            memset(&g, 0, sizeof g);
            long long mx = 0;
            const int n = (int)std::min<size_t>(16, fr.size() - q0);
            for (int q = 0; q < n; q++) {
                const stm_long f = fr[q0 + (size_t)q];
                g.m[q].f = (int)f; g.m[q].off = off[q0 + (size_t)q]; g.m[q].slot = P.c_slot[(size_t)f];
                mx = std::max(mx, P.c_slot[(size_t)f]);
            }
            LCHK(stm_launch_front_msg(c, g, n, mx, P.d_msg.p, out, st));
            P.stats.nlaunch++;
        }
        return 0;
    };
    if (int e = launch(uo, uoff.data(), 1)) return e;
    int e = 0;
    if (tr->group_begin && tr->group_begin(tr->ctx)) return fail(STMMQR_ERR_DEVICE, "transport: group begin");
    for (stm_long q = 0; q < nout && !e; q++)
        if (tr->send(tr->ctx, P.d_msg.p + off_o[(size_t)q], (size_t)front_msg_doubles(P, out_front[q]) * sizeof(double), out_peer[q], (void *)st))
            e = fail(STMMQR_ERR_DEVICE, "transport: send of a contribution block failed");
    for (stm_long q = 0; q < nin && !e; q++)
        if (tr->recv(tr->ctx, P.d_msg.p + off_i[(size_t)q], (size_t)front_msg_doubles(P, in_front[q]) * sizeof(double), in_peer[q], (void *)st))
            e = fail(STMMQR_ERR_DEVICE, "transport: receive of a contribution block failed");
    if (tr->group_end && tr->group_end(tr->ctx) && !e) e = fail(STMMQR_ERR_DEVICE, "transport: group end");
    if (e) return e;
    std::vector<stm_long> fin(in_front, in_front + nin);
    return launch(fin, off_i.data(), 0);
}

// The contribution block of a shared front, complete on every rank of its group in the columns that rank owns, collected on the
// group's first rank -- the native form of the tail of sharded.run_shared_front.  Message of place j: its runs of columns back to
// back (k_front_cols), sized by the symbolic bound of cm so that nobody asks the device for cm.
static long long front_cols_bound(const stmmqr_plan &P, stm_long f, int part, int nparts, int *nown, long long *max_run)
{
    const FrontSym &s = P.fs[f];
    const long long cn = s.fn - s.fp, fm = s.fm_ub;
    const long long cm = P.do_rank ? std::min(fm, cn) : std::min(std::max(fm - std::min(fm, (long long)s.fp), 0LL), cn);
    auto coff = [&](long long j) -> long long { return j < cm ? j * (j + 1) / 2 : cm * (cm + 1) / 2 + (j - cm) * cm; };
    long long tot = 0;
    *nown = 0; *max_run = 0;
    for (long long q = s.fp / STM_NB; q * STM_NB < s.fn; q++) {
        if (q % nparts != part) continue;
        const long long j0 = std::max(0LL, q * STM_NB - (long long)s.fp), j1 = std::min(cn, (q + 1) * STM_NB - (long long)s.fp);
        if (j1 <= j0) continue;
        (*nown)++;
        *max_run = std::max(*max_run, coff(j1) - coff(j0));
        tot += coff(j1) - coff(j0);
    }
    return tot;
}

int stmmqr_shared_front_gather(stmmqr_plan *plan, stm_long f, int first_rank, int nranks, const stmmqr_transport *tr)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    if (!tr || !tr->send || !tr->recv) return fail(STMMQR_ERR_INVALID, "null transport");
    stmmqr_plan &P = *plan;
    if (f < 0 || f >= P.nf || (size_t)f >= P.shared.size() || !P.shared[(size_t)f]) return fail(STMMQR_ERR_INVALID, "not a shared front");
    const int R = nranks, i = tr->rank - first_rank;
    if (R < 1 || i < 0 || i >= R) return fail(STMMQR_ERR_INVALID, "this rank is not in the front's group");
    if (R == 1 || P.fs[f].parent < 0) return 0;              // (the root's contribution block is empty)
    HIPCHK(hipSetDevice(P.device));
    if (int e = stm_check_c_slot(P, f, 0, "stmmqr_shared_front_gather")) return e;
    std::vector<long long> nd((size_t)R, 0), mr((size_t)R, 0), off((size_t)R, 0);
    std::vector<int> nown((size_t)R, 0);
    long long top = 0;
    for (int j = (i == 0 ? 1 : i); j < (i == 0 ? R : i + 1); j++) {
        nd[(size_t)j] = front_cols_bound(P, f, j, R, &nown[(size_t)j], &mr[(size_t)j]);
        off[(size_t)j] = top;
        top += nd[(size_t)j];
    }
    if (top == 0) return 0;
    if (int e = msg_arena(P, top)) return e;
    const DevCtx c = P.ctx();
    hipStream_t st = P.stream;
    int e = 0;
    if (i != 0) {
        LCHK(stm_launch_front_cols(c, (int)f, i, R, nown[(size_t)i], mr[(size_t)i], P.d_msg.p, 1, st));
        P.stats.nlaunch++;
    }
    if (tr->group_begin && tr->group_begin(tr->ctx)) return fail(STMMQR_ERR_DEVICE, "transport: group begin");
    if (i != 0) {
        if (tr->send(tr->ctx, P.d_msg.p, (size_t)nd[(size_t)i] * sizeof(double), first_rank, (void *)st)) e = fail(STMMQR_ERR_DEVICE, "transport: send of a shared front's columns failed");
    } else
        for (int j = 1; j < R && !e; j++)
            if (nd[(size_t)j] > 0 && tr->recv(tr->ctx, P.d_msg.p + off[(size_t)j], (size_t)nd[(size_t)j] * sizeof(double), first_rank + j, (void *)st))
                e = fail(STMMQR_ERR_DEVICE, "transport: receive of a shared front's columns failed");
    if (tr->group_end && tr->group_end(tr->ctx) && !e) e = fail(STMMQR_ERR_DEVICE, "transport: group end");
    if (e) return e;
    if (i == 0)
        for (int j = 1; j < R; j++)
            if (nd[(size_t)j] > 0) {
                LCHK(stm_launch_front_cols(c, (int)f, j, R, nown[(size_t)j], mr[(size_t)j], P.d_msg.p + off[(size_t)j], 0, st));
                P.stats.nlaunch++;
            }
    return 0;
}

// One sharded factorization between stmmqr_factorize_begin and stmmqr_factorize_finish as ONE call: per phase the exchange of the
// blocks that enter it, the shared front this rank takes part in (panel loop + gather) or the rank's own fronts of the phase.
// The only host waits left are the four bytes stmmqr_factorize_group reads after a group (did a bounded panel wait run out?).
int stmmqr_factorize_phases(stmmqr_plan *plan, const stmmqr_shard_phases *ph, const stmmqr_transport *tr)
{
    if (!plan || !plan->begun) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_begin was not called");
    if (!ph || ph->nphase < 1 || !ph->out_ptr || !ph->in_ptr || !ph->shared_front || !ph->has_group)
        return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_phases: null phase lists");
    if (!tr) return fail(STMMQR_ERR_INVALID, "null transport");
    {
        // the message buffers of the largest phase, once, before anything is enqueued (growing them later would wait for the device)
        stmmqr_plan &P = *plan;
        long long need = 0;
        for (int k = 0; k < ph->nphase; k++) {
            long long t = 0;
            std::vector<stm_long> seen;
            for (stm_long q = ph->out_ptr[k]; k > 0 && q < ph->out_ptr[k + 1]; q++) {
                const stm_long f = ph->out_front[q];
                if (f < 0 || f >= P.nf) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_phases: no such front");
                if (std::find(seen.begin(), seen.end(), f) == seen.end()) { seen.push_back(f); t += front_msg_doubles(P, f); }
            }
            for (stm_long q = ph->in_ptr[k]; k > 0 && q < ph->in_ptr[k + 1]; q++) {
                const stm_long f = ph->in_front[q];
                if (f < 0 || f >= P.nf) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_phases: no such front");
                t += front_msg_doubles(P, f);
            }
            need = std::max(need, t);
            const stm_long f = ph->shared_front[k];
            if (f >= 0 && f < P.nf && ph->shared_span && ph->shared_first && P.fs[f].parent >= 0) {
                const int R = ph->shared_span[k], i = tr->rank - ph->shared_first[k];
                long long g = 0, mr;
                int no;
                for (int j = (i == 0 ? 1 : i); j < (i == 0 ? R : i + 1) && j < R && j >= 0; j++) g += front_cols_bound(P, f, j, R, &no, &mr);
                need = std::max(need, g);
            }
        }
        HIPCHK(hipSetDevice(P.device));
        if (need > 0)
            if (int e = msg_arena(P, need)) return e;
    }
    for (int k = 0; k < ph->nphase; k++) {
        if (k > 0 && tr->size > 1) {
            const stm_long o0 = ph->out_ptr[k], o1 = ph->out_ptr[k + 1], i0 = ph->in_ptr[k], i1 = ph->in_ptr[k + 1];
            if (o1 > o0 || i1 > i0)
                if (int e = stmmqr_factorize_exchange(plan, o1 - o0, ph->out_front + o0, ph->out_peer + o0, i1 - i0, ph->in_front + i0,
                                                      ph->in_peer + i0, tr))
                    return e;
        }
        const stm_long f = ph->shared_front[k];
        if (f >= 0) {
            if (!ph->shared_first || !ph->shared_span) return fail(STMMQR_ERR_INVALID, "stmmqr_factorize_phases: null phase lists");
            if (int e = shared_front_loop(plan, k, f, ph->shared_first[k], ph->shared_span[k], tr, false)) return e;
            if (int e = stmmqr_shared_front_gather(plan, f, ph->shared_first[k], ph->shared_span[k], tr)) return e;
        }
        if (ph->has_group[k])
            if (int e = stmmqr_factorize_group(plan, k, 0)) return e;
    }
    return 0;
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

}  // extern "C"
