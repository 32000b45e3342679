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
#include "stmmqr_plan.h"

thread_local std::string g_err;
stmmqr_options g_opt = {STM_NB, 64, 0, 0, 0, 1, STM_TALL_MIN, 2, 0, 4};      // (panel_algo 0: by panel height; lookahead 2: passenger launches)
size_t g_chunk[4] = {32, 5000, 4, 4};     // FCHUNK, SMALL, MINCHUNK, MINCHUNK_RATIO (SparseQR.h:16-19)

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
    // ---- how many panels of a front get a step (FrontSym::nsched).  A front of fm rows and fn > fm columns runs out of rows at
    // column fm: the panel that holds that column finalises every column behind it (dev_panel: "no rows left") and the panels after
    // it have nothing to do -- yet each was a step of the timeline, and the parent could not start before the last of them: 446 of the
    // 1249 steps on the critical path of the default workload (8000 of its 15 313 panels), 40 of epb1's 132.  fm is only known on
    // the device, but fm_est -- the rows if no pivot column dies -- is exact for a full-rank matrix: a plan that holds the whole tree
    // schedules floor(min(fm_est, fn) / 32) + 1 panels per front.  Should a front NOT be finished by its last scheduled panel
    // (pivot columns died below it: more rows reach it than estimated), k_cpack's extra workgroup raises abort[2] and the
    // factorization is run again on the full schedule, which the plan then keeps (P.full_schedule; stats.retries counts it).
    // STMMQR_EARLY_END=0: every panel a step, as before. ----
    {
        bool whole = (ngroups == 1) && nf > 0;
        for (long f = 0; f < nf && whole; f++) whole = (P.group[f] == 0) && !((size_t)f < P.shared.size() && P.shared[(size_t)f]);
        const bool early = (whole || P.early_phased) && !P.full_schedule && !(getenv("STMMQR_EARLY_END") && atoi(getenv("STMMQR_EARLY_END")) == 0);
        const int slack = getenv("STMMQR_EARLY_SLACK") ? atoi(getenv("STMMQR_EARLY_SLACK")) : 0;
        for (long f = 0; f < nf; f++) {
            FrontSym &s = P.fs[f];
            s.nsched = s.npanels;
            if (early && is_big((int)f) && !is_pair((int)f) && P.group[f] >= 0 && !((size_t)f < P.shared.size() && P.shared[(size_t)f])) s.nsched = std::min(s.npanels, std::min(s.fm_est, s.fn) / STM_NB + 1 + slack);
        }
        P.early_end = early;
    }
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
            end[f] = t0 + (is_big((int)f) ? P.fs[f].nsched : 1);
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
                    lend[f] = lstart[f] + (is_big((int)f) ? P.fs[f].nsched : 1);
                    e1 = std::max(e1, lend[f]);
                }
                lvl_end[lv + 1] = e1;
            }
            nstep_level = lvl_end[nlev];
        }
        // Which order (STMMQR_SCHED unset): the envelope rule reaches the minimum number of steps (nstep here) but its steps
        // are ~10-15 % longer than those of the level-synchronous order (a launch lasts as long as its slowest front and
        // rows are only a proxy for that; measured) -- it is taken when it removes more than a fifth of the steps.
        // (Round 5: with the update beyond block 0 riding on the chain's launches and the fronts ending where they run out of rows, the
        //  envelope order wins wherever it removes steps at all -- default workload 803 against 985 steps: 94.4 against 106.3 ms, sme3Dc
        //  stand-in 541 / 637: 65.9 / 71.3, xenon1-METIS 406 / 436: 43.9 / 45.0, epb1 96 / 124: 6.3 / 7.6; a tie in steps goes to the
        //  level-synchronous order: c5mini 250 / 250: 42.3 / 42.6.)
        const bool use_level = sched_policy == 1 || (sched_policy == 0 && 100L * nstep > 97L * nstep_level);
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
                for (int q = 0; q < P.fs[f].nsched; q++) panel_at[start[f] + q].push_back({f, q});
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
                tails[f] = (is_big((int)f) ? P.fs[f].nsched : 1) + (parent_in[f] >= 0 ? tails[parent_in[f]] : 0);
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
                            if (++e.p >= P.fs[e.f].nsched) { finish(e.f); continue; }
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
                if (fp.second == P.fs[fp.first].nsched - 1) ending[t].push_back(fp.first);
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
                static const long pg = getenv("STMMQR_PART_GRAIN") ? atol(getenv("STMMQR_PART_GRAIN")) : 2048, pc = getenv("STMMQR_PART_CAP") ? atol(getenv("STMMQR_PART_CAP")) : 4096;
                const int parts = (int)std::min(pc, std::max(1L, (work + pg - 1) / pg));     // (entries per workgroup of the assembly / packing launches: 16384 until round 5 -- 2048: epb1 5.81 -> 5.60 ms, default 93.3 -> 92.9)
                P.lists.push_back(parts);
                S.asm_maxparts = std::max(S.asm_maxparts, parts);
                if (i < S.n_small) maxfm_small = std::max(maxfm_small, (long)s.fm_ub);
            }
            // (rows padded as dev_panel pads them, so that a whole panel fits whenever the cap allows: the sub-panel width
            //  of a front -- and with it the rounding -- must not depend on which other fronts share its step)
            S.lds_small = (int)std::min((long)LDS_CAP_SMALL, (((maxfm_small + 63) & ~63L) | 1) * STM_NB + 64);
            // (the wave-pipelined panels of k_front_wg: an image of 32 x 128 rows, and dev_gram_T's 4 x 768 doubles of scratch)
            if (S.n_small > 0) S.lds_small = std::max(S.lds_small, STM_NB * 128 + 64);
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
                static const long pg = getenv("STMMQR_PART_GRAIN") ? atol(getenv("STMMQR_PART_GRAIN")) : 2048, pc = getenv("STMMQR_PART_CAP") ? atol(getenv("STMMQR_PART_CAP")) : 4096;
                const int parts = (int)std::min(pc, std::max(1L, (work + pg - 1) / pg));
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
    if (!P.d_abort.p) LCHK(P.d_abort.alloc(4));
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
    HIPCHK(hipMemsetAsync(P.d_abort.p, 0, 4 * sizeof(int), st));
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
    HIPCHK(hipMemsetAsync(P.d_abort.p, 0, 3 * sizeof(int), st));   // ([3], a refused message of the subtree exchange, stays)
    // slab recycling: a recycling plan holds the whole tree in this one group, and the aborted attempt has staged packed blocks behind
    // the arena's bump pointer; the rerun stages every block again, so the pointer (and the overflow word) go back to zero -- otherwise
    // the second set lands behind the first and overflows an arena that holds the estimate + 12.5 %
    if (P.recycle) HIPCHK(hipMemsetAsync(P.d_rhtop.p, 0, 2 * sizeof(long long), st));
    HIPCHK(hipStreamSynchronize(st));                       // (`zero` lives on this stack frame)
    return 0;
}

// One piece of one timeline step (stmmqr_factorize_step): what = STMMQR_STEP_* bits; the update takes the column blocks
// cb_first, cb_first + cb_stride, ... (at most cb_count of them when cb_count >= 0) of the step's fronts.
// (StepReq: stmmqr_plan.h)

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
        const int pass_rows = getenv("STMMQR_PASS_ROWS") ? atoi(getenv("STMMQR_PASS_ROWS")) : 16384;
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
                if (S.nca_use && !S.npipe_use && pend && !(abl & 6) && !(getenv("STMMQR_CA_RIDERS") && atoi(getenv("STMMQR_CA_RIDERS")) == 0)) {
                    // (every panel of the step is Gram-based: the riders of the step before take THAT launch)
                    const Step &Q = *pend;
                    pend = nullptr;
                    LCHK(stm_launch_panel_ca_pc(c, act, pl, S.n_act, S.nca, 1, L0 + Q.act_off, L0 + Q.plist_off, Q.n_norm, 1, Q.maxcb - 1, Q.maxsl,
                                                P.d_Wp2.p, P.d_wlists.p + Q.wp_off, st));
                    nlaunch++;
                } else if (S.nca_use) { LCHK(stm_launch_panel_ca(c, act, pl, S.n_act, S.nca, 1, st)); nlaunch++; }
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
                // (pass_rows: 5120 until the riders of a step before Gram-based panels got a launch to ride on (k_panel_ca_pc) -- beyond
                //  ~5000 rows the one-launch block 0 is the slower form of T + block 0, but the riders no longer cost a launch of their
                //  own on the chain: c5mini 41.1 -> 37.0 ms, default 91.4 -> 90.9 at 8192 ... unlimited; 16384 = where the pair-update
                //  fronts begin, which never ride)
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
            // (a front that ends at a panel with trailing columns -- FrontSym::nsched -- is packed only after the riders of that update)
            if (pend && P.early_end && (S.n_cpk > 0 || (P.recycle && S.n_rhp > 0)) && (e = flush_alone())) return e;
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
            // (early end: a front may end at a panel with trailing columns; its step is packed after the whole update)
            const bool offload = S.n_act > 0 && S.maxcb > 1 && worth_it(S) && !(P.early_end && (S.n_cpk > 0 || (P.recycle && S.n_rhp > 0)));
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

// (what the other host translation units need of the planner / scheduler: stmmqr_plan.h)
int stm_run_schedule(stmmqr_plan &P, bool detail, int grp, const StepReq *req) { return run_schedule(P, detail, grp, req); }
int stm_ensure_device(int device) { return ensure_device(device); }

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
    if (((P.early_end && !P.early_phased) || P.early_end_failed) && !P.whole_call) {
        // phased use (begin / group / finish by the caller): no retry loop around the factorization, so the plan schedules every
        // panel of every front (the cut schedule needs stmmqr_factorize_device's rerun when a front outlives it)
        P.early_end_failed = false;
        P.full_schedule = true;
        P.begun = false;
        std::vector<int> grp(P.group.begin(), P.group.end());
        int e = stmmqr_plan_set_groups(plan, grp.data());
        if (e) return e;
    }
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
    int hard[2] = {0, 0};                                     // abort[2]: a front outlived its schedule; abort[3]: a refused message
    HIPCHK(hipMemcpyAsync(hard, P.d_abort.p + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (hard[1]) return fail(STMMQR_ERR_DEVICE, "a contribution block received from another rank does not fit its front's symbolic bounds");
    if (hard[0]) {
        // (rank-deficient fronts: more rows reached a front than the full-rank estimate its schedule was cut to)
        P.early_end_failed = true;
        P.evused = 0;
        return fail(STMMQR_ERR_RESCHEDULE, "a front was not finished by its last scheduled panel (the schedule is rebuilt with every panel)");
    }
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
        if (getenv("STMMQR_DBG_EARLY") && s.nsched < s.npanels && (!nm.done || nm.g < std::min(nm.fm, s.fn)))
            fprintf(stderr, "[early] front %ld fn %d fp %d fm %d fm_est %d fm_ub %d g %d rank %d done %d nsched %d npanels %d\n", f, s.fn, s.fp, nm.fm, s.fm_est,
                    s.fm_ub, nm.g, nm.rank, nm.done, s.nsched, s.npanels);
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
    int arena_retries = 0, reschedules = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        plan->serial_panels = (attempt == 1);
        plan->panel_wait_failed = false;
        plan->whole_call = true;
        int e = stmmqr_factorize_begin(plan, Ap, Ai, Ax, ax_on_device, tol, ntol);
        if (!e) plan->stats.retries = (attempt == 1 ? 1 : 0) + arena_retries;   // (visible in stmmqr_stats: bench.py asserts 0)
        if (!e) plan->stats.reschedules = reschedules;
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
        if (e && plan->early_end_failed) {
            // a front had rows left at its last scheduled panel: once more with every panel scheduled (and from now on)
            plan->early_end_failed = false;
            plan->full_schedule = true;
            if (g_opt.verbose) fprintf(stderr, "[stmmqr_hip] a front outlived its schedule (rank-deficient fronts): factorizing again on the full schedule\n");
            plan->begun = false;
            std::vector<int> grp(plan->group.begin(), plan->group.end());
            int e2 = stmmqr_plan_set_groups(plan, grp.data());
            if (e2) return e2;
            reschedules++;
            attempt = -1;
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

int stmmqr_plan_set_early_end(stmmqr_plan *plan, int mode)
{
    if (!plan) return fail(STMMQR_ERR_INVALID, "null plan");
    stmmqr_plan &P = *plan;
    if (P.begun && !P.early_end_failed) return fail(STMMQR_ERR_INVALID, "stmmqr_plan_set_early_end between factorize_begin and factorize_finish");
    P.begun = false;
    P.early_end_failed = false;
    P.early_phased = (mode != 0);
    P.full_schedule = (mode == 0);
    std::vector<int> grp(P.group.begin(), P.group.end());
    for (size_t f = 0; f < grp.size(); f++) if (f < P.shared.size() && P.shared[f]) grp[f] |= STMMQR_GROUP_SHARED;
    return stmmqr_plan_set_groups(plan, grp.data());
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
    if (!P.d_abort.p) LCHK(P.d_abort.alloc(4));
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
int stm_check_c_slot(const stmmqr_plan &P, stm_long f, long long csize, const char *what)
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

}  // extern "C"
