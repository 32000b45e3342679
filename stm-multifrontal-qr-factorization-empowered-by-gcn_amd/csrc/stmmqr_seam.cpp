// stmmqr_seam.cpp -- the drop-in seam: qr_factorize with the reference's structs (STMMQR/include/SparseQR.h:127-135), the reference's
// allocator accounting (SparseCore_malloc / free), the seam's plan cache.
#include "stmmqr_plan.h"

extern "C" {

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
                          "STMMQR_RH_EST_SCALE", "STMMQR_EARLY_END", "STMMQR_EARLY_SLACK", "STMMQR_PART_GRAIN", "STMMQR_PART_CAP"}) {
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
        st = stm_ensure_device(-1);
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

