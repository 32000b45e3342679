/* include/stmmqr_hip.h -- C ABI of libstmmqr_hip.so, the MI355X-native numerical-factorization path
 * for STM-Multifrontal-QR.
 *
 * Drop-in scope: the hot path of the reference, STMMQR/src/qr/SparseQR_factorize.c +
 * SparseQR_multithreads.c (qr_factorize -> qr_kernel -> {qr_fsize, qr_assemble, qr_front, qr_cpack,
 * qr_rhpack} + qr_stranspose2 + qr_hpinv), executed on one gfx950 device.  All entry points are plain
 * C (extern "C", pointers and sizes only).  "Long" below is the reference's Sparse_long = C long
 * (STMMQR/include/SparseBase_config.h:29); all numerics are fp64 (Makefile.option:62).
 *
 * Three groups of symbols:
 *   1. the drop-in seam with the reference's own names and signatures (a maintainer links this library
 *      instead of compiling SparseQR_factorize.c / SparseQR_multithreads.c);
 *   2. the same path on plain arrays (stmmqr_*), used by the Python host side, the tests and bench.py;
 *   3. configuration / introspection.
 *
 * Error convention (reference: SparseQR_factorize.c:329-333,378-383,540-545): nothing throws across
 * this boundary; failures free partial state, return NULL / a negative code and leave
 * cc->status < SPARSE_OK.  A missing or unusable GPU is an error (STMMQR_ERR_DEVICE), never a CPU
 * fallback.
 */
#ifndef STMMQR_HIP_H
#define STMMQR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef long stm_long;

/* ---- status codes (values of cc->status in the reference: SparseCore.h "SPARSE_OK" family) ---- */
#define STMMQR_OK                0
#define STMMQR_ERR_OUT_OF_MEMORY (-2)   /* SPARSE_OUT_OF_MEMORY */
#define STMMQR_ERR_TOO_LARGE     (-3)   /* SPARSE_TOO_LARGE     */
#define STMMQR_ERR_INVALID       (-4)   /* SPARSE_INVALID       */
#define STMMQR_ERR_DEVICE        (-5)   /* SPARSE_GPU_PROBLEM: no gfx950 device / HIP runtime error */
#define STMMQR_ERR_RESCHEDULE    (-6)   /* stmmqr_factorize_finish: a front still had rows at the last panel the plan scheduled for it
                                           (stmmqr_plan_set_early_end); nothing is lost but this attempt -- factorize again */

/* ================================================================================================
 * Layout-compatible mirrors of the structs that cross the seam
 * ================================================================================================ */

/* sparse_csc  (STMMQR/include/SparseCore.h:514-556) */
typedef struct stm_sparse_csc {
    size_t nrow, ncol, nzmax;
    void *p, *i, *nz, *x, *z;
    int stype, itype, xtype, dtype, sorted, packed;
} stm_sparse_csc;

/* qr_symbolic (STMMQR/include/SparseQR_struct.h:26-137): produced by qr_analyze, borrowed, never modified */
typedef struct stm_qr_symbolic {
    stm_long m, n, anz;
    stm_long *Sp, *Sj, *Qfill, *PLinv, *Sleft;
    stm_long nf, maxfn;
    stm_long *Parent, *Child, *Childp, *Super, *Rp, *Rj, *Post;
    stm_long rjsize, do_rank_detection, maxstack, hisize, keepH;
    stm_long *Hip;
    stm_long ntasks, ns;
    stm_long *TaskChildp, *TaskChild, *TaskStack, *TaskFront, *TaskFrontp, *On_stack, *Stack_maxstack;
    stm_long *Fm, *Cm;
} stm_qr_symbolic;

/* qr_numeric (STMMQR/include/SparseQR_struct.h:145-209): returned, owned by the caller, released by the
 * reference's qr_freenum (SparseQR.c:1225-1272), hence every array is malloc'ed with the sizes stored here
 * and accounted in cc->malloc_count / memory_inuse exactly like SparseCore_malloc does. */
typedef struct stm_qr_numeric {
    double **Rblock;
    double **Stacks;
    stm_long *Stack_size;
    stm_long hisize, n, m, nf, ntasks, ns, maxstack;
    char *Rdead;
    stm_long rank, rank1, maxfrank;
    double norm_E_fro;
    stm_long keepH, rjsize;
    stm_long *HStair;
    double *HTau;
    stm_long *Hii, *HPinv, *Hm, *Hr;
    stm_long maxfm;
} stm_qr_numeric;

/* sparse_common is opaque here: the few fields the path touches are reached through byte offsets.
 * The defaults are the offsets of the stock reference build (LP64); a host that compiles the reference
 * with a different layout passes its own offsetof() values once (INTEGRATION.md shows the stub). */
typedef struct sparse_common_struct stm_sparse_common;
typedef struct stm_common_layout {
    size_t status, malloc_count, memory_usage, memory_inuse, blas_ok;
    size_t SPQR_grain, SPQR_small, SPQR_shrink, SPQR_flopcount, SPQR_flopcount_bound;
} stm_common_layout;
void stmmqr_set_common_layout(const stm_common_layout *layout);
void stmmqr_get_common_layout(stm_common_layout *layout);
/* SparseCore_malloc / SparseCore_free / cc->status through that layout (src/core/SparseCore_common.c:603-655): what a host-side
 * companion of this library hands to code that releases it with the reference's allocator (libstmmqr_hip_api.so does) */
void *stmmqr_cc_malloc(size_t n, size_t size, stm_sparse_common *cc);
void stmmqr_cc_free(size_t n, size_t size, void *p, stm_sparse_common *cc);
void stmmqr_cc_set_status(stm_sparse_common *cc, int code);

/* ================================================================================================
 * 1. Drop-in seam (reference names; prototypes STMMQR/include/SparseQR.h:127-268)
 * ================================================================================================ */

/* SparseQR.h:127-135; call sites SparseQR.c:349 (&A,FALSE,tol,n) and :371 (&Y,TRUE,tol,n2) */
stm_qr_numeric *qr_factorize(stm_sparse_csc **Ahandle, stm_long freeA, double tol, stm_long ntol,
                             stm_qr_symbolic *QRsym, stm_sparse_common *cc);

/* The seam keeps the plan of the last qr_symbolic it factorized (symbolic upload, schedule, workspaces, front arenas: about as
 * expensive to build as one factorization of the BASELINE matrices): a later call with an EQUAL qr_symbolic -- compared by a hash of
 * its scalars and arrays, the options and the plan-time environment -- and the same device reuses it, and skips the value map
 * (qr_stranspose2) too when A's pattern is unchanged.  A cached plan keeps its device memory until stmmqr_plan_cache_clear() /
 * stmmqr_shutdown(); env STMMQR_PLAN_CACHE = 0 turns the cache off, = n keeps the n most recent plans (default 1).
 * STMMQR_SEAM_TIMING=1 prints where a call spent its time.  Match: the interval SparseQR.c:346-377 brackets ("Factorize time"). */
void stmmqr_plan_cache_clear(void);
/* releases a qr_numeric returned by qr_factorize with the reference's accounting (= qr_freenum, SparseQR.c:1225-1272), for hosts
 * that link this library without the reference */
void stmmqr_free_numeric(stm_qr_numeric **QRnum, stm_sparse_common *cc);

/* SparseQR.h:137-143 ; globals FCHUNK/SMALL/MINCHUNK/MINCHUNK_RATIO (SparseQR.h:16-19) become library state */
int chunk_getSettings(size_t fchunk, size_t small_, size_t minchunk, size_t minchunk_ratio);

/* Inner seams on HOST buffers (device copies are made inside; used for unit parity).
 * cc may be NULL.  SparseQR.h:145-268. */
void qr_stranspose2(stm_sparse_csc *A, stm_long *Qfill, stm_long *Sp, stm_long *PLinv, double *Sx, stm_long *W);
void qr_hpinv(stm_qr_symbolic *QRsym, stm_qr_numeric *QRnum, stm_long *W);
stm_long qr_fsize(stm_long f, stm_long *Super, stm_long *Rp, stm_long *Rj, stm_long *Sleft, stm_long *Child,
                  stm_long *Childp, stm_long *Cm, stm_long *Fmap, stm_long *Stair);
void qr_assemble(stm_long f, stm_long fm, int keepH, stm_long *Super, stm_long *Rp, stm_long *Rj, stm_long *Sp,
                 stm_long *Sj, stm_long *Sleft, stm_long *Child, stm_long *Childp, double *Sx, stm_long *Fmap,
                 stm_long *Cm, double **Cblock, stm_long *Hr, stm_long *Stair, stm_long *Hii, stm_long *Hip,
                 double *F, stm_long *Cmap);
stm_long qr_csize(stm_long c, stm_long *Rp, stm_long *Cm, stm_long *Super);
stm_long qr_front(stm_long m, stm_long n, stm_long npiv, double tol, stm_long ntol, stm_long fchunk, double *F,
                  stm_long *Stair, char *Rdead, double *Tau, double *W, double *wscale, double *wssq,
                  stm_sparse_common *cc);
stm_long qr_fcsize(stm_long m, stm_long n, stm_long npiv, stm_long g);
stm_long qr_cpack(stm_long m, stm_long n, stm_long npiv, stm_long g, double *F, double *C);
stm_long qr_rhpack(int keepH, stm_long m, stm_long n, stm_long npiv, stm_long *Stair, double *F, double *R,
                   stm_long *p_rm);
void qr_larftb(int method, stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, double *V,
               double *Tau, double *C, double *W, stm_sparse_common *cc);

/* ================================================================================================
 * 2. The same path on plain arrays
 * ================================================================================================ */

/* symbolic inputs as plain arrays (same meaning as the qr_symbolic fields of the same name) */
typedef struct stmmqr_symbolic_view {
    stm_long m, n, anz, nf, maxfn, rjsize, hisize, do_rank_detection;
    const stm_long *Sp, *Sj, *Qfill, *PLinv, *Sleft;
    const stm_long *Child, *Childp, *Super, *Rp, *Rj, *Post, *Hip, *Fm;
    stm_long maxstack;          /* QRsym->maxstack: the analysis' bound for the reference's one stack, hence for all of R+H; sizes the
                                   R+H arena of the slab recycling.  0 = unknown (the arena then holds all recycled slabs)          */
} stmmqr_symbolic_view;

/* per-call measurements (all times in milliseconds, HIP events on the library's own stream) */
typedef struct stmmqr_stats {
    double flops;              /* the reference's FLOP_COUNT formula (SparseQR_factorize.c:1571)          */
    double ms_total;           /* upload-excluded device time: gather S .. pack R+H                        */
    double ms_assemble;        /* front_setup + assemble kernels                                            */
    double ms_front;           /* panel + trailing-update + small-front kernels (detail runs)              */
    double ms_pack;            /* R+H count / scan / copy kernels                                           */
    double ms_h2d, ms_d2h;     /* PCIe transfers of A values in, packed factors out                         */
    double ms_host;            /* host-side planning + hpinv                                                */
    double bytes_assemble;     /* algorithmic bytes of the assembly kernels, SURVEY.md 8(d) formula        */
    double bytes_pack;
    double flops_update;       /* flops executed by the MFMA trailing-update kernel (4*m*n*k per block)    */
    double ms_update;          /* time of the MFMA trailing-update launches only                            */
    stm_long nlaunch;          /* kernel launches in the timed region                                       */
    stm_long nlevels;
    /* detail runs only (stats->nlaunch == -1 on entry): HIP-event pairs around every launch of a category, recorded on
       the plan's stream without any synchronisation (the schedule runs as in an untimed call)                       */
    double ms_panel;           /* panel kernels of the large fronts (k_panel / k_panel_ca)                          */
    double ms_small;           /* k_front_wg: whole small fronts                                                    */
    stm_long npanel_launch, nupdate_launch;   /* event pairs behind ms_panel / ms_update (one per timeline step)    */
    stm_long nsteps;           /* steps of the factorization timeline (a big front advances one panel per step)     */
    double flops_update_pair;  /* the part of flops_update on fronts that take the pair update (two panels per sweep:
                                  12 instead of 24 algorithmic bytes per updated entry and panel)                     */
    stm_long retries;          /* times (part of) the factorization was run again with one-workgroup panels because a bounded
                                  inter-workgroup wait of a panel kernel ran out (0 in a healthy run: bench.py asserts it)  */
    double device_bytes;       /* device memory held by the plan when the factorization finished (arenas, factors, workspaces) */
    stm_long reschedules;      /* times the factorization was run again on the FULL schedule because a front still had rows at the
                                  last panel the plan had scheduled for it (the schedule of a whole-tree plan stops every front where
                                  the full-rank row estimate says it runs out of rows; rank-deficient input: once per plan, which
                                  then keeps the full schedule)                                                           */
} stmmqr_stats;

typedef struct stmmqr_plan stmmqr_plan;     /* device-resident symbolic plan + arenas; reusable across calls */

/* Build the device plan from the symbolic analysis (host work + one upload).  Returns NULL on error
 * (*status gets the code).  device < 0 selects the current HIP device.  The front and contribution-block arenas are
 * sized from the fronts the plan factorizes, shares or imports (all of them until stmmqr_plan_set_groups says otherwise)
 * and allocated by the first factorization: STMMQR_ERR_OUT_OF_MEMORY then comes from stmmqr_factorize_begin /
 * stmmqr_factorize_device, and one rank of a sharded run never holds the whole tree. */
stmmqr_plan *stmmqr_plan_create(const stmmqr_symbolic_view *sym, int device, int *status);
void stmmqr_plan_destroy(stmmqr_plan *plan);

/* Give the plan the pattern of A (column pointers / row indices): builds the qr_stranspose2 gather map
 * (SparseQR_factorize.c:755-785) once.  stmmqr_factorize_device does this itself when Ap/Ai are non-NULL. */
int stmmqr_plan_set_pattern(stmmqr_plan *plan, const stm_long *Ap, const stm_long *Ai);

/* Numeric factorization of A (CSC, Long indices) with a plan.  Results stay resident in HBM inside the plan;
 * stmmqr_plan_download copies them out.  Ax may be a host pointer (uploaded, timed as ms_h2d) or, when
 * ax_on_device != 0, a device pointer. */
int stmmqr_factorize_device(stmmqr_plan *plan, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                            int ax_on_device, double tol, stm_long ntol, stmmqr_stats *stats);

/* The same factorization in phases (multi-GPU): begin (upload values) -> factorize_group(g) for the groups of
 * stmmqr_plan_set_groups in increasing order, with stmmqr_plan_import_front calls in between -> finish (pack). */
int stmmqr_factorize_begin(stmmqr_plan *plan, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                           int ax_on_device, double tol, stm_long ntol);
int stmmqr_factorize_group(stmmqr_plan *plan, int group, int detail);
int stmmqr_factorize_finish(stmmqr_plan *plan, stmmqr_stats *stats);

/* Subtree sharding.  group[f] >= 0: front f is factorized on this device in phase group[f];
 * -1: on another device (its packed contribution block, row ids, fm/rank/cm arrive by import_front before the
 * phase of its parent).  A child must never be in a later phase than its parent. */
int stmmqr_plan_set_groups(stmmqr_plan *plan, const int *group /* [nf] */);
/* info[0..5] = fm, rank, cm, csize, fn, fp of a factorized (or imported) front */
int stmmqr_plan_front_info(stmmqr_plan *plan, stm_long f, stm_long *info);
/* packed contribution block (csize doubles, qr_cpack layout) + cm row ids of front f; C may be a device pointer */
int stmmqr_plan_export_front(stmmqr_plan *plan, stm_long f, double *C, stm_long *rows, int c_on_device);
int stmmqr_plan_import_front(stmmqr_plan *plan, stm_long f, stm_long fm, stm_long rank, stm_long cm, const double *C,
                             const stm_long *rows, int c_on_device);

/* A front SHARED between plans (one plan per device): the reference moves whole contribution blocks between
 * its stacks (SparseQR_factorize.c:1228) and leaves a big front to one thread team; here the trailing matrix of one front is
 * spread over the devices of a group by 32-column blocks.  Every plan of the group marks the front
 *     group[f] = phase | STMMQR_GROUP_SHARED            (alone in its group, one of the large fronts, no pair update)
 * assembles it, and then walks the group's timeline -- one step per 32-column panel -- with stmmqr_factorize_step:
 *     PREP (step 0)                       set up + assemble (every plan: each holds the whole front)
 *     PANEL (step q)                      only the plan that owns panel q (q % nparts == part), then
 *         stmmqr_plan_export_panel        -> the others  stmmqr_plan_import_panel   (columns, T, Tau/Stair/Rdead, progress)
 *     UPDATE|GRAM (step q)                every plan, on its own column blocks: block b of step q holds the columns of panel
 *                                         q + 1 + b, so plan `part` takes cb_first = (part - q - 1) mod nparts, stride nparts
 *                                         (GRAM: build T of the step's panel; once per step and plan, before or with the first
 *                                         UPDATE; cb_count limits the launch, e.g. to block 0 before the next PANEL)
 *     POST (last step)                    pack the contribution block (complete in the owned columns only:
 *                                         stmmqr_plan_export_front_cols / _import_front_cols gather it on one plan)
 * The arithmetic of a column block does not depend on the plan that runs it: same bits as the unshared front with
 * options.pair_update = 0.  After stmmqr_factorize_finish every plan's packed R+H block of the front is valid in the columns
 * of its own panels (stmmqr_plan_front_rhoff gives the column offsets for the merge). */
#define STMMQR_GROUP_SHARED  (1 << 30)
#define STMMQR_STEP_PREP     1
#define STMMQR_STEP_PANEL    2
#define STMMQR_STEP_UPDATE   4
#define STMMQR_STEP_GRAM     8
#define STMMQR_STEP_POST     16
/* steps of a group's timeline (a shared front: its number of panels); -1 = no such group */
int stmmqr_plan_group_steps(stmmqr_plan *plan, int group);
int stmmqr_factorize_step(stmmqr_plan *plan, int group, int step, int what, int cb_first, int cb_stride, int cb_count);
/* one panel message: ndoubles is the same for every panel of the front; buf may be a device pointer (on_device != 0).
 * export returns when the buffer is complete; import is ordered on the plan's stream. */
int stmmqr_plan_panel_doubles(stmmqr_plan *plan, stm_long f, stm_long *ndoubles);
int stmmqr_plan_export_panel(stmmqr_plan *plan, stm_long f, stm_long p, double *buf, int on_device);
int stmmqr_plan_import_panel(stmmqr_plan *plan, stm_long f, stm_long p, const double *buf, int on_device);
/* the columns of the packed contribution block that belong to the panels of `part` (buf == NULL: count only) */
int stmmqr_plan_export_front_cols(stmmqr_plan *plan, stm_long f, int part, int nparts, double *buf, int on_device,
                                  stm_long *ndoubles);
int stmmqr_plan_import_front_cols(stmmqr_plan *plan, stm_long f, int part, int nparts, const double *buf, int on_device);
/* The panel loop of a shared front, native (round 4): ONE call per rank and shared front does what the step-by-step interface
 * above does from a host loop -- PREP, then for every panel the owner's block 0 / PANEL / export / send / rest of its update and
 * the others' posted receive / update / import -- with everything enqueued on the plan's stream and a comm stream ordered by
 * events: no stream synchronisation and no interpreter between two steps; the send of panel t overlaps the owner's update of step
 * t - 1, a receiver posts the receive of panel t before its update of step t - 1.  This rank is tr->rank, the group is the ranks
 * [first_rank, first_rank + nranks).  Returns when the device has finished the front.
 * The transport is a table of callbacks on DEVICE buffers, enqueued on the stream they are given:
 *   stmmqr_rccl_transport_create   RCCL point-to-point (ncclSend / ncclRecv in a group) -- the xGMI path of a multi-GPU node;
 *                                  librccl is loaded at run time; the 128-byte id comes from stmmqr_rccl_unique_id on one rank
 *                                  and reaches the others by any means (bench.py: a torch.distributed broadcast)
 *   a caller's own table           tests play the ranks of a group on ONE GPU this way (tests/test_gpu_sharded.py) */
typedef struct stmmqr_transport {
    void *ctx;
    int (*send)(void *ctx, const void *dev_buf, size_t bytes, int peer, void *hip_stream);
    int (*recv)(void *ctx, void *dev_buf, size_t bytes, int peer, void *hip_stream);
    int (*group_begin)(void *ctx);          /* may be NULL */
    int (*group_end)(void *ctx);            /* may be NULL */
    int rank, size;
} stmmqr_transport;
int stmmqr_factorize_shared_front(stmmqr_plan *plan, int group, stm_long f, int first_rank, int nranks, const stmmqr_transport *tr);
/* The subtree exchange, native (round 5): the contribution blocks that enter a phase from other ranks.  The unit of exchange is the
 * reference's: the packed C block of a child front with its row ids (SparseQR_factorize.c:1228; between the reference's tasks it
 * stays in shared memory).  Block q of the `out` list goes to rank out_peer[q], block q of the `in` list comes from in_peer[q]; for
 * one pair of ranks the fronts appear in the same order on both sides.  A block is ONE message whose size the plan knows (8 doubles
 * of header + the front's symbolic C slot + fn - fp row ids), packed by a kernel that reads the front's state on the device, sent
 * and received on the plan's stream in one transport group, unpacked by a kernel that writes the state
 * stmmqr_plan_import_front would: the host neither waits for the device nor learns (fm, rank, cm).  A message that does not fit the
 * front's symbolic bounds makes the factorization fail (stats.failed at stmmqr_factorize_finish).
 * stmmqr_shared_front_gather: after the panel loop of a shared front, the columns of its contribution block that the other ranks
 * of the group own, collected on the group's first rank (the native form of stmmqr_plan_export/import_front_cols; nothing to do
 * for a root).
 * stmmqr_factorize_phases: everything between stmmqr_factorize_begin and stmmqr_factorize_finish of a sharded factorization as ONE
 * call -- per phase k: the exchange (lists [out_ptr[k], out_ptr[k+1]) / [in_ptr[k], in_ptr[k+1])), then the shared front this rank
 * takes part in (shared_front[k] >= 0: panel loop + gather, group [shared_first[k], + shared_span[k])) or its own fronts of the
 * phase (has_group[k]: stmmqr_factorize_group(plan, k)).  The only host waits left are the four bytes stmmqr_factorize_group reads
 * after a group. */
int stmmqr_factorize_exchange(stmmqr_plan *plan, stm_long nout, const stm_long *out_front, const int *out_peer, stm_long nin,
                              const stm_long *in_front, const int *in_peer, const stmmqr_transport *tr);
int stmmqr_shared_front_gather(stmmqr_plan *plan, stm_long f, int first_rank, int nranks, const stmmqr_transport *tr);
/* How many panels of a front get a step ("How many panels"): a front of fm rows needs floor(fm / 32) + 1 of its
 * ceil(fn / 32) panels, and fm -- known on the device only -- equals the plan-time estimate unless pivot columns die.
 * stmmqr_factorize_device cuts the schedule of a whole-tree plan there by itself and runs a factorization that outlives it again on
 * the full schedule.  A caller of the phased interface (begin / group / finish: sharded plans) has no such loop around it and gets the
 * full schedule unless it asks: mode 1 = cut schedule, and the CALLER handles STMMQR_ERR_RESCHEDULE from stmmqr_factorize_finish --
 * every rank of a sharded factorization must then factorize again (sharded.factorize_sharded agrees on that with one tiny exchange per
 * factorization); the plan that failed schedules every panel from its next begin on, the others are told by mode 0.
 * Rebuilds the schedule; not between begin and finish (except after that failure).  Shared fronts always keep every panel. */
int stmmqr_plan_set_early_end(stmmqr_plan *plan, int mode);
typedef struct stmmqr_shard_phases {
    int nphase;
    const stm_long *out_ptr, *out_front;    /* [nphase + 1], [out_ptr[nphase]] */
    const int *out_peer;
    const stm_long *in_ptr, *in_front;
    const int *in_peer;
    const stm_long *shared_front;           /* [nphase]: -1 = none */
    const int *shared_first, *shared_span;  /* [nphase] */
    const int *has_group;                   /* [nphase] */
} stmmqr_shard_phases;
int stmmqr_factorize_phases(stmmqr_plan *plan, const stmmqr_shard_phases *ph, const stmmqr_transport *tr);
int stmmqr_device_copy(void *dst, const void *src, size_t bytes, void *hip_stream);   /* D2D, complete on return, after the stream's work */
int stmmqr_rccl_unique_id(char id[128]);
int stmmqr_rccl_transport_create(int world, int rank, const char id[128], stmmqr_transport **out);
void stmmqr_rccl_transport_destroy(stmmqr_transport *tr);
int stmmqr_transport_sendrecv(const stmmqr_transport *tr, const void *sendbuf, size_t sendbytes, int dst, void *recvbuf, size_t recvbytes,
                              int src, void *hip_stream);

/* off[0..fn]: start of each column of front f inside its packed R+H block (after stmmqr_factorize_finish) */
int stmmqr_plan_front_rhoff(stmmqr_plan *plan, stm_long f, stm_long *off);

/* device memory the plan holds right now, in bytes (what stmmqr_stats.device_bytes reports at the end of a factorization) */
double stmmqr_plan_device_bytes(const stmmqr_plan *plan);

/* out[0..1] = the reference's flop count of front f (FLOP_COUNT, SparseQR_factorize.c:1571) and the part of it done by
 * trailing updates: every plan of a shared front counts the whole front, the merge keeps one */
int stmmqr_plan_front_flops(stmmqr_plan *plan, stm_long f, double *out);

/* sizes needed by the caller to allocate the outputs of stmmqr_plan_download */
int stmmqr_plan_result_sizes(const stmmqr_plan *plan, stm_long *rh_total, stm_long *rank);

/* Copy the results of the last stmmqr_factorize_device to host arrays laid out like qr_numeric:
 * Stack[rh_total] holds the packed R+H blocks in Post order (= the single shrunk stack of the reference's
 * serial run), Rblock_off[nf] the offset of each front's block in it.  Any pointer may be NULL. */
int stmmqr_plan_download(stmmqr_plan *plan, double *Stack, stm_long *Rblock_off, char *Rdead, stm_long *HStair,
                         double *HTau, stm_long *Hii, stm_long *HPinv, stm_long *Hm, stm_long *Hr,
                         stm_long *scalars /* [rank, rank1, maxfrank, maxfm] */, stmmqr_stats *stats);

/* one-shot convenience: plan + factorize + download (what qr_factorize does internally) */
int stmmqr_factorize_arrays(const stmmqr_symbolic_view *sym, const stm_long *Ap, const stm_long *Ai,
                            const double *Ax, double tol, stm_long ntol, double *Stack, stm_long stack_cap,
                            stm_long *Rblock_off, char *Rdead, stm_long *HStair, double *HTau, stm_long *Hii,
                            stm_long *HPinv, stm_long *Hm, stm_long *Hr, stm_long *scalars, stmmqr_stats *stats);

/* Q-apply and least-squares solve on the factors that are still resident in HBM after
 * stmmqr_factorize_device -- no download of the packed R+H.
 *   stmmqr_plan_qmult  replaces QR_qmult (STMMQR/include/SparseQR.h:403-409; SparseQR.c:1790-2020) for the methods
 *                      QR_QTX (0: X <- Q'X) and QR_QX (1: X <- Q X) on a dense m x nrhs array, in place.  Row order as in
 *                      the reference: Q'X comes back in the row order of R (HPinv applied), Q X expects it.
 *   stmmqr_plan_solve  replaces QR_solve(QR_RETX_EQUALS_B) (SparseQR.h:411-417; SparseQR.c:2024-2216 + qr_rsolve
 *                      :2218-2517): X = E * R^{-1} * (Q'B)(1:n), the solution the driver's residual check uses
 *                      (qrtest.c:11-53); dead pivot columns get x = 0 (the reference's basic solution).  The singleton
 *                      block R1 of SparseQR() is not part of the plan: it stays with the caller. */
int stmmqr_plan_qmult(stmmqr_plan *plan, int method, double *X, stm_long ldx, stm_long nrhs);
int stmmqr_plan_solve(stmmqr_plan *plan, const double *B, stm_long ldb, double *X, stm_long ldx, stm_long nrhs);
/*   stmmqr_plan_qmult also takes the methods QR_XQT (2: X <- X Q') and QR_XQ (3: X <- X Q) of qr_panel (SparseQR.c:1591-1700,
 *                      2040-2075): X is then k x m with ldx >= k and `nrhs` = k.  All vectors cross PCIe in one transfer.
 *   stmmqr_plan_rsolve replaces QR_solve for all four systems (SparseQR_definitions.h:27-30; SparseQR.c:2118-2216,
 *                      qr_rsolve :2218-2517, qr_private_rtsolve :2522): 0 QR_RX_EQUALS_B  X = R\B, 1 QR_RETX_EQUALS_B
 *                      X = E(R\B) -- B is m x nrhs in R's row order (what QR_QTX returns), X n x nrhs --, 2 QR_RTX_EQUALS_B
 *                      X = R'\B, 3 QR_RTX_EQUALS_ETB  X = R'\(E'B) -- B n x nrhs, X m x nrhs (zero beyond the rank). */
int stmmqr_plan_rsolve(stmmqr_plan *plan, int system, const double *B, stm_long ldb, double *X, stm_long ldx,
                       stm_long nrhs);

/* dense single-front kernels on host buffers (inner seams without the cc argument) */
stm_long stmmqr_front(stm_long m, stm_long n, stm_long npiv, double tol, stm_long ntol, double *F,
                      stm_long *Stair, char *Rdead, double *Tau, double *flops);
int stmmqr_larftb_qtx(stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, const double *V,
                      const double *Tau, double *C);
/* qr_larftb with a return code, all four methods (0 QR_QTX, 1 QR_QX: C is m x n and V m x k; 2 QR_XQT, 3 QR_XQ: C is m x n
 * and V n x k; SparseQR_factorize.c:1851-1904).  The reference-named export qr_larftb calls this and turns a failure into
 * cc->status < 0 + stmmqr_last_error(), never into a silent return. */
int stmmqr_larftb(int method, stm_long m, stm_long n, stm_long k, stm_long ldc, stm_long ldv, const double *V,
                  const double *Tau, double *C);
/* device time (ms, HIP events) of the kernels of the last qr_front / stmmqr_front / qr_assemble seam call on this
 * thread's device, -1 if none: the seams take host buffers, so their wall time is dominated by the copies; the kernel
 * micro-benchmarks of SURVEY.md 8(d) (bench.py --workload micro) read this instead */
double stmmqr_last_seam_ms(void);

/* ================================================================================================
 * 3. Configuration / introspection
 * ================================================================================================ */
/* Defaults: {32, 64, 0, 0, 0, 1, 0, 1, 0, 1}.  (Two options of round 3 were removed in round 4 with the kernels behind them, both
 * measured slower at every setting: mid_front_cols = whole mid-size fronts in one workgroup, pair_update = 2 = the single-sweep pair
 * update; profiles/EXPERIMENTS.md.)   Read when a plan is created (or a seam is called); the numerical results do not
 * depend on them beyond rounding.
 * Environment (diagnosis, tests and experiments only; read at plan time unless noted): STMMQR_DBG (bit mask,
 * csrc/stmmqr_kernels.h), STMMQR_QBIG_MIN (entries of a front from which Q-apply / back substitution split its rows over
 * workgroups; default 2097152), STMMQR_CHUNK (fronts per panel launch with STMMQR_DBG bit 9), STMMQR_SCHED (step order:
 * 1 level-synchronous, 2 as soon as possible, 3 envelope rule; unset: chosen per front group), STMMQR_RIDE (envelope rule:
 * factor on the height a riding front may have), STMMQR_CA_MIN (rows from which the Gram-based panel is used; 4096),
 * STMMQR_PAIR_MIN (rows from which a front takes the pair update; 16384), STMMQR_LA_MIN / STMMQR_LA_MAXPWG (look-ahead:
 * tiles of trailing update from which a step is offloaded, 2500, and the most panel workgroups such a step may have, 48;
 * read per factorization), STMMQR_SIDE_RESERVE (compute units the side stream leaves alone; 32), STMMQR_DUMPSTEPS=file
 * (detail runs: one line per step with what ran and how long); STMMQR_DBG bits of general use: 16 diagnosis counters printed
 * by stmmqr_factorize_finish (panels by actual rows, launched / useful update workgroups, refresh rounds), 16384 no
 * wave-pipelined panels (every short panel through the multi-workgroup pipeline).
 * Round 5: STMMQR_EARLY_END=0 (every panel of every front a step of the timeline; default: a front is scheduled up to the panel
 * where the full-rank row estimate says it runs out of rows) and STMMQR_EARLY_SLACK (extra panels per front, 0);
 * STMMQR_PASSENGERS=0 / STMMQR_PASS_ROWS (16384) / STMMQR_PASS_MAXWG (384) / STMMQR_CA_RIDERS=0 (passenger launches: off, the rows
 * up to which a step rides, the workgroups of its block-0 launch, riders on the Gram-based panel launch); STMMQR_PART_GRAIN / _CAP
 * (entries per workgroup, 2048, and workgroups per front, 4096, of the assembly / packing launches); STMMQR_LIVE_PANELS=0 (Q-apply /
 * back substitution visit every panel, not only those that can hold a reflector); STMMQR_RHS_BATCH (right-hand sides per pass of
 * the resident-factor operations, 32); STMMQR_NATIVE_PHASES=0 and STMMQR_EARLY_END_SHARDED=0 (sharded.py: the Python phase loop;
 * the full schedule on sharded plans). */
typedef struct stmmqr_options {
    int panel_width;        /* Householder panel width on device (<= 32); reference FCHUNK = 32                    */
    int big_front_cols;     /* fronts with fn >= this use the multi-workgroup panel/update path                     */
    int verbose;
    int use_graph;          /* replay the level schedule as a hipGraph (captured per plan and tol); 0: no gain measured */
    int panel_algo;         /* panel of the large fronts: 1 the column pipeline (one workgroup reduction per column),
                               2 the Gram-based panel (one Gram matrix per panel, any height), 0 (default) by the panel's
                               estimated rows: Gram-based above 4096 (STMMQR_CA_MIN), the column pipeline below -- whose
                               panels of at most 512 actual rows are taken by ONE workgroup, a wave per 4 columns     */
    int split_update;       /* row-parallel (2-launch) trailing update for fronts of >= 3 row slabs (1)             */
    int tall_min_rows;      /* panels with more rows than this run as a pipeline of column groups (plan time; 0)    */
    int lookahead;          /* 2 (default since round 5): one stream, the trailing update beyond the next panel's columns rides
                               on the chain's own launches as extra workgroups (passenger launches); 1: that update, the
                               packing of finished fronts and the assembly of the next ones run on a second stream beside
                               the panel chain; 0: one stream, serial order.  Same bits every way.                     */
    int fused_update;       /* 0 (default): the row-parallel trailing update is two launches (k_upd_w, k_upd_c);
                               1: ONE launch that keeps its tiles of C in registers between V'C and the application
                               (C read once, written once per panel).  Same bits either way; measured 1.1x - 2.5x slower
                               on MI355X (the slab workgroups of a column block idle while one of them adds the partial
                               sums: DESIGN.md 5), kept as an experiment.  ONE exception to "0 = two launches": an offloaded
                               look-ahead step applies T + column block 0 in one fused launch when every workgroup of that launch
                               fits the compute units the side stream leaves alone (counted with the fronts' row BOUNDS) and the
                               panels are expected to reach at most STMMQR_LA_FUSED_ROWS rows (5120; 0 turns it off).      */
    int pair_update;        /* fronts of >= 16384 rows apply the block reflectors of several consecutive panels in ONE sweep over
                               the columns beyond the next panel (their update is bound by what a sweep moves per MFMA):
                               4 (default): four panels per sweep (k_upd_wq / k_upd_yq / k_upd_cq: 0.75 passes over the trailing
                               matrix per panel; round 4: configs[4] stand-in 4127 -> 3640 ms against pairs);
                               1: two panels per sweep (k_upd_w2 / k_upd_y2 / k_upd_c2: 1.5 passes; the round-2/3 form);
                               0: every front panel by panel (3 passes).  A property of the front (plan time); it changes
                               rounding only.  (A single-sweep form of the pair, 16 instead of 24 bytes per entry and pair, was
                               built in round 3, measured slower at one wave per SIMD and removed in round 4:
                               profiles/EXPERIMENTS.md.)                                                                    */
} stmmqr_options;
void stmmqr_get_options(stmmqr_options *opt);
void stmmqr_set_options(const stmmqr_options *opt);

/* ---- the driver's Matrix Market reader ------------------------------------------------------------
 * SparseCore_read_matrix (STMMQR/src/core/SparseCore_read_write.c:982) as test/qrtest.c:112 calls it (prefer = 1): a
 * coordinate file (real / integer / pattern; general / symmetric / hermitian / skew-symmetric, or without a banner) is
 * returned as an UNSYMMETRIC CSC with both triangles, duplicates summed, columns sorted, explicit zeros kept.
 * Ap (ncol+1), Ai, Ax are malloc'ed: release with stmmqr_free.  Returns 0 or a negative STMMQR_ERR_* code
 * (stmmqr_mm_last_error() has the text); dense ("array") and complex files are refused as the driver refuses them. */
int stmmqr_read_matrix_market(const char *path, stm_long *nrow, stm_long *ncol, stm_long *nnz, stm_long **Ap,
                              stm_long **Ai, double **Ax);
const char *stmmqr_mm_last_error(void);
void stmmqr_free(void *p);

/* ---- own symbolic phase ---------------------------------------------------------------------------
 * stmmqr_analyze replaces qr_analyze (STMMQR/include/SparseQR.h; src/qr/SparseQR_analyze.c:20-700) as SparseQR() calls it
 * (SparseQR.c:338 ordering GIVEN with Q1fill, :360 FIXED on Y): the supernodal analysis of A(:,Q)'A(:,Q) (elimination
 * tree, column counts, relaxed supernodes: SparseChol_analyze_p2 / SparseChol_super_symbolic2), the frontal tree and its
 * weighted post-order, S = A(P,Q) in row form, front sizes, staircases, flop and stack bounds.  A: CSC pattern (m x n,
 * Long indices; values are not needed).  Quser: the fill-reducing column permutation (NULL = natural order).
 * relax: supernode amalgamation knobs (NULL = the library defaults 4/16/48, 0.8/0.1/0.05; stmmqr_relax_for_qr gives what
 * the driver's Relaxfactor_setting(n, nnz, RELAX_FOR_QR) sets, SparseCore_common.c:1172-1203, qrtest.c:153).
 * The result is bit-identical to the reference's qr_symbolic for the same inputs (tests/test_symbolic.py); it says
 * ntasks = ns = 1 (the reference's serial analysis): tree parallelism is this library's scheduler.
 * Host-only: no device is touched. */
typedef struct stmmqr_relax { stm_long nrelax[3]; double zrelax[3]; } stmmqr_relax;
typedef struct stmmqr_analysis stmmqr_analysis;       /* owns every array the qr_symbolic view points to */
void stmmqr_relax_for_qr(stm_long n, stm_long nnz, stmmqr_relax *relax);
int stmmqr_analyze(stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const stm_long *Quser,
                   int do_rank_detection, const stmmqr_relax *relax, stmmqr_analysis **out);
const stm_qr_symbolic *stmmqr_analysis_symbolic(const stmmqr_analysis *a);   /* borrowed, valid until stmmqr_analysis_free */
/* info[0..7] = flop bound (cc->SPQR_flopcount_bound), fl and lnz of the Cholesky analysis, QR_CHUNK_FLAG (fl/lnz >= 1000,
 * SparseChol_analyze.c:727-730), bound on nnz(R), bound on nnz(H) (SPQR_istat[0..1]), maxstack, nf */
int stmmqr_analysis_info(const stmmqr_analysis *a, double *info);
void stmmqr_analysis_free(stmmqr_analysis *a);

/* ---- SURVEY.md 8 (f2 stage 2, f4): SparseQR() without the reference ---------------------------------------------------
 * stmmqr_sparseqr replaces SparseQR (STMMQR/include/SparseQR.h:25-31; src/qr/SparseQR.c:66-450): column singletons
 * (qr_1colamd :515-1128), the fill-reducing ordering (COLAMD: src/base/colamd.c + the column elimination tree's post-order,
 * SparseChol_colamd), R1 / Y (:216-329), the symbolic analysis (stmmqr_analyze) and the numeric factorization on the device.
 * ordering: the reference's QR_ORDERING_* values (SparseQR_definitions.h:6-21): 0 FIXED, 1 NATURAL, 2 COLAMD, 3 GIVEN (Quser,
 * an extension: the reference's SparseQR has no such argument), 7 DEFAULT (= COLAMD as in the stock build); AMD (5), NESDIS (6),
 * METIS (10, 11) and the best-of strategies (4, 8, 9) are third-party packages in the reference and are refused here.
 * tol <= -2 (QR_DEFAULT_TOL): 20 (m + n) eps max_j |A(:,j)|_2 as the reference's qr_tol; -2 < tol < 0: no rank detection.  relax: see stmmqr_analyze.  Bit-identical to the reference's Q1fill / P1inv / R1 / Y / qr_symbolic
 * on the committed fixtures (tests/test_sparseqr_symbolic.py).
 * stmmqr_sparseqr_symbolic is the host half alone (no device), stmmqr_sparseqr_numeric the device half. */
typedef struct stmmqr_qr stmmqr_qr;
int stmmqr_sparseqr(int ordering, double tol, stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                    const stm_long *Quser, const stmmqr_relax *relax, int device, stmmqr_qr **out);
int stmmqr_sparseqr_symbolic(int ordering, double tol, stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai,
                             const double *Ax, const stm_long *Quser, const stmmqr_relax *relax, stmmqr_qr **out);
int stmmqr_sparseqr_numeric(stmmqr_qr *qr, int device);
void stmmqr_sparseqr_free(stmmqr_qr *qr);
/* info[0..11] = rank, n1rows, n1cols, nf, analyze seconds (ordering + analysis: Ana_time), factorize seconds (the qr_factorize
 * interval: Fac_time), flops (the reference's count), flop bound, device ms of the factorization, ordering used,
 * QR_CHUNK_FLAG, retries */
int stmmqr_sparseqr_info(const stmmqr_qr *qr, double *info);
double stmmqr_sparseqr_tol(const stmmqr_qr *qr);                                   /* the tolerance really used: qr_tol(A) for QR_DEFAULT_TOL, -1 (EMPTY) for other negative ones (SparseQR.c:126-139) */
const stm_long *stmmqr_sparseqr_q1fill(const stmmqr_qr *qr);                      /* [n] column permutation (singletons first) */
const stm_qr_symbolic *stmmqr_sparseqr_symbolic_view(const stmmqr_qr *qr);       /* qr_symbolic of A or Y */
stmmqr_plan *stmmqr_sparseqr_plan(stmmqr_qr *qr);                                 /* the device plan holding the factors */
int stmmqr_sparseqr_y(const stmmqr_qr *qr, const stm_long **Yp, const stm_long **Yi, const double **Yx);
/* QR_qmult (SparseQR.h:403-409): X nrow x ncol (m x k for methods 0 QR_QTX / 1 QR_QX, k x m for 2 QR_XQT / 3 QR_XQ), result
 * in Y (same shape).  QR_solve (SparseQR.h:411-417): systems 0..3 as in stmmqr_plan_rsolve, singleton rows included. */
int stmmqr_sparseqr_qmult(stmmqr_qr *qr, int method, const double *X, stm_long ldx, stm_long nrow, stm_long ncol, double *Y,
                          stm_long ldy);
int stmmqr_sparseqr_solve(stmmqr_qr *qr, int system, const double *B, stm_long ldb, stm_long nrhs, double *X, stm_long ldx);

/* ---- R / H out in sparse form, LQ (STMMQR/src/qr/SparseLQ.c; prototypes SparseQR.h:282-340) ---------
 * qr_rcount / qr_rconvert / qr_trapezoidal keep the reference's names, argument lists and results (bit for bit: they move
 * data).  They read a qr_numeric on the host -- the one qr_factorize returned.  stmmqr_plan_export_r does the same for the
 * factorization a plan holds in HBM (one download): R as compressed sparse columns over the columns of the factorized matrix,
 * optionally H (one column per live reflector, unit diagonal stored, rows = the permuted row ids) and its Tau; every output
 * array is malloc'ed (stmmqr_free).  stmmqr_sparselq = SparseLQ: the QR object of A' (L = R'). */
void qr_rcount(stm_qr_symbolic *QRsym, stm_qr_numeric *QRnum, stm_long n1rows, stm_long econ, stm_long n2, int getT, stm_long *Ra,
               stm_long *Rb, stm_long *H2p, stm_long *p_nh);
void qr_rconvert(stm_qr_symbolic *QRsym, stm_qr_numeric *QRnum, stm_long n1rows, stm_long econ, stm_long n2, int getT,
                 stm_long *Rap, stm_long *Rai, double *Rax, stm_long *Rbp, stm_long *Rbi, double *Rbx, stm_long *H2p, stm_long *H2i,
                 double *H2x, double *H2Tau);
stm_long qr_trapezoidal(stm_long n, stm_long *Rp, stm_long *Ri, double *Rx, stm_long bncols, stm_long *Qfill,
                        int skip_if_trapezoidal, stm_long **p_Tp, stm_long **p_Ti, double **p_Tx, stm_long **p_Qtrap,
                        stm_sparse_common *cc);
int stmmqr_plan_export_r(stmmqr_plan *plan, const stm_qr_symbolic *sym, stm_long econ, stm_long **Rp, stm_long **Ri, double **Rx,
                         stm_long *nh, stm_long **Hp, stm_long **Hi, double **Hx, double **HTau);
int stmmqr_sparselq(int ordering, double tol, stm_long m, stm_long n, const stm_long *Ap, const stm_long *Ai, const double *Ax,
                    const stmmqr_relax *relax, int device, stmmqr_qr **out);

void stmmqr_shutdown(void);                           /* optional end-of-use call for dlopen()ing hosts: device sync  */
/* Device buffers for hosts without HIP bindings of their own (FFI callers of stmmqr_export_front_dev /
 * stmmqr_import_front_dev, device-resident A values): allocated by the HIP runtime THIS library is bound to, on the
 * current device.  0 or a negative STMMQR_ERR_* code. */
int stmmqr_device_alloc(size_t bytes, void **ptr);
int stmmqr_device_free(void *ptr);
int stmmqr_device_count(void);                        /* number of visible HIP devices (0 = none)          */
const char *stmmqr_device_name(int device);           /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"        */
const char *stmmqr_last_error(void);                  /* thread-local message of the last failure          */
const char *stmmqr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STMMQR_HIP_H */
