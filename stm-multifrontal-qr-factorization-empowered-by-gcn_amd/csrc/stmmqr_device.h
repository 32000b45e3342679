// stmmqr_device.h -- structures shared by the host planner and the gfx950 kernels.
//
// Device-side view of one front of the multifrontal QR (reference: qr_kernel's per-front loop,
// STMMQR/src/qr/SparseQR_factorize.c:886-974).  All indices are 32-bit on the device (rjsize, hisize,
// anz < 2^31 is checked by the planner); arena offsets are 64-bit.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define STM_NB 32            // Householder panel width (reference FCHUNK = 32, qrtest.c:152)
#define STM_BIGROW 0x3fffffff

// symbolic, immutable after planning
struct FrontSym {
    long long foff;          // offset (doubles) of F in the front arena; F is column-major, ld rows
    long long coff;          // offset (doubles) of the packed contribution block in the C arena
    int ld;                  // leading dimension of F (>= fm upper bound, even)
    int fn, fp;              // columns, pivotal columns            (Rp[f+1]-Rp[f], Super[f+1]-Super[f])
    int col1;                // first pivotal column                (Super[f])
    int rp;                  // Rp[f]: slot of this front in Rj / HStair / HTau / Rjrel / Cmap
    int hip;                 // Hip[f]: slot in Hii
    int child0, child1;      // children = Child[child0 .. child1)
    int srow0, srow1;        // rows of S assembled here = [Sleft[col1], Sleft[col1+fp])
    int fm_ub;               // symbolic upper bound on the number of rows
    int npanels;             // ceil(fn / STM_NB)
    int parent;              // parent front or -1
    int fm_est;              // rows of F if no pivot column dies (exact for full-rank fronts): launch planning only
    int tpan;                // index of this front's first panel in the kept-T array (DevCtx::Tall), panel p at tpan + p
    int nsched;              // panels of this front that get a step of the timeline (<= npanels: build_schedule, "how many panels")
    int qbig;                // Q-apply on the resident factors: the rows of this front are split over workgroups, one launch
                             // per panel (k_qbig_*), instead of one workgroup for the whole front (plan time)
};

// one pending block reflector (written by the panel kernel, read by the update kernel)
#define STM_PD_RING 4                                           // panel descriptions and T factors kept per front
#define STM_PDI(p) ((p) & (STM_PD_RING - 1))
#define STM_TSLOT(slot, p) ((long long)STM_PD_RING * (slot) + STM_PDI(p))
struct PanelDesc {
    int pg1, pt;             // rows [pg1, pt)
    int pk1, pnb;            // built from columns [pk1, pk1+pnb)
    int pc0;                 // to be applied to columns [pc0, fn)
    int mode;                // header of the panel pipeline: 1 the column groups run, 0 nothing to do / whole panel by group 0
    int pdiag[STM_NB];       // row of the unit diagonal of each reflector (BIGROW: none)
    // tall-panel pipeline state (sub-panels of STM_SW columns, one launch per sub-panel)
    int tmax;                // rows [pg1, tmax) are touched by the panel
    int nlive;               // live reflectors so far
    int sw;                  // sub-panel width of this panel: 8; 4 / 2 when the rows need 8 / 16 registers per thread and column
    int done_group;          // group that ran out of rows (g reached fm), -1 if none
    int t_deferred;          // 1: the panel kernel left T to k_upd_w (Gram block + the last slab workgroup builds T);
                             // 2: last panel of a front (no trailing update): T is built by k_cpack's extra workgroup
    int pad3;
    int sg[STM_NB / 2];      // first active row (g) at the start of sub-panel s
    int st[STM_NB / 2];      // one past the last row reached by the reflectors of sub-panel s
    double lensum;           // sum of (t - g) over the live columns so far (flop accounting of the trailing update)
};

#define STM_SW 8             // sub-panel width of the tall-panel pipeline (4 above STM_TALL_WIDE rows)
#define STM_TALL_MIN 0       // default of stmmqr_options::tall_min_rows: panels with more (estimated) rows take the pipeline
                             // (0: every panel of a large front; the one-workgroup LDS panel is kept for the small-front kernel)
#define STM_TALL_NTH 512     // threads of the panel kernel
#define STM_TALL_MAX (16 * STM_TALL_NTH)  // rows a sub-panel can hold in registers (16 per thread)
#define STM_TALL_WIDE (4 * STM_TALL_NTH)  // more rows than this: 4-column sub-panels (64 doubles of register image)
#define STM_TALL_XWIDE (8 * STM_TALL_NTH) // more rows than this: 2-column sub-panels
#define STM_WP_ROWS 512      // panels whose staircase reaches at most this many rows are factorized by ONE workgroup, a wave per
                             // 4 columns, with the panel's image in LDS (32 x 512 doubles: every pipeline launch carries it)
#define STM_UPD_LDS_HOST (2 * 32 * 66 + 32 * 33)   // = STM_UPD_LDS_DOUBLES of the kernels (one half's chunk images of dev_update_block)
#define STM_PROG 64          // FrontNum::prog advances by this much per panel (2 per column group + 1, <= 16 groups)

// Does panel p of this front take the tall-panel pipeline?  Planned on the host (number of launches) and re-evaluated
// on the device from the same symbolic data, so both always agree.
static inline __host__ __device__ int stm_tall_panel(const FrontSym &s, int p, int tall_min)
{
    int g = p * STM_NB;
    if (g > s.fp) g = s.fp;
    if (g > s.fm_est) g = s.fm_est;
    return s.fm_est - g > tall_min;
}
// estimated rows of panel p (exact for full-rank fronts)
static inline __host__ __device__ int stm_panel_rows_est(const FrontSym &s, int p)
{
    int g = p * STM_NB;
    if (g > s.fp) g = s.fp;
    if (g > s.fm_est) g = s.fm_est;
    return s.fm_est - g;
}
// Which panel kernel factorizes panel p of this front?  A property of the front alone (symbolic), never of the fronts that
// happen to share its level: the two kernels round differently and sharded == unsharded must stay bit-identical.
// algo = stmmqr_options::panel_algo: 1 column pipeline, 2 Gram-based, 0 Gram-based above STM_TALL_WIDE rows.
#define STM_CA_MIN_ROWS (8 * 512)      // (= STM_TALL_XWIDE: above it the pipeline drops to 2-column groups, above 8192 rows to one workgroup;
                                       //  measured: up to 4096 rows the pipeline is the faster one)
static inline __host__ __device__ int stm_use_ca(const FrontSym &s, int p, int algo, int min_rows = STM_CA_MIN_ROWS)
{
    return algo == 2 || (algo == 0 && stm_panel_rows_est(s, p) > min_rows);
}
// planned number of column groups for panel p: 1 (not tall), 4, or 8 / 16 when 4- / 2-column sub-panels may be needed
static inline __host__ __device__ int stm_tall_launches(const FrontSym &s, int p, int tall_min)
{
    if (!stm_tall_panel(s, p, tall_min)) return 1;
    int g = p * STM_NB;
    if (g > s.fp) g = s.fp;
    if (g > s.fm_est) g = s.fm_est;
    const int rows = s.fm_est - g;
    return (rows > STM_TALL_XWIDE) ? STM_NB / 2 : (rows > STM_TALL_WIDE) ? STM_NB / 4 : STM_NB / STM_SW;
}

// dynamic LDS (doubles) of the one-workgroup panel of this front: the padded rows of a whole panel, capped at 128 KiB
#define STM_LDS_CAP_DOUBLES 16384
static inline __host__ __device__ int stm_front_lds(const FrontSym &s)
{
    const long need = (long)(((s.fm_ub + 63) & ~63) | 1) * STM_NB + 64;
    return need < STM_LDS_CAP_DOUBLES ? (int)need : STM_LDS_CAP_DOUBLES;
}

// Gram-based panel (stmmqr_capanel.hip): rows of the panel below its pivot rows are cut into slabs of STM_CA_R rows, one
// workgroup each.  The number of slab workgroups of a front is symbolic (every workgroup of a launch must agree on it).
#define STM_CA_R 480
static inline __host__ __device__ int stm_ca_slabs(const FrontSym &s)
{
    const int nb = (s.fm_ub - 1 + STM_CA_R - 1) / STM_CA_R;
    return nb < 1 ? 1 : nb;
}

#define STM_UPD_SLAB 256     // rows of a workgroup tile of the row-parallel trailing update (k_upd_w / k_upd_c); W is accumulated
                             // per slab and the slabs are added in order in BOTH update forms (bit-identical results)
// the row-parallel trailing update's workspace slice of a front at panel p: (ncb + 1) column-block slots (trailing
// blocks + the Gram block) of nsl slabs of NB x 32 partial sums; both counts are symbolic (host sizing = device indexing)
static inline __host__ __device__ int stm_upd_ncb(const FrontSym &s, int p)
{
    int k2 = (p + 1) * STM_NB;
    if (k2 > s.fn) k2 = s.fn;
    return (s.fn - k2 + 31) / 32;
}
static inline __host__ __device__ int stm_upd_nsl(const FrontSym &s) { return (s.fm_ub + STM_UPD_SLAB - 1) / STM_UPD_SLAB; }

// Pair update: a workgroup of k_upd_w2 / k_upd_c2 takes 1, 2 or 4 slabs of the pair's rows (stm_pair_spw), so a column block has
// at most this many partial sums: the stride of its slots in the workspace (host sizing = device indexing; nslf = slabs of the
// front's row bound; `tune` -- env STMMQR_TUNE at plan time, measurement sweeps -- may force one slab per workgroup: full stride)
static inline __host__ __device__ int stm_pair_slots(int nslf, int tune)
{
    if (tune & 15) return nslf;
    return nslf < 16 ? nslf : (nslf + 3) / 4 > 16 ? (nslf + 3) / 4 : 16;
}
// quad update: a workgroup takes up to 8 slabs (stm_quad_spw: 8 from 64 slabs on, else the pair's rule), so at most this many partials
static inline __host__ __device__ int stm_quad_slots(int nslf, int tune)
{
    if (tune & 15) return nslf;
    if (tune >> 8) return stm_pair_slots(nslf, 0);               // (another 8-slab threshold, measurement sweeps: the pair's stride covers it)
    return nslf < 16 ? nslf : (nslf + 7) / 8 > 16 ? (nslf + 7) / 8 : 16;
}
#define STM_PAIR_MIN_ROWS 16384  // fronts with at least this many (estimated) rows take the pair update (stmmqr_options::pair_update);
                                 // measured: 27 000 rows -12 %, 7818 rows +17 % (and no look-ahead for pair steps)

#define STM_QB_ROWS 512      // rows of a front per workgroup of the split Q-apply (k_qbig_step)
// split Q-apply (k_qbig_*): one entry per large front of a tree level
struct QbDesc {
    int f;                   // front
    int xoff, dqoff, wqoff;  // offsets of its slices of the level's x (doubles), reflector numbering (ints), slab partials
    int nslab;               // row slabs of QB_ROWS rows (from fm_ub)
    int np_live;             // panels that can hold a live reflector in THIS factorization (<= npanels; set per factorization from the
                             // front's fm / rank: the panels behind the column where the rows ran out are not visited)
};

// numeric, written by the kernels
struct FrontNum {
    int fm;                  // rows of F                           (qr_fsize)
    int g;                   // rows eliminated so far = next diagonal row
    int rank;                // live pivotal columns                (qr_front's return value)
    int done;                // 1 once g reached fm and the tail columns were finalised
    int cm;                  // rows of the contribution block      (qr_cpack's return value)
    int pad_rs;
    int hdr;                 // tall-panel pipeline: p+1 once the header (mode, pg1, tmax, sw) of panel p is published
    int prog;                // ... STM_PROG*p + 2*(finished groups of panel p) + (1: first half of the next one); monotone
    int perr;                // ... set when a bounded wait ran out (the factorization is reported as failed)
    int gcnt;                // arrival counter of the Gram slabs in k_upd_w (back to 0 after every panel)
    int tready;              // fused update: step + 1 once T of the panel of that step is in its slot (Gram block's last slab)
    long long rsize;         // entries of the packed R+H block     (qr_rhpack's return value; 2.2e9 at the largest size)
    double flops;            // reference flop count of this front  (FLOP_COUNT, :1571)
    double flops_upd;        // the part of `flops` that the trailing update does: sum (t-g) * 4 * (fn - k2), k2 = panel end
    // pending block reflectors, a ring by panel number: the look-ahead schedule factorizes panel p+1 while the tail of update p
    // is still reading the description of panel p, and the sweeps of the large fronts (pair / quad update) read the last 2 / 4
    PanelDesc pd[STM_PD_RING];
};
