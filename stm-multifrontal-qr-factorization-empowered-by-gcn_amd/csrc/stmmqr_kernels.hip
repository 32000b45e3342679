// stmmqr_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the multifrontal-QR numeric phase.
//
// One file = the whole device side of the reference's hot loop (STMMQR/src/qr/SparseQR_factorize.c):
//   k_gather_sx      <- qr_stranspose2          (:755-785)   S = A(P,Q) values, pure gather
//   k_setup          <- qr_fsize + the integer half of qr_assemble (:1066-1145, :1239-1248, :1205)
//   k_assemble       <- qr_assemble             (:1151-1285) scatter of S rows and packed child C blocks
//   dev_panel        <- qr_front's column loop  (:1434-1609) + dlarft (T factor of the block reflector)
//   dev_update_block <- qr_larftb / dlarfb      (:1851-1904) C -= V (T' (V' C)) on v_mfma_f64_16x16x4_f64
//   k_front_wg       one workgroup factorizes a whole (small) front; k_panel / k_update: large fronts
//   dev_cpack        <- qr_cpack                (:1639-1685)
//   k_rh_count / k_rh_scan / k_rh_copy <- qr_rhpack (:1691-1784) + the stack compaction of qr_factorize (:597-701)
//
// Design notes (DESIGN.md has the long form):
//  * fronts are column-major with a fixed leading dimension; a front's rows are known only on the device
//    (dead pivot columns change them), so every kernel reads FrontNum for its extents;
//  * the tall-skinny panel is staged in LDS when it fits (<= lds_doubles), Householder norms and v'C dot
//    products use wave64 shuffles + one cross-wave LDS step;
//  * the block reflector is applied with fp64 MFMA tiles (16x16x4): W = V'C over 64-row chunks, W = T'W,
//    C -= V W; V is read in place from F with the unit-diagonal/zero mask applied on the fly.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include "stmmqr_device.h"
#include "stmmqr_kernels.h"
#include "stmmqr_wave.h"
#include "stmmqr_devutil.h"

// small LDS working set of the panel routines, declared once per kernel and shared by every code path
struct PanelShared {
    double red[16];
    int st_out[STM_NB];        // Stair / dead flag of the panel columns, flushed to global once per panel (a global
    int dead[STM_NB];          //  store inside the column loop makes every barrier wait for its completion)
    double nextss;             // |x|^2 of the next column, produced by the wave that just updated it
    int nextss_col;            // ... valid for this panel column (-1: none)
    int stair[STM_NB];
    int diag[STM_NB];
    double tau[STM_NB];
    double G[STM_NB][STM_NB + 1];
    double T[STM_NB][STM_NB + 1];
    double part[8 * 32];
    double top[2][8];
    double rsum[2][64];        // register sub-panel: per-wave sums of the 8 reductions of a column step (two buffers)
    double rsumB[5][64];       // blocked application of a half group of reflectors: up to 40 sums in one exchange
    double Ts[8][9];
    double gp[32];
};

#define NT 256
#define NW (NT / 64)
#define BN 32                 // trailing-update column block
#define RB 64                 // trailing-update row chunk
#define VS (RB + 2)           // LDS row stride of the V / C chunk images (doubles)
#define WS (BN + 1)

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// sum over the whole workgroup of NTH threads; every thread gets the result.  s_red: NTH/64 doubles of LDS.
template <int NTH>
__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    v = wave_sum(v);
    __syncthreads();                       // protect s_red from the previous use
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < NTH / 64; w++) r += s_red[w];
    return r;
}

// inclusive scan of one int per thread across the workgroup; *total = sum.  s_scan: NW ints.
__device__ __forceinline__ int block_incl_scan(int v, int *s_scan, int *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) s_scan[wid] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        int sw = s_scan[w];
        if (w < wid) base += sw;
        tot += sw;
    }
    *total = tot;
    return x + base;
}

// ------------------------------------------------------------------------------------------------
// magnitude guard: sig = {sg, 1/sg}, sg = 2^-e when max|A| = 2^e lies beyond 2^+-300 (so that the sums of squares of the
// panel kernels stay representable: stm_larfg_guarded), else 1.  No host involvement: the values may be device resident.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_amax(const double *__restrict__ Ax, int anz, unsigned long long *amaxbits)
{
    double mx = 0;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < anz; s += gridDim.x * 256) {
        const double a = fabs(Ax[s]);
        if (a > mx && a <= 1.7976931348623157e308) mx = a;                  // (infinities / NaNs are not a scale)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(amaxbits, (unsigned long long)__double_as_longlong(mx));   // (order of positive doubles = order of their bits)
}
__global__ void k_sigma(const unsigned long long *amaxbits, double *sig)
{
    const double amax = __longlong_as_double((long long)*amaxbits);
    double sg = 1.0;
    if (amax > 0) {
        const int e = ilogb(amax);
        if (e > 300 || e < -300) sg = ldexp(1.0, -e);
    }
    sig[0] = sg; sig[1] = 1.0 / sg;
}

// ------------------------------------------------------------------------------------------------
// qr_stranspose2: Sx[s] = Ax[smap[s]]   (smap is symbolic: planner, from Ap/Ai/Qfill/PLinv/Sp)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_sx(const double *__restrict__ Ax, const int *__restrict__ smap,
                                                   double *__restrict__ Sx, int anz)
{
    for (int s = blockIdx.x * 256 + threadIdx.x; s < anz; s += gridDim.x * 256) Sx[s] = Ax[smap[s]];
}

// ------------------------------------------------------------------------------------------------
// qr_fsize + row bookkeeping of qr_assemble.  One workgroup per front of the level.
//   Stair[j]  <- one past the last row whose leftmost column is <= j   (the staircase qr_front consumes)
//   Cmap[..]  <- row of the parent that receives row ci of child c      (qr_assemble :1239-1248)
//   Hii[..]   <- S-row ids of the front's rows                          (:1205, :1246)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_setup(DevCtx c, const int *__restrict__ flist)
{
    __shared__ int s_scan[NW];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int tid = threadIdx.x;
    int *St = c.Stair + s.rp;
    int *Cur = c.Cursor + s.rp;
    const int *Sl = c.Sleft + s.col1;

    for (int j = tid; j < s.fn; j += NT) St[j] = (j < s.fp) ? Sl[j + 1] - Sl[j] : 0;
    __syncthreads();
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const int cm = c.fnum[ch].cm;
        const int pc = c.fs[ch].rp + c.fs[ch].fp;
        for (int ci = tid; ci < cm; ci += NT) atomicAdd(&St[c.Rjrel[pc + ci]], 1);
    }
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < s.fn; base += NT) {
        const int j = base + tid;
        // the counts were built with L2 atomics: read them past the (possibly stale) vector L1
        const int v = (j < s.fn) ? __hip_atomic_load(&St[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int tot;
        const int incl = block_incl_scan(v, s_scan, &tot);
        if (j < s.fn) {
            St[j] = carry + incl;
            Cur[j] = carry + incl - v + ((j < s.fp) ? Sl[j + 1] - Sl[j] : 0);
        }
        carry += tot;
    }
    __syncthreads();
    const int fm = carry;
    int *Hi = c.Hii + s.hip;
    for (int r = s.srow0 + tid; r < s.srow1; r += NT) {
        const int k = c.Sj0[r] - s.col1;
        const int i = (k > 0 ? St[k - 1] : 0) + (r - Sl[k]);
        Hi[i] = r;
    }
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const int cm = c.fnum[ch].cm;
        const int pc = c.fs[ch].rp + c.fs[ch].fp;
        const int *Hic = c.Hii + c.fs[ch].hip + c.fnum[ch].rank;
        for (int ci = tid; ci < cm; ci += NT) {
            const int j = c.Rjrel[pc + ci];      // distinct for distinct ci of one child: no race
            const int i = Cur[j];
            Cur[j] = i + 1;
            c.Cmap[pc + ci] = i;
            Hi[i] = Hic[ci];
        }
        __syncthreads();
    }
    if (tid == 0) {
        FrontNum *nm = &c.fnum[f];
        nm->fm = fm; nm->g = 0; nm->rank = min(fm, s.fp); nm->done = 0; nm->hdr = 0; nm->prog = 0; nm->perr = 0; nm->gcnt = 0;
        for (int q = 0; q < STM_PD_RING; q++) nm->pd[q].pnb = 0; nm->cm = 0; nm->rsize = 0; nm->flops = 0; nm->flops_upd = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// qr_assemble: scatter S rows and the children's packed C blocks into the (pre-zeroed) front.
// grid = (max parts, fronts of the level); part p of front f handles every nparts-th 256-element slab.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_assemble(const DevCtx &c, const FrontSym &s, int part, int nparts)
{
    const int tid = threadIdx.x;
    double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int *St = c.Stair + s.rp;
    const int *Sl = c.Sleft + s.col1;
    for (int r = s.srow0 + part * NT + tid; r < s.srow1; r += nparts * NT) {
        const int k = c.Sj0[r] - s.col1;
        const int i = (k > 0 ? St[k - 1] : 0) + (r - Sl[k]);
        for (int p = c.Sp[r]; p < c.Sp[r + 1]; p++) F[i + c.Sjrel[p] * ld] = c.Sx[p];
    }
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const int cm = c.fnum[ch].cm;
        if (cm <= 0) continue;
        const FrontSym cs = c.fs[ch];
        const int cn = cs.fn - cs.fp;
        const int pc = cs.rp + cs.fp;
        const long long tri = (long long)cm * (cm + 1) / 2;
        const long long csize = tri + (long long)cm * (cn - cm);
        const double *C = c.Carena + cs.coff;
        const int *cmap = c.Cmap + pc;
        const int *crel = c.Rjrel + pc;
        for (long long e = (long long)part * NT + tid; e < csize; e += (long long)nparts * NT) {
            int cj, ci;
            if (e < tri) {
                cj = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while ((long long)cj * (cj + 1) / 2 > e) cj--;
                while ((long long)(cj + 1) * (cj + 2) / 2 <= e) cj++;
                ci = (int)(e - (long long)cj * (cj + 1) / 2);
            } else {
                const long long r = e - tri;
                cj = cm + (int)(r / cm);
                ci = (int)(r % cm);
            }
            F[cmap[ci] + crel[cj] * ld] = C[e];
        }
    }
}

__global__ __launch_bounds__(NT) void k_assemble(DevCtx c, const int *__restrict__ flist,
                                                 const int *__restrict__ nparts_list)
{
    const int fi = blockIdx.y;
    const int nparts = nparts_list[fi];
    if ((int)blockIdx.x >= nparts) return;
    const FrontSym s = c.fs[flist[fi]];
    dev_assemble(c, s, blockIdx.x, nparts);
}

// One 64-row chunk of V (panel columns, unit-lower-trapezoidal mask applied) and of C (one column block) goes through
// registers into LDS.  The loads are unconditional on clamped indices (a predicated load is a branch around each
// access, and the 16 loads of a thread would be issued one round trip at a time); the callers issue the loads of the
// next chunk before the MFMA loop of the current one.
struct UpdChunk { double v[8], c[8]; };
__device__ __forceinline__ void upd_chunk_load(UpdChunk &ck, const double *Vg, const double *Cg, long long ld, int i, int mp,
                                               int nbp, int nc, int lcg)
{
    const int ic = min(i, mp - 1);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        ck.v[q] = Vg[ic + (long long)min(col, nbp - 1) * ld];
        ck.c[q] = Cg[ic + (long long)min(col, nc - 1) * ld];
    }
}
__device__ __forceinline__ void upd_chunk_to_lds(const UpdChunk &ck, int i, int mp, int nbp, int nc, const int *s_pd, int g1,
                                                 int lrow, int lcg, double *Vs, double *Cs, bool c_is_v = false, bool plain = false)
{
    if (plain) {                                               // (uniform: a chunk below every unit diagonal, inside the panel's rows,
#pragma unroll                                                 //  full blocks -- nothing to mask: the same values by a plain copy)
        for (int q = 0; q < 8; q++) {
            const int col = lcg * 8 + q;
            Vs[col * VS + lrow] = ck.v[q];
            Cs[col * VS + lrow] = c_is_v ? ck.v[q] : ck.c[q];
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        const int d = s_pd[col] - g1;                          // (BIGROW beyond nbp: everything masked)
        const double v = (i < mp && col < nbp && i >= d) ? ((i == d) ? 1.0 : ck.v[q]) : 0.0;
        Vs[col * VS + lrow] = v;
        Cs[col * VS + lrow] = c_is_v ? v : ((i < mp && col < nc) ? ck.c[q] : 0.0);     // (Gram block: C = V)
    }
}

__device__ __forceinline__ void upd_chunk_v_to_lds(const UpdChunk &ck, int i, int mp, int nbp, const int *s_pd, int g1, int lrow,
                                                   int lcg, double *Vs, bool plain = false)
{
    if (plain) {
#pragma unroll
        for (int q = 0; q < 8; q++) Vs[(lcg * 8 + q) * VS + lrow] = ck.v[q];
        return;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        const int d = s_pd[col] - g1;
        Vs[col * VS + lrow] = (i < mp && col < nbp && i >= d) ? ((i == d) ? 1.0 : ck.v[q]) : 0.0;
    }
}
// last row (relative to g1) below which a chunk of a full panel needs no masks; BIGROW: never (a dead reflector, a short panel)
__device__ __forceinline__ int upd_plain_from(const int *s_pd, int g1, int lane)
{
    const int dm = wave_max_int(lane < STM_NB ? s_pd[lane] : -1);
    return dm >= STM_BIGROW ? STM_BIGROW : dm - g1;
}

// ------------------------------------------------------------------------------------------------
// qr_larftb(QR_QTX): C <- (I - V T V')' C for one block of <= BN columns, on fp64 MFMA.
//   C = F(g1:g1+mp, c0:c0+nc), V = F(g1:g1+mp, k1:k1+nbp) with the unit diagonal of reflector j at absolute row
//   diag[j] (STM_BIGROW: no reflector) and zeros above it, T = NB x NB upper triangular (column-major, ld NB).
//   lds: >= 2*BN*VS + STM_NB*WS doubles.  diag / T may live in LDS or global memory.
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md 3): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[i=(l>>4)+4r][j=l&15].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_T_from_gram(double (*G)[STM_NB + 1], double (*T)[STM_NB + 1], const double *tau, int nc, int tid);

// W2 = T' W1 (T upper triangular, NB x NB; W1 NB x BN) on the matrix cores: wave `wid` (0..3) gets the 16 x 16 tile (wid >> 1, wid & 1)
// of W2 in the MFMA result layout, W2[16 (wid >> 1) + (lane >> 4) + 4 r][16 (wid & 1) + (lane & 15)] = result[r].  T(q, l) is read at
// Tm[q * sq + l * sl] (any memory; it must hold zeros below the diagonal and beyond the panel's reflectors), W1[q][x] at W1[q * WS + x]
// (LDS).  Rows l < 16 only see q < 16 (the skipped products are exact zeros).  EVERY form of the trailing update forms W2 here, so
// that a front gets the same bits whichever form its step uses (the scalar loops this replaces were 2.7 us of every update
// workgroup's prologue: 160 LDS reads per thread).
__device__ __forceinline__ d4 dev_w2_tile(const double *Tm, int sq, int sl, const double *W1, int wid, int lane)
{
    const int mi = wid >> 1, ni = wid & 1, l15 = lane & 15, l4 = lane >> 4;
    d4 acc = {0, 0, 0, 0};
    const int nk = mi ? STM_NB / 4 : STM_NB / 8;
#pragma unroll
    for (int kk = 0; kk < STM_NB / 4; kk++) {
        if (kk < nk) {
            const double a = Tm[(4 * kk + l4) * sq + (16 * mi + l15) * sl];       // A[i = l][k = q] = T(q, l)
            const double b = W1[(4 * kk + l4) * WS + 16 * ni + l15];               // B[k = q][j = x]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
    return acc;
}

// tau != nullptr: T was left to the update by the panel kernel (PanelDesc::t_deferred) -- G = V'V is accumulated beside
// W1 (per 256-row slab, slabs added in order: bit-identical to the Gram block of k_upd_w) and T is built here by every
// workgroup for itself (dev_T_from_gram); Tout / Tkeep (may be null) receive it from the caller's first column block.
// TN = true (qr_larftb seam, method QR_QX only): C <- (I - V T V') C, i.e. W2 = T W1 instead of T' W1.
#define STM_UPD_LDS_DOUBLES (2 * BN * VS + STM_NB * WS)
static_assert(STM_UPD_LDS_DOUBLES == STM_UPD_LDS_HOST, "host sizing of the update kernels' LDS");
template <bool TN = false>
__device__ void dev_update_block(double *F, long long ld, int g1, int mp, int k1, int nbp, const int *diag,
                                 const double *T, int c0, int nc, double *lds, const double *tau = nullptr,
                                 double *Tout = nullptr, double *Tkeep = nullptr)
{
    if (nbp <= 0 || mp <= 0 || nc <= 0) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;

    double *Vs = lds;                       // [STM_NB][VS]
    double *Cs = Vs + STM_NB * VS;          // [BN][VS]
    double *Ws = Cs + BN * VS;              // [STM_NB][WS]
    __shared__ int s_pd[STM_NB];

    __syncthreads();                        // previous users of lds are done
    if (tid < STM_NB) s_pd[tid] = (tid < nbp) ? diag[tid] : STM_BIGROW;
    __syncthreads();

    const int lrow = tid & 63, lcg = tid >> 6;          // loader mapping: 64 rows x 4 column groups of 8
    const double *Vg = F + g1 + (long long)k1 * ld;
    double *Cg = F + g1 + (long long)c0 * ld;

    // ---- phase 1: W1 = V' C ----
    const int mi = wid >> 1, ni = wid & 1;
    // W1 is accumulated per slab of 256 rows and the slabs are added in order -- exactly the association of the
    // row-parallel form (k_upd_w partials summed by k_upd_c), so that a front gets bit-identical results whichever of the
    // two its level happens to use
    const bool build_t = (tau != nullptr);
    d4 acc = {0, 0, 0, 0}, tot = {0, 0, 0, 0}, gacc = {0, 0, 0, 0}, gtot = {0, 0, 0, 0};
    UpdChunk ck;
    upd_chunk_load(ck, Vg, Cg, ld, lrow, mp, nbp, nc, lcg);
    for (int r0 = 0; r0 < mp; r0 += RB) {
        const int i = r0 + lrow;
        upd_chunk_to_lds(ck, i, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
        __syncthreads();
        if (r0 + RB < mp) upd_chunk_load(ck, Vg, Cg, ld, i + RB, mp, nbp, nc, lcg);
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk++) {
            const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
            const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            if (build_t) {
                const double bv = Vs[(16 * ni + l15) * VS + 4 * kk + l4];
                gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, gacc, 0, 0, 0);
            }
        }
        if ((r0 + RB) % STM_UPD_SLAB == 0 || r0 + RB >= mp) {   // end of a slab (SLAB of the row-parallel form)
#pragma unroll
            for (int r = 0; r < 4; r++) { tot[r] += acc[r]; acc[r] = 0; gtot[r] += gacc[r]; gacc[r] = 0; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; r++) Ws[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = tot[r];
    double *s_Tm = Cs;                       // T(q, l) at s_Tm[q * WS + l] when it is built here (the chunk images are free)
    if (build_t) {
        double (*Gm)[STM_NB + 1] = reinterpret_cast<double (*)[STM_NB + 1]>(Vs);
        __shared__ double s_tau_u[STM_NB];
#pragma unroll
        for (int r = 0; r < 4; r++) Gm[16 * mi + l4 + 4 * r][16 * ni + l15] = gtot[r];
        if (tid < STM_NB) s_tau_u[tid] = (tid < nbp) ? tau[tid] : 0.0;
        __syncthreads();
        dev_T_from_gram(Gm, reinterpret_cast<double (*)[STM_NB + 1]>(s_Tm), s_tau_u, nbp, tid);
        if (Tout || Tkeep)
            for (int e = tid; e < STM_NB * STM_NB; e += (int)blockDim.x) {
                const int a = e % STM_NB, b = e / STM_NB;
                const double tv = (a <= b && a < nbp && b < nbp) ? s_Tm[a * WS + b] : 0.0;
                if (Tout) Tout[e] = tv;
                if (Tkeep) Tkeep[e] = tv;
            }
    }
    __syncthreads();

    // ---- phase 2: W2 = T' W1 (T upper triangular) ----
    {
        const int l = tid & 31, cg = tid >> 5;          // 8 groups x 4 columns
        double w2[4] = {0, 0, 0, 0};
        if (TN) {
            for (int q = l; q < nbp; q++) {                // row l of the upper triangular T
                const double tq = build_t ? s_Tm[l * WS + q] : T[l + q * STM_NB];
#pragma unroll
                for (int x = 0; x < 4; x++) w2[x] += tq * Ws[q * WS + cg * 4 + x];
            }
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 4; x++) Ws[l * WS + cg * 4 + x] = w2[x];
        } else {
            d4 t2 = {0, 0, 0, 0};
            if (wid < 4) t2 = build_t ? dev_w2_tile(s_Tm, WS, 1, Ws, wid, lane) : dev_w2_tile(T, 1, STM_NB, Ws, wid, lane);
            __syncthreads();
            if (wid < 4) {
#pragma unroll
                for (int r = 0; r < 4; r++) Ws[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = t2[r];
            }
        }
    }
    __syncthreads();

    // ---- phase 3: C -= V W2 ----
    upd_chunk_load(ck, Vg, Cg, ld, lrow, mp, nbp, nc, lcg);
    for (int r0 = 0; r0 < mp; r0 += RB) {
        const int i = r0 + lrow;
        upd_chunk_to_lds(ck, i, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
        __syncthreads();
        if (r0 + RB < mp) upd_chunk_load(ck, Vg, Cg, ld, i + RB, mp, nbp, nc, lcg);
        d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const double a = Vs[(4 * kk + l4) * VS + 16 * wid + l15];
            const double b0 = Ws[(4 * kk + l4) * WS + l15];
            const double b1 = Ws[(4 * kk + l4) * WS + 16 + l15];
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, u1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wid + l4 + 4 * r;
            Cs[l15 * VS + row] -= u0[r];
            Cs[(16 + l15) * VS + row] -= u1[r];
        }
        __syncthreads();
        if (i < mp) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int col = lcg * 8 + q;
                if (col < nc) Cg[i + col * ld] = Cs[col * VS + lrow];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// dlarft('F','C'): T (NB x NB upper triangular, column-major, zero padded) of nc <= NB reflectors stored in the
// columns of Vg (= &F(0, first column), absolute row indexing, leading dimension ld), rows [r0, r1).
// diag[j] = absolute row of the unit diagonal of reflector j (BIGROW: none), tau[j] its coefficient.
//   G = V'V by fp64 MFMA (each wave sweeps every NW-th group of 4 rows; the same register is the A and the B
//   operand of the diagonal tiles), cross-wave sum through LDS scratch, then
//   T(0:b-1,b) = -tau_b T(0:b-1,0:b-1) G(0:b-1,b)   (SURVEY.md A.4).   scratch: >= NW*3*256 doubles.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// dlarft (forward, columnwise) from the Gram matrix G = V'V of a panel, tau = 0 columns included as zero columns:
//   T(a,b) = -tau_b sum_{a <= l < b} T(a,l) G(l,b),  T(b,b) = tau_b.
// Blocked 16 + 16: the two diagonal blocks are independent recurrences (one row of T per lane, two waves side by
// side), the off-diagonal block is T12 = -T11 (G12 T22), two 16 x 16 products by 256 threads -- the chain of
// dependent LDS reads + FMAs is a quarter of the 32-column recurrence's.  G's lower-left block is used as scratch.
// G, T: LDS, row stride STM_NB + 1; needs >= 256 threads; T is written completely (zeros below the diagonal and for
// columns >= nc).  Ends with a barrier.
// ------------------------------------------------------------------------------------------------
// tid: index of the thread among the (at least) 256 that work on THIS G / T (threadIdx.x, or threadIdx.x & 255 when the two halves
// of a 512-thread workgroup each build their own); the barriers are the whole workgroup's either way.
// Round 5: the recurrence is blocked all the way down -- T of 2h columns from the T of its two halves,
//   T = [T11, -T11 (G12 T22); 0, T22]      (dlarft's own merge rule; G12 = V1'V2),
// for h = 1, 2, 4, 8, 16: ten short steps of independent dot products (at most 16 terms, 256 outputs) instead of two 16-step chains of
// dependent LDS round trips + the 16/16 merge (3.7 us -> 1 us: T sits on the chain of every step, between the panel and block 0).
// A reflector with tau = 0 (dead / identity column) gives a zero row and column, as the column-by-column form does.
// X = G12 T22 is kept in G's lower triangle (the block below the diagonal block pair: never read as G).
template <int h>
__device__ __forceinline__ void dev_T_merge(double (*G)[STM_NB + 1], double (*T)[STM_NB + 1], int tid)
{
    const int pr = tid / (h * h), ij = tid % (h * h), i = ij / h, j = ij % h;
    const int A0 = 2 * h * pr, B0 = A0 + h;
    const bool mine = tid < (STM_NB / 2) * h;
    // (T11, T22 hold explicit zeros below their diagonals, so the sums run over all k with no predicates: the extra products
    //  are exact zeros)
    if (mine) {                                                      // X(i, j) = sum_k G12(i, k) T22(k, j)
        double p0 = 0, p1 = 0;
#pragma unroll
        for (int k = 0; k < h; k += 2) {
            p0 += G[A0 + i][B0 + k] * T[B0 + k][B0 + j];
            if (k + 1 < h) p1 += G[A0 + i][B0 + k + 1] * T[B0 + k + 1][B0 + j];
        }
        G[B0 + i][A0 + j] = p0 + p1;
    }
    lds_barrier();                                                   // (LDS only: stores to global memory stay in flight)
    if (mine) {                                                      // T12(i, j) = -sum_k T11(i, k) X(k, j)
        double p0 = 0, p1 = 0;
#pragma unroll
        for (int k = 0; k < h; k += 2) {
            p0 += T[A0 + i][A0 + k] * G[B0 + k][A0 + j];
            if (k + 1 < h) p1 += T[A0 + i][A0 + k + 1] * G[B0 + k + 1][A0 + j];
        }
        T[A0 + i][B0 + j] = -(p0 + p1);
    }
    lds_barrier();
}
__device__ __forceinline__ void dev_T_from_gram(double (*G)[STM_NB + 1], double (*T)[STM_NB + 1], const double *tau, int nc, int tid)
{
    for (int e = tid; e < STM_NB * STM_NB; e += 256) {
        const int a = e >> 5, b = e & 31;
        T[a][b] = (a == b && a < nc) ? tau[a] : 0.0;
    }
    lds_barrier();
    dev_T_merge<1>(G, T, tid);
    dev_T_merge<2>(G, T, tid);
    dev_T_merge<4>(G, T, tid);
    dev_T_merge<8>(G, T, tid);
    dev_T_merge<16>(G, T, tid);
}

template <int NTH>
__device__ void dev_gram_T(const double *Vg, long long ld, int r0, int r1, int nc, const int *diag, const double *tau,
                           double (*s_G)[STM_NB + 1], double (*s_T)[STM_NB + 1], double *Tout, double *scratch, unsigned long long *tl = nullptr)
{
    constexpr int NWV = NTH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const bool two = nc > 16;
    const int d0 = (l15 < nc) ? diag[l15] : STM_BIGROW;
    const int d1 = (16 + l15 < nc) ? diag[16 + l15] : STM_BIGROW;
    // (columns and rows are clamped so that every load is unconditional -- a predicated load is a branch around the
    //  access and the loads of a trip would be issued one round trip at a time -- and masked afterwards)
    const int ncc = max(nc, 1);
    const double *V0 = Vg + (long long)min(l15, ncc - 1) * ld, *V1 = Vg + (long long)min(16 + l15, ncc - 1) * ld;
    d4 g00 = {0, 0, 0, 0}, g01 = {0, 0, 0, 0}, g11 = {0, 0, 0, 0};
    const int nk = (r1 - r0 + 3) / 4;
    // four row groups per trip: the 8 loads of the NEXT trip are issued before the MFMAs of the current one (the loop is
    // bound by the memory latency of a trip otherwise)
    double x0[4], x1[4], y0[4], y1[4];
    auto load_trip = [&](int kk, double (&p0)[4], double (&p1)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = r0 + 4 * (kk + u * NWV) + l4;
            const int ic = max(min(i, r1 - 1), 0);
            p0[u] = V0[ic]; p1[u] = V1[ic];
        }
    };
    if (wid < nk) load_trip(wid, x0, x1);
    for (int kk = wid; kk < nk; kk += 4 * NWV) {
        const int kn = kk + 4 * NWV;
        if (kn < nk) load_trip(kn, y0, y1);
        double a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = r0 + 4 * (kk + u * NWV) + l4;
            const bool in = (kk + u * NWV < nk) && (i < r1);
            a0[u] = (in && i >= d0) ? ((i == d0) ? 1.0 : x0[u]) : 0.0;
            a1[u] = (in && two && i >= d1) ? ((i == d1) ? 1.0 : x1[u]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], a0[u], g00, 0, 0, 0);
            if (two) {
                g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], a1[u], g01, 0, 0, 0);
                g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], a1[u], g11, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { x0[u] = y0[u]; x1[u] = y1[u]; }
    }
    __syncthreads();                                    // scratch is free
    if (tl && tid == 0) tl[1] = wall_clock64();
    // cross-wave sum in groups of 4 waves (scratch: 4 * 768 doubles)
    for (int grp = 0; grp < NWV / 4; grp++) {
        if ((wid >> 2) == grp) {
            double *sc = scratch + (wid & 3) * 768;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (l4 + 4 * r) * 16 + l15;
                if (grp == 0) { sc[o] = g00[r]; sc[256 + o] = g01[r]; sc[512 + o] = g11[r]; }
                else { sc[o] += g00[r]; sc[256 + o] += g01[r]; sc[512 + o] += g11[r]; }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < 768; e += NTH) {
        const double v = scratch[e] + scratch[768 + e] + scratch[1536 + e] + scratch[2304 + e];
        const int tile = e >> 8, a = (e >> 4) & 15, b = e & 15;
        s_G[a + (tile == 2 ? 16 : 0)][b + (tile >= 1 ? 16 : 0)] = v;
    }
    __syncthreads();
    if (tl && tid == 0) tl[2] = wall_clock64();
    dev_T_from_gram(s_G, s_T, tau, nc, threadIdx.x);
    if (tl && tid == 0) tl[3] = wall_clock64();
    for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
        const int a = e % STM_NB, b = e / STM_NB;
        Tout[e] = (a < nc && b < nc && a <= b) ? s_T[a][b] : 0.0;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Apply the reflectors of an LDS-resident sub-panel to the remaining columns of the panel (qr_private_apply1 /
// dlarf semantics, :1359-1381, reflector after reflector).  One wave owns one column: the column's active rows
// [gs, r1) live in that wave's registers (<= 64 per lane, i.e. 4096 rows), V is read from LDS with the
// unit-diagonal / zero mask, each v'c is a wave64 shuffle reduction.  Columns taller than 4096 active rows use
// the slower two-pass form on the column in global memory.
//   Vl[(i - gs) + j*pst] = F(i, k1 + j0 + j) for the sub-panel columns j < sw.
// ------------------------------------------------------------------------------------------------
template <int NTH>
__device__ __forceinline__ void dev_apply_subpanel(const double *Vl, long long pst, int gs, int r1, int sw, const int *diag,
                                   const double *tau, double *Fc /* = &F(0, first remaining column) */, long long ld,
                                   int ncols)
{
    constexpr int NWV = NTH / 64;
    constexpr int MAXQ = 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rows = r1 - gs;
    if (rows <= 0) return;
    const int nq = (rows + 63) >> 6;
    for (int jc = wid; jc < ncols; jc += NWV) {
        double *cg = Fc + jc * ld + gs;                 // cg[r] = F(gs + r, column)
        if (nq <= MAXQ && NTH >= 512) {
            double creg[MAXQ];
#pragma unroll
            for (int q = 0; q < MAXQ; q++) {
                const int r = lane + 64 * q;
                creg[q] = (q < nq && r < rows) ? cg[r] : 0.0;
            }
            for (int i = 0; i < sw; i++) {
                const double ti = tau[i];
                if (ti == 0.0) continue;
                const double *v = Vl + i * pst + lane;  // explicit unit-lower-trapezoidal image: no masks
                double z = 0;
#pragma unroll
                for (int q = 0; q < MAXQ; q++)
                    if (q < nq) z += v[64 * q] * creg[q];
                z = wave_sum(z) * ti;
#pragma unroll
                for (int q = 0; q < MAXQ; q++)
                    if (q < nq) creg[q] -= z * v[64 * q];
            }
#pragma unroll
            for (int q = 0; q < MAXQ; q++) {
                const int r = lane + 64 * q;
                if (q < nq && r < rows) cg[r] = creg[q];
            }
        } else {
            for (int i = 0; i < sw; i++) {
                const double ti = tau[i];
                if (ti == 0.0) continue;
                const double *v = Vl + i * pst;
                double z = 0;
                for (int r = lane; r < rows; r += 64) z += v[r] * cg[r];
                z = wave_sum(z) * ti;
                for (int r = lane; r < rows; r += 64) cg[r] -= z * v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// workgroup-wide sum of K values per thread; every thread gets all K sums.  s_part: (NTH/64)*K doubles.
// ------------------------------------------------------------------------------------------------
template <int NTH, int K>
__device__ __forceinline__ void block_reduce_vec(double (&x)[K], double *s_part)
{
    constexpr int NWV = NTH / 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K; i++) x[i] = wave_sum(x[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < K; i++) s_part[wid * K + i] = x[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K; i++) {
        double r = 0;
#pragma unroll
        for (int w = 0; w < NWV; w++) r += s_part[w * K + i];
        x[i] = r;
    }
}

// rows [g, tmax) a panel will touch (uniform over the workgroup; 0 when the front is already finished)
__device__ __forceinline__ int panel_rows(const FrontSym &s, const FrontNum *num, const int *St, int p)
{
    if (num->done) return 0;
    const int k1 = p * STM_NB, k2 = min(s.fn, k1 + STM_NB);
    const int g = num->g;
    return max(0, min(num->fm, max(St[k2 - 1], g + (k2 - k1))) - g);
}

// ------------------------------------------------------------------------------------------------
// Register-resident sub-panel (qr_front's column loop, reference :1434-1609, for a sub-panel of <= 8 columns whose
// active rows fit RPT rows per thread).  Thread tid owns rows gs + tid + NTH*r of ALL sub-panel columns, so a
// column step touches no memory except its one workgroup reduction:
//   pass 1 : part[x] = sum_{g<i<t} F(i,k) F(i,k+x)   (x = 0: |x|^2 of dlarfg; x > 0: the v'c of dlarf, unscaled)
//            one 8-value workgroup reduction (halving butterfly + one LDS exchange, ONE barrier, double-buffered)
//   scalar : beta / tau / 1/(alpha-beta), dead-column test -- every thread redundantly
//   pass 2 : v = x * scal ;  c_x -= tau (top_x + scal part_x) v   for the remaining columns of the sub-panel
// The current column always lives in register column 0: a finished column is retired to the LDS image
// (lds[(i-gs) + j*pst], the layout dev_panel's write-back / apply tail expects) and the register columns rotate
// down by one, so the loop body exists once (an unrolled body per column overflows the instruction cache) and every
// register index is static.  State (g, rank, ...) follows dev_panel's conventions.
// ------------------------------------------------------------------------------------------------
template <int NTH, int RPT, int SWT>
__device__ __forceinline__ void dev_subpanel_reg(PanelShared &ps, double *F, long long ld, int *St, double *Tau, char *Rdead,
                                                 int k1, int j0, int sw, int nbp, int gs, int tmax, int m, int n, int npiv,
                                                 int ntol, double tol, int &g, int &rank, double &flops, double &lensum, int &nlive,
                                                 int &tlast, int &done, int &ncols_done, double *lds, long long pst, double sg, double isg)
{
    constexpr int NWV = NTH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int msp = (max(tmax - gs, 0) + 63) & ~63;
    double a[RPT][SWT];
#pragma unroll
    for (int x = 0; x < SWT; x++) {
        const double *src = F + (long long)(k1 + j0 + min(x, sw - 1)) * ld;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int i = gs + tid + NTH * r;
            const double val = src[max(min(i, tmax - 1), 0)];           // unconditional load, masked afterwards
            a[r][x] = (x < sw && i < tmax) ? val : 0.0;
        }
    }
    for (int j = 0; j < sw; j++) {
        const int jp = j0 + j, k = k1 + jp;
        if (!done && g >= m) {
            // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
            for (int kk = k + tid; kk < n; kk += NTH) {
                if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
                else St[kk] = m;
                Tau[kk] = 0;
            }
            for (int jj = jp + tid; jj < nbp; jj += NTH) { ps.diag[jj] = STM_BIGROW; ps.tau[jj] = 0; }
            done = 1;
            ncols_done = jp;
        }
        if (!done) {
            const int t = max(g + 1, ps.stair[jp]);
            const int par = j & 1;
            double part[8];
#pragma unroll
            for (int x = 0; x < 8; x++) part[x] = 0;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = gs + tid + NTH * r;
                const double xv = (i > g && i < t) ? a[r][0] * sg : 0.0;    // (one operand carries the magnitude guard)
#pragma unroll
                for (int x = 0; x < SWT; x++) part[x] += xv * a[r][x];      // (part[SWT..7] stay zero)
            }
            const bool owner = (tid == g - gs);         // holds row g in a[0][.]  (g - gs < 8 <= NTH)
            if (owner) {
#pragma unroll
                for (int x = 0; x < SWT; x++) ps.top[par][x] = a[0][x];
            }
            const double rw = wave_reduce8(part);
            if (lane < 8) ps.rsum[par][wid * 8 + lane] = rw;
            __syncthreads();
            const double rb = wave_sum_stride8((lane < NWV * 8) ? ps.rsum[par][lane] : 0.0);
            double sum[8];
            sum[0] = lane_bcast<red8_lane(0)>(rb); sum[1] = lane_bcast<red8_lane(1)>(rb);
            sum[2] = lane_bcast<red8_lane(2)>(rb); sum[3] = lane_bcast<red8_lane(3)>(rb);
            sum[4] = lane_bcast<red8_lane(4)>(rb); sum[5] = lane_bcast<red8_lane(5)>(rb);
            sum[6] = lane_bcast<red8_lane(6)>(rb); sum[7] = lane_bcast<red8_lane(7)>(rb);
            const double alpha = ps.top[par][0];
            const double ss = sum[0];
            double tau = 0, beta = alpha, scal = 0, scals = 0;
            if (ss != 0.0) stm_larfg_guarded(alpha, ss, sg, isg, beta, tau, scal, scals);   // (no active row below the diagonal: ss == 0 exactly)
            const bool dead = (k < ntol) && (fabs(beta) <= tol);
            if (dead) {
                // zero the column from the diagonal down, no reflector, g does not advance (:1495-1544)
#pragma unroll
                for (int r = 0; r < RPT; r++)
                    if (gs + tid + NTH * r >= g) a[r][0] = 0.0;
                if (tid == 0) { ps.st_out[jp] = 0; ps.dead[jp] = 1; ps.diag[jp] = STM_BIGROW; ps.tau[jp] = 0; }
                if (k == npiv - 1) rank = g;            // (:1604-1608) also taken on a dead last pivot
            } else {
                if (tid == 0) { ps.st_out[jp] = t; ps.dead[jp] = 0; ps.diag[jp] = g; ps.tau[jp] = tau; }
                flops += (double)(t - g) * (3.0 + 4.0 * (double)(n - k - 1));
                lensum += (double)(t - g);
                if (tau != 0.0) {
                    nlive++;
                    double w[8];
#pragma unroll
                    for (int x = 1; x < SWT; x++) w[x] = tau * (ps.top[par][x] + scals * sum[x]);
#pragma unroll
                    for (int r = 0; r < RPT; r++) {
                        const int i = gs + tid + NTH * r;
                        if (i > g && i < t) {
                            const double v = a[r][0] * scal;
                            a[r][0] = v;
#pragma unroll
                            for (int x = 1; x < SWT; x++) a[r][x] -= w[x] * v;
                        }
                    }
                    if (owner) {
#pragma unroll
                        for (int x = 1; x < SWT; x++) a[0][x] -= w[x];
                    }
                }
                if (owner) a[0][0] = beta;
                tlast = t;
                g++;
                if (k == npiv - 1) rank = g;
            }
        }
        // ---- retire register column 0 to the LDS image and rotate ----
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int il = tid + NTH * r;               // row - gs
            if (il < msp) lds[il + j * pst] = a[r][0];
#pragma unroll
            for (int x = 0; x + 1 < SWT; x++) a[r][x] = a[r][x + 1];
            a[r][SWT - 1] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// qr_front, one panel of <= STM_NB columns (reference: the column loop :1434-1609 with the panel policy
// fixed to k1 = p*NB, which changes rounding only).  Executed by one whole workgroup.
//
// The panel F(g:tmax, k1:k2) is processed in sub-panels of `sw` columns that fit the LDS budget (a 4000-row panel
// gets 4-column sub-panels, a 500-row panel is one sub-panel): each sub-panel is staged in LDS, reduced column
// by column there (norms: wave64 shuffles + one LDS step; dlarf: one wave per remaining sub-panel column),
// written back, and its block reflector is applied to the remaining columns of the panel with the fp64-MFMA
// routine above -- a blocked QR inside the panel, so the tall panel streams through the CU once per sub-panel
// instead of once per column.  Panels taller than the LDS budget for one column fall back to in-place work.
// Produces: R and V in F, Tau, Stair, Rdead, the T factor of the whole panel (Tout, NB x NB) and the pending
// block-reflector description in FrontNum (pg1, pt, pk1, pnb, pc0, pdiag).
// ------------------------------------------------------------------------------------------------
template <int NTH, bool INPLACE>
__device__ __forceinline__ void dev_panel(PanelShared &ps, const FrontSym &s, FrontNum *num, double *F, int *St, double *Tau, char *Rdead,
                          int p, double tol, int ntol_global, double *Tout, double *lds, int lds_doubles, int dbg = 0,
                          unsigned long long *dbgbuf = nullptr, double *Tkeep = nullptr, const double *sigp = nullptr)
{
    constexpr int NWV = NTH / 64;
    double *s_red = ps.red;
    int *s_diag = ps.diag;
    double *s_tau = ps.tau;
    double (*s_G)[STM_NB + 1] = ps.G;
    double (*s_T)[STM_NB + 1] = ps.T;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = num->fm, n = s.fn, npiv = s.fp;
    const long long ld = s.ld;
    const int k1 = p * STM_NB;
    const int k2 = min(n, k1 + STM_NB);
    const int nbp = k2 - k1;
    const int was_done = num->done;
    int g = num->g, rank = num->rank;
    __syncthreads();                                   // everyone has read FrontNum before anyone writes it
    if (was_done) {
        if (tid == 0) num->pd[STM_PDI(p)].pnb = 0;
        return;
    }
    const int ntol = min(ntol_global - s.col1, npiv);
    const int g1 = g;
    const double sg = sigp ? sigp[0] : 1.0, isg = sigp ? sigp[1] : 1.0;      // magnitude guard (stm_larfg_guarded)
    if (tid == 0) ps.nextss_col = -1;
    if (tid < nbp) ps.stair[tid] = St[k1 + tid];       // the panel's staircase, once (a global load per column
    __syncthreads();                                   //  step would sit on the critical path)
    const int tmax = min(m, max(ps.stair[nbp - 1], g1 + nbp));
    const int mp = tmax - g1;
    double flops = 0, lensum = 0;
    int done = 0, tlast = g1, ncols_done = nbp;

    // sub-panel width: as many columns as fit in LDS with the rows of the first sub-panel
    // INPLACE (rows too tall for even one LDS column, or nothing to do): work straight on F; otherwise sub-panels of
    // SW columns in LDS.  The two cases are separate instantiations so that every access of the LDS case is a
    // ds_* instruction (a pointer that may be either LDS or global compiles to slow FLAT accesses).
    int SW = INPLACE ? nbp : max(1, min(nbp, lds_doubles / (((mp + 63) & ~63) | 1)));
    constexpr bool in_place = INPLACE;
    // tall sub-panels: all waves split the ROWS of every column step and one workgroup reduction delivers the
    // column norm and all v'c dot products of the rest of the sub-panel (<= 7) at once
    const bool tall = !in_place && mp > 768;
    if (tall) SW = min(SW, 8);
    // register-resident sub-panels (<= 8 rows per thread, <= 8 columns): dev_subpanel_reg
    const int reg_min = 768;
    const bool regpath = !in_place && mp > reg_min && mp <= 8 * NTH && !(dbg & 256);
    if (regpath) SW = min(SW, mp > 4 * NTH ? 4 : 8);     // 8 rows per thread leave registers for 4 columns only

#ifdef STMMQR_STAMPS                                    /* phase timers (debug builds only: they cost 24 live VGPRs) */
    unsigned long long tph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tc0 = 0, tc1 = 0;
#define CSTAMP(idx) do { if (dbg & 32) { const unsigned long long t_ = clock64(); tph[idx] += t_ - tc1; tc1 = t_; } } while (0)
#define STAMP(idx) do { if (dbg & 16) { __syncthreads(); const unsigned long long t_ = clock64(); tph[idx] += t_ - tc0; tc0 = t_; } } while (0)
    if (dbg & 16) tc0 = clock64();
#else
#define CSTAMP(idx) do { } while (0)
#define STAMP(idx) do { } while (0)
#endif
    for (int j0 = 0; j0 < nbp && !done; j0 += SW) {
        const int sw = min(SW, nbp - j0);
        const int gs = g;                               // first active row of this sub-panel
        const int ms = tmax - gs;
        long long pst;
        double *Pb;                                     // Pb[(i-roff) + (j-coff)*pst] = F(i, k1+j)
        int roff = 0, coff = 0;                         // (never form a pointer outside the LDS object)
        bool use_reg = false;
        if constexpr (!INPLACE) {
            const int msp = (max(ms, 0) + 63) & ~63;    // zero-padded so that 64-row slots can be read unmasked
            pst = msp | 1;
            use_reg = regpath && ms <= 8 * NTH;
            if (!use_reg)
                for (int j = wid; j < sw; j += NWV) {
                    const double *src = F + (k1 + j0 + j) * ld;
                    // (unconditional loads on a clamped row, masked afterwards: see dev_tall_group)
                    for (int i = lane; i < msp; i += 64) {
                        const double val = src[gs + max(min(i, ms - 1), 0)];
                        lds[i + j * pst] = (i < ms) ? val : 0.0;
                    }
                }
            Pb = lds; roff = gs; coff = j0;
        } else {
            pst = ld;
            Pb = F + k1 * ld;
        }
        int nlive = 0;
        if (use_reg) {
            // ---- register-resident column loop; leaves the finished sub-panel in the LDS image ----
            if (ms <= NTH)
                dev_subpanel_reg<NTH, 1, 8>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
            else if (ms <= 2 * NTH)
                dev_subpanel_reg<NTH, 2, 8>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
            else if (ms <= 4 * NTH)
                dev_subpanel_reg<NTH, 4, 8>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
            else
                dev_subpanel_reg<NTH, 8, 4>(ps, F, ld, St, Tau, Rdead, k1, j0, sw, nbp, gs, tmax, m, n, npiv, ntol, tol, g, rank,
                                         flops, lensum, nlive, tlast, done, ncols_done, lds, pst, sg, isg);
        } else {
        __syncthreads();
        STAMP(0);
        for (int j = j0; j < j0 + sw; j++) {
            const int k = k1 + j;
            if (g >= m) {
                // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
                for (int kk = k + tid; kk < n; kk += NTH) {
                    if (kk < npiv) { Rdead[kk] = 1; St[kk] = 0; }
                    else St[kk] = m;
                    Tau[kk] = 0;
                }
                for (int jj = j + tid; jj < nbp; jj += NTH) { s_diag[jj] = STM_BIGROW; s_tau[jj] = 0; }
                done = 1;
                ncols_done = j;                         // columns >= j were finalised above, straight in global memory
                break;
            }
            const int t = max(g + 1, ps.stair[j]);
            double *col = Pb + (g - roff) + (j - coff) * pst;   // col[0] = F(g,k)
            const int len = t - g;                      // >= 1
            // ---- dlarfg ----
#ifdef STMMQR_STAMPS
            if (dbg & 32) tc1 = clock64();
#endif
            const double alpha = col[0];                // read before the barriers below: thread 0 overwrites it
            const int nrest = tall ? (j0 + sw - 1 - j) : 0;     // remaining sub-panel columns (tall scheme, <= 7)
            double part[8];
            double top[8];
#pragma unroll
            for (int x = 0; x < 8; x++) { part[x] = 0; top[x] = 0; }
            if (tall) {
#pragma unroll
                for (int x = 1; x < 8; x++)
                    if (x <= nrest) top[x] = col[x * pst];      // F(g, k+x)
                for (int i = 1 + tid; i < len; i += NTH) {
                    const double x0 = col[i], xv = x0 * sg;         // (one operand carries the magnitude guard)
                    part[0] += xv * x0;
#pragma unroll
                    for (int x = 1; x < 8; x++)
                        if (x <= nrest) part[x] += xv * col[i + x * pst];
                }
                CSTAMP(5);
                block_reduce_vec<NTH, 8>(part, ps.part);
                CSTAMP(6);
            } else if (ps.nextss_col == j && !(dbg & 128)) {
                part[0] = ps.nextss;                    // computed by the dlarf sweep of the previous column
                __syncthreads();                        // every wave has read alpha / nextss before anyone writes
            } else {
                double ss0 = 0;
                for (int i = 1 + tid; i < len; i += NTH) { const double xv = col[i]; ss0 += (xv * sg) * xv; }
                part[0] = block_sum<NTH>(ss0, s_red);
            }
            const double ss = part[0];
            double tau = 0, beta = alpha, scal = 0, scals = 0;
            if (len > 1 && ss != 0.0) stm_larfg_guarded(alpha, ss, sg, isg, beta, tau, scal, scals);
            const bool dead = (k < ntol) && (fabs(beta) <= tol);
            if (dead) {
                // zero the column from the diagonal down, no reflector, g does not advance (:1495-1544)
                for (int i = tid; i < tmax - g; i += NTH) col[i] = 0;
                if (tid == 0) { ps.st_out[j] = 0; ps.dead[j] = 1; s_diag[j] = STM_BIGROW; s_tau[j] = 0; }
                if (k == npiv - 1) rank = g;            // (:1604-1608) also taken on a dead last pivot
                __syncthreads();
                continue;
            }
            // (col[0] = beta is stored only after a barrier that every reader of alpha = col[0] has passed)
            if (tid == 0) { ps.st_out[j] = t; ps.dead[j] = 0; s_diag[j] = g; s_tau[j] = tau; if (tall) col[0] = beta; }
            CSTAMP(7);
            flops += (double)len * (3.0 + 4.0 * (double)(n - k - 1));
            lensum += (double)len;
            if (tall) {
                // ---- scale x and apply H_k to the rest of the sub-panel in the same sweep over the rows ----
                if (tau != 0.0) {
                    nlive++;
                    double w[8];
#pragma unroll
                    for (int x = 1; x < 8; x++) w[x] = tau * (top[x] + scals * part[x]);
                    for (int i = 1 + tid; i < len; i += NTH) {
                        const double v = col[i] * scal;
                        col[i] = v;
#pragma unroll
                        for (int x = 1; x < 8; x++)
                            if (x <= nrest) col[i + x * pst] -= w[x] * v;
                    }
                    if (tid == 0) {
#pragma unroll
                        for (int x = 1; x < 8; x++)
                            if (x <= nrest) col[x * pst] -= w[x];
                    }
                }
            } else {
                if (tau != 0.0)
                    for (int i = 1 + tid; i < len; i += NTH) col[i] *= scal;
                __syncthreads();
                if (tid == 0) col[0] = beta;            // no one reads F(g,k) any more in this step
                // ---- dlarf on the rest of the SUB-panel: one wave per column, v'c by DPP reduction (A.3) ----
                if (tau != 0.0) {
                    nlive++;
                    if (!(dbg & 4))
                    for (int jj = j + 1 + wid; jj < j0 + sw; jj += NWV) {
                        double *cc = Pb + (g - roff) + (jj - coff) * pst;
                        double w = (lane == 0) ? cc[0] : 0.0;
                        {
                            // four independent partial sums: the LDS read latency overlaps instead of chaining
                            double w1 = 0, w2 = 0, w3 = 0;
                            int i = 1 + lane;
                            for (; i + 192 < len; i += 256) {
                                w += col[i] * cc[i];
                                w1 += col[i + 64] * cc[i + 64];
                                w2 += col[i + 128] * cc[i + 128];
                                w3 += col[i + 192] * cc[i + 192];
                            }
                            for (; i < len; i += 64) w += col[i] * cc[i];
                            w += w1 + w2 + w3;
                        }
                        w = wave_sum(w) * tau;
                        if (lane == 0) cc[0] -= w;
                        if (jj == j + 1) {
                            // the next column: update it and accumulate its |x|^2 (rows below ITS diagonal g+1,
                            // up to its own staircase) in the same sweep
                            const int lenn = max(g + 2, ps.stair[jj]) - g;      // rows g .. g+lenn-1
                            double sq = 0;
                            for (int i = 1 + lane; i < max(len, lenn); i += 64) {
                                double cv = cc[i];
                                if (i < len) { cv -= w * col[i]; cc[i] = cv; }
                                if (i >= 2 && i < lenn) sq += (cv * sg) * cv;
                            }
                            sq = wave_sum(sq);
                            if (lane == 0) { ps.nextss = sq; ps.nextss_col = jj; }
                        } else {
                            for (int i = 1 + lane; i < len; i += 64) cc[i] -= w * col[i];
                        }
                    }
                }
            }
            tlast = t;
            g++;
            if (k == npiv - 1) rank = g;
            __syncthreads();
            CSTAMP(8);
        }
        }
        __syncthreads();
        STAMP(1);
        if (!in_place) {
            for (int j = wid; j < sw; j += NWV) {
                double *dst = F + (k1 + j0 + j) * ld;
                for (int i = gs + lane; i < tmax; i += 64) dst[i] = lds[(i - gs) + j * pst];
            }
        }
        __syncthreads();
        STAMP(2);
        // ---- apply this sub-panel's reflectors to the remaining columns of the panel; V is still in LDS ----
        if (!in_place && !done && j0 + sw < nbp && nlive > 0 && tlast > gs && !(dbg & 1)) {
            // turn the LDS image into the explicit unit-lower-trapezoidal V (R entries above the diagonals were
            // already written back): the apply loop then needs no masks
            for (int e = tid; e < sw * sw; e += NTH) {
                const int i = e % sw, j = e / sw;       // rows gs..gs+sw-1 are the only ones at or above a diagonal
                const int d = s_diag[j0 + j] - gs;
                if (i < ms) lds[i + j * pst] = (s_tau[j0 + j] == 0.0) ? 0.0 : ((i < d) ? 0.0 : ((i == d) ? 1.0 : lds[i + j * pst]));
            }
            if (nlive < sw)                              // dead / identity columns: whole column is not a reflector
                for (int j = 0; j < sw; j++)
                    if (s_tau[j0 + j] == 0.0)
                        for (int i = tid; i < ms; i += NTH) lds[i + j * pst] = 0.0;
            __syncthreads();
            dev_apply_subpanel<NTH>(lds, pst, gs, tlast, sw, s_diag + j0, s_tau + j0, F + (long long)(k1 + j0 + sw) * ld,
                                    ld, nbp - (j0 + sw));
            __syncthreads();
        }
        STAMP(5);
    }
    __syncthreads();
    if (tid < ncols_done) {                            // HStair / HTau / Rdead of this panel, one coalesced flush
        St[k1 + tid] = ps.st_out[tid];
        Tau[k1 + tid] = s_tau[tid];
        if (ps.dead[tid]) Rdead[k1 + tid] = 1;
    }
    STAMP(3);
    // ---- T of the whole panel for the trailing update ----
    if (!(dbg & 2)) dev_gram_T<NTH>(F + (long long)k1 * ld, ld, g1, tlast, nbp, s_diag, s_tau, s_G, s_T, Tout, lds);
    if (Tkeep)                                           // the same T, kept for the Q-apply on the resident factors
        for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
            const int a = e % STM_NB, b = e / STM_NB;
            Tkeep[e] = (a < nbp && b < nbp && a <= b) ? s_T[a][b] : 0.0;
        }
    STAMP(4);
#ifdef STMMQR_STAMPS
    if ((dbg & 16) && tid == 0 && dbgbuf)
        for (int e = 0; e < 12; e++) atomicAdd(&dbgbuf[e], tph[e]);
#endif
#undef STAMP
#undef CSTAMP
    PanelDesc *pd = &num->pd[STM_PDI(p)];
    if (tid < STM_NB) pd->pdiag[tid] = (tid < nbp) ? s_diag[tid] : STM_BIGROW;
    if (tid == 0) {
        num->g = g; num->rank = rank; num->done = done;
        pd->pg1 = g1; pd->pt = tlast; pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2;
        num->flops += flops;
        num->flops_upd += 4.0 * (double)(n - k2) * lensum;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Tall-panel pipeline (panels with more rows than one workgroup can stream through LDS cheaply).
//
// The panel's columns are cut into sub-panels ("groups") of 8 columns (4 above STM_TALL_WIDE rows) and ONE launch runs
// one workgroup per group, blockIdx.y = b:
//     * the group's rows [g1, tmax) x 8 columns are loaded ONCE into registers (thread tid owns rows g1 + tid + NTH*r),
//     * for s = 0 .. b-1: wait until group s has been factorized (FrontNum::prog, release/acquire at agent scope), then
//       apply its reflectors, read back from F with the unit-diagonal mask -- dlarf semantics, reflector after
//       reflector, each v'C one 8-value workgroup reduction (reference qr_private_apply1, :1359-1381),
//     * factorize the group in the same registers (the column step of dev_subpanel_reg), store every finished column
//       straight to F, publish prog.
// A group only ever waits for groups with a smaller blockIdx.y of the same front, i.e. for workgroups that were
// dispatched before it, so the wait cannot deadlock whatever the residency; the spin is bounded all the same.
// Compared with one workgroup per panel this keeps every column in registers for its whole life inside the panel
// (one read, one write of F per column), spreads the in-panel dlarf over up to eight CUs, overlaps the loads of the
// later groups with the factorization of the earlier ones, and makes the cost of a column step independent of the
// panel height (<= STM_TALL_MAX rows).  The last group also builds T of the whole panel (dev_gram_T) and the
// block-reflector description for the trailing update.
// ------------------------------------------------------------------------------------------------
// (lds_barrier, ld_agent / st_agent: stmmqr_devutil.h)

// whole workgroup: wait until *flag >= target (written by another workgroup of this launch), then acquire.
// Returns false if the bounded spin ran out (never expected; the caller gives up on the panel).
// `seen` (thread 0): a value of the flag loaded earlier (the poll of a consumer that is behind its producer costs a
// memory round trip although the flag has long been set: it loads the flag before its previous block of work instead).
__device__ __forceinline__ bool wait_progress(const int *flag, int target, int seen = -1)
{
    __shared__ int s_ok;
    __syncthreads();                               // (s_ok of a previous wait has been read by everyone)
    if (threadIdx.x == 0) {
        int ok = (seen >= target);
        for (int it = 0; !ok && it < (1 << 26); it++) {
            if (ld_agent(flag) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        s_ok = ok;
        // ONE lane acquires for the whole CU (the invalidate acts on the CU's L1 and on stale other-XCD lines; every
        // wave fencing costs 2-4x as much); the wait holds the barrier until the invalidate has completed.
        // (Dropping the acquire in favour of sc1 loads of the handed-over columns was measured: no gain.)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return s_ok != 0;
}
// publish: all global stores of this workgroup happen-before the flag value.
// Everything handed from one workgroup of the panel launch to another is stored WRITE-THROUGH (st_agent: sc1 stores, the
// panel columns included), so no L2 write-back (agent-scope release: buffer_wbl2 writes back every dirty line of the XCD's
// L2, microseconds when other fronts' updates have just run there) is needed: every wave waits for its own stores to be
// acknowledged (a workgroup barrier alone does not wait for them), the barrier joins the waves, one lane stores the flag.
__device__ __forceinline__ void publish_progress(int *flag, int value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_agent(flag, value);
}

template <int NTH>
__device__ __forceinline__ void block_reduce8(PanelShared &ps, int &par, const double (&part)[8], double (&sum)[8])
{
    constexpr int NWV = NTH / 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const double rw = wave_reduce8(part);
    if (lane < 8) ps.rsum[par][wid * 8 + lane] = rw;
    lds_barrier();
    const double rb = wave_sum_stride8((lane < NWV * 8) ? ps.rsum[par][lane] : 0.0);
    sum[0] = lane_bcast<red8_lane(0)>(rb); sum[1] = lane_bcast<red8_lane(1)>(rb);
    sum[2] = lane_bcast<red8_lane(2)>(rb); sum[3] = lane_bcast<red8_lane(3)>(rb);
    sum[4] = lane_bcast<red8_lane(4)>(rb); sum[5] = lane_bcast<red8_lane(5)>(rb);
    sum[6] = lane_bcast<red8_lane(6)>(rb); sum[7] = lane_bcast<red8_lane(7)>(rb);
    par ^= 1;
}

template <int NTH, int RPT, int SWT>
__device__ __forceinline__ void dev_tall_group(PanelShared &ps, const FrontSym &s, FrontNum *num, PanelDesc *pd, double *F,
                                               int *St, double *Tau, char *Rdead, int p, int b, int g1, int tmax,
                                               double tol, int ntol_global, double *Tout, double *lds, int dbg = 0,
                                               unsigned long long *dbgbuf = nullptr, double *Tkeep = nullptr, int defer_ok = 0,
                                               const double *sigp = nullptr, int *abortp = nullptr)
{
    const int tid = threadIdx.x;
    const int m = num->fm, n = s.fn, npiv = s.fp;              // (fm is fixed before the panel kernels run)
    const long long ld = s.ld;
    const int k1 = p * STM_NB, k2 = min(n, k1 + STM_NB), nbp = k2 - k1;
    const int c0 = SWT * b, sw = min(SWT, nbp - c0);           // my columns: k1 + c0 + x, x < sw
    const int ns = (nbp + SWT - 1) / SWT;
    const int rb = g1;                                         // first row of the register image
    int par = 0;
    if (tid < SWT) ps.stair[tid] = (tid < sw) ? St[k1 + c0 + tid] : 0;     // (only this group ever writes these)
#ifdef STMMQR_STAMPS
    unsigned long long ts0 = clock64(), ts1;
    const bool stamp_me = (dbg & 16) && dbgbuf && (b == ns - 1);
#define TSTAMP(idx) do { if (stamp_me) { __syncthreads(); ts1 = clock64(); if (tid == 0) atomicAdd(&dbgbuf[idx], ts1 - ts0); ts0 = ts1; } } while (0)
    // timeline of panel 1 (dbg & 32): wall clock (100 MHz) of thread 0 at the events of every group, dbgbuf[16 + 64 b + idx]
    const bool tl_on = (dbg & 32) && dbgbuf && p == (((dbg >> 24) & 127) ? ((dbg >> 24) & 127) : 1) && tid == 0 && s.parent < 0;      // (the root front only; panel = dbg bits 24-30)
#define TL(idx) do { if (tl_on && ((idx) < 8 || (idx) >= 20) && !((idx) >= 26 && (idx) < 30)) dbgbuf[16 + 64 * b + (idx)] = wall_clock64(); } while (0)
#define TLW(idx) do { if ((dbg & 32) && s.parent < 0 && p == (((dbg >> 24) & 127) ? ((dbg >> 24) & 127) : 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TL(idx); } while (0)
#define TCY(idx) do { if (tl_on && j == 4) dbgbuf[16 + 64 * b + 48 + (idx)] = clock64(); } while (0)
#else
#define TCY(idx) do { } while (0)
#define TSTAMP(idx) do { } while (0)
#define TL(idx) do { } while (0)
#define TLW(idx) do { } while (0)
#endif
    TL(0);

    // (loads are unconditional on a clamped index and masked afterwards: a predicated load becomes a branch around
    //  each access and the 8 x RPT loads would be issued one round trip at a time)
    double a[RPT][SWT];
#pragma unroll
    for (int x = 0; x < SWT; x++) {
        const double *src = F + (long long)(k1 + c0 + min(x, sw - 1)) * ld;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int i = rb + tid + NTH * r;
            const double val = src[min(i, tmax - 1)];
            a[r][x] = (x < sw && i < tmax) ? val : 0.0;
        }
    }
    TSTAMP(6);
    TLW(1);
#ifdef STMMQR_STAMPS
    int tl_h = 0;
#endif
    // ---- apply the reflectors of the groups before mine, as they become available ----
    int prev_done = 0;
    constexpr int HW = SWT / 2;                                // reflectors per published half group
    constexpr int NPV = HW * SWT, NGP = (NPV + 7) / 8;         // V'C products, in exchange groups of eight
    // (a group publishes twice: after the first half of its columns and at the end, so that the next group applies the
    //  first half of the reflectors while the second half is still being factorized)
    int seen = -1;
    bool have_chain = false;
    int ch_g = 0, ch_rank = 0, ch_pt = 0, ch_nl = 0;
    double ch_ls = 0, ch_fl = 0;
    for (int sp = 0; sp < b && !prev_done; sp++)
    for (int half = 0; half < 2; half++) {
        if (!wait_progress(&num->prog, STM_PROG * p + 2 * sp + 1 + half, seen)) {
            if (tid == 0) { st_agent(&num->perr, 1); if (abortp) st_agent(abortp + 1, 1); }      // (= STM_SET_PERR)
            return;
        }
        TSTAMP(7);
#ifdef STMMQR_STAMPS
        tl_h = (sp == b - 1) ? half : 8;                       // (only the last two half applications: the ones on the chain)
#endif
        TL(2 + 3 * tl_h);
        const int pc0 = SWT * sp + half * (SWT / 2);
        // (group sp ran out of rows?  Not num->done: a group that starts late would see the flag of a LATER group and
        //  skip the reflectors of the groups in between)
        if (half == 1) prev_done = (ld_agent(&pd->done_group) == sp);
        if (half == 1 && sp == b - 1) {
            // the scalars that travel along the chain of groups were stored before this flag: load them now, the
            // round trip hides behind the block application below (after the loop it would delay my first column)
            ch_g = ld_agent(&num->g); ch_rank = ld_agent(&num->rank); ch_pt = ld_agent(&pd->pt);
            ch_nl = ld_agent(&pd->nlive); ch_ls = ld_agent(&pd->lensum); ch_fl = ld_agent(&num->flops);
            have_chain = true;
        }
        // The HW reflectors of this half are applied as ONE block reflector, C -= V T' (V'C): the HW x SWT products V'C
        // and the strict upper triangle of V'V go through a single workgroup exchange (one barrier instead of one per
        // reflector), T is the HW x HW dlarft recurrence done redundantly by every thread.  tau == 0 / dead columns enter
        // as v = 0, tau = 0.  (The blocking is fixed -- half groups -- so the rounding does not depend on the timing.)
        double v[HW][RPT], tq[HW];
#pragma unroll
        for (int q = 0; q < HW; q++) {
            const int d = ld_agent(&pd->pdiag[pc0 + q]);
            const double tau = ld_agent(&Tau[k1 + pc0 + q]);
            const bool live = !(tau == 0.0 || d >= STM_BIGROW);
            tq[q] = live ? tau : 0.0;
            const double *vc = F + (long long)(k1 + pc0 + q) * ld;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                const double val = vc[min(i, tmax - 1)];       // (rows beyond a column's staircase are zero in F)
                v[q][r] = (!live || i < d || i >= tmax) ? 0.0 : ((i == d) ? 1.0 : val);
            }
        }
#ifdef STMMQR_STAMPS
        if (stamp_me) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP(12);
        TLW(3 + 3 * tl_h);
#endif
        if (tid == 0) seen = ld_agent(&num->prog);             // for the next wait: in flight during the block below
        // exchange group 0: the strict upper triangle of V'V; groups 1..: V'C, eight products per group
        const int lane = tid & 63, wid = tid >> 6;
        if constexpr (HW > 1) {
            double gv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) gv[e] = 0.0;
#pragma unroll
            for (int q1 = 0; q1 < HW; q1++) {
#pragma unroll
                for (int q2 = q1 + 1; q2 < HW; q2++) {
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < RPT; r++) acc += v[q1][r] * v[q2][r];
                    gv[q1 * HW - (q1 * (q1 + 1)) / 2 + (q2 - q1 - 1)] = acc;
                }
            }
            const double rw = wave_reduce8(gv);
            if (lane < 8) ps.rsumB[0][wid * 8 + lane] = rw;
        }
#pragma unroll
        for (int gi = 0; gi < NGP; gi++) {
            double pv[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int q = min((gi * 8 + e) / SWT, HW - 1), x = (gi * 8 + e) % SWT;
                double acc = 0.0;
                if (gi * 8 + e < NPV) {
#pragma unroll
                    for (int r = 0; r < RPT; r++) acc += v[q][r] * a[r][x];
                }
                pv[e] = acc;
            }
            const double rw = wave_reduce8(pv);
            if (lane < 8) ps.rsumB[1 + gi][wid * 8 + lane] = rw;
        }
        lds_barrier();
        // T (upper triangular, dlarft forward/columnwise): T(0:j,j) = -tau_j T(0:j,0:j) (V(:,0:j)' v_j)
        double Tq[HW][HW];
        if constexpr (HW == 1) Tq[0][0] = tq[0];
        else {
            const double rs = wave_sum_stride8((lane < (NTH / 64) * 8) ? ps.rsumB[0][lane] : 0.0);
            const double G8[8] = {lane_bcast<red8_lane(0)>(rs), lane_bcast<red8_lane(1)>(rs), lane_bcast<red8_lane(2)>(rs),
                                  lane_bcast<red8_lane(3)>(rs), lane_bcast<red8_lane(4)>(rs), lane_bcast<red8_lane(5)>(rs),
                                  lane_bcast<red8_lane(6)>(rs), lane_bcast<red8_lane(7)>(rs)};
#pragma unroll
            for (int j = 0; j < HW; j++) {
#pragma unroll
                for (int i = 0; i < HW; i++) Tq[i][j] = 0.0;
                Tq[j][j] = tq[j];
#pragma unroll
                for (int i = 0; i < j; i++) {
                    double acc = 0.0;
#pragma unroll
                    for (int l = i; l < j; l++) acc += Tq[i][l] * G8[l * HW - (l * (l + 1)) / 2 + (j - l - 1)];
                    Tq[i][j] = -tq[j] * acc;
                }
            }
        }
        // W = T' (V'C), accumulated group by group (eight products live at a time), then C -= V W
        double Wq[HW][SWT];
#pragma unroll
        for (int q = 0; q < HW; q++) {
#pragma unroll
            for (int x = 0; x < SWT; x++) Wq[q][x] = 0.0;
        }
#pragma unroll
        for (int gi = 0; gi < NGP; gi++) {
            const double rs = wave_sum_stride8((lane < (NTH / 64) * 8) ? ps.rsumB[1 + gi][lane] : 0.0);
            const double P8[8] = {lane_bcast<red8_lane(0)>(rs), lane_bcast<red8_lane(1)>(rs), lane_bcast<red8_lane(2)>(rs),
                                  lane_bcast<red8_lane(3)>(rs), lane_bcast<red8_lane(4)>(rs), lane_bcast<red8_lane(5)>(rs),
                                  lane_bcast<red8_lane(6)>(rs), lane_bcast<red8_lane(7)>(rs)};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (gi * 8 + e < NPV) {
                    const int l = min((gi * 8 + e) / SWT, HW - 1), x = (gi * 8 + e) % SWT;
#pragma unroll
                    for (int q = l; q < HW; q++) Wq[q][x] += Tq[l][q] * P8[e];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPT; r++) {
#pragma unroll
            for (int q = 0; q < HW; q++) {
#pragma unroll
                for (int x = 0; x < SWT; x++) a[r][x] -= v[q][r] * Wq[q][x];
            }
        }
        TSTAMP(11);
        TL(4 + 3 * tl_h);
    }
    TSTAMP(8);
    // a group before mine ran out of rows (g reached fm): its reflectors were still due on my columns (applied above);
    // nothing is left to factorize.  The group right after it finalises the panel, the others only store.
    if (prev_done) {
#pragma unroll
        for (int x = 0; x < SWT; x++) {
            if (x < sw) {
                double *dst = F + (long long)(k1 + c0 + x) * ld;
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + tid + NTH * r;
                    if (i < tmax) dst[i] = a[r][x];
                }
            }
        }
        if (ld_agent(&pd->done_group) != b - 1) return;
    }
    // ---- factorize my sub-panel (column step as in dev_subpanel_reg; finished columns go straight to F) ----
    // g / rank / pt / nlive / flops travel along the chain of groups: group b reads them after the acquire on group b-1
    if (!have_chain) {
        ch_g = ld_agent(&num->g); ch_rank = ld_agent(&num->rank); ch_pt = ld_agent(&pd->pt);
        ch_nl = ld_agent(&pd->nlive); ch_ls = ld_agent(&pd->lensum); ch_fl = (b == 0) ? num->flops : ld_agent(&num->flops);
    }
    int g = (b == 0) ? g1 : ch_g;
    int rank = ch_rank, done = prev_done, nlive = 0;
    int tlast = (b == 0) ? g1 : ch_pt;
    const int nl_before = (b == 0) ? 0 : ch_nl;
    const double ls_before = (b == 0) ? 0.0 : ch_ls;
    long long iflops = 0, ilen = 0;                            // the reference's flop count: integers, exact in fp64
    const int gs = g;
    const int ntol = min(ntol_global - s.col1, npiv);
    const double sg = sigp ? sigp[0] : 1.0, isg = sigp ? sigp[1] : 1.0;      // magnitude guard (stm_larfg_guarded)
    // HStair / HTau / pdiag / Rdead of my columns [j0, j1): from LDS to global memory, write-through, before a publish
    auto flush_cols = [&](int j0, int j1) {
        lds_barrier();
        if (tid >= j0 && tid < j1) {
            const int kk = k1 + c0 + tid;
            st_agent(&St[kk], ps.st_out[tid]); st_agent(&Tau[kk], ps.tau[tid]); st_agent(&pd->pdiag[c0 + tid], ps.diag[tid]);
            if (ps.dead[tid]) st_agent(&Rdead[kk], (char)1);
        }
    };
    int flushed = 0, jdone = SWT;
    lds_barrier();                                             // ps.stair
    // (fully unrolled over the group's columns: column j works on the register columns j .. SWT-1 only -- no dot products or
    //  updates of columns that have already been retired, no rotation of the register image; same operations on the live
    //  columns in the same order, so the same bits as the rotating loop)
#pragma unroll
    for (int j = 0; j < SWT; j++) {
        if (j >= sw || prev_done) continue;
        const int jp = c0 + j, k = k1 + jp;
        if (j == SWT / 2 && b + 1 < ns) {
            flush_cols(0, min(j, jdone)); flushed = j;
            publish_progress(&num->prog, STM_PROG * p + 2 * b + 1);   // first half is in F
        }
        if (!done && g >= m) {
            // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
            for (int kk = k + tid; kk < n; kk += NTH) {
                if (kk < npiv) { st_agent(&Rdead[kk], (char)1); st_agent(&St[kk], 0); }
                else st_agent(&St[kk], m);
                st_agent(&Tau[kk], 0.0);
            }
            for (int jj = jp + tid; jj < STM_NB; jj += NTH) st_agent(&pd->pdiag[jj], STM_BIGROW);
            done = 1;
            jdone = j;
        }
        if (!done) {
            // Straight-line column step: the dead-column and tau == 0 cases are folded into the scalars (a branch
            // around the update would make the compiler copy the whole register image at the join).
            TCY(0);
            const int t = max(g + 1, ps.stair[j]);
            double part[8], sum[8];
#pragma unroll
            for (int x = 0; x < 8; x++) part[x] = 0;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                const double xv = (i > g && i < t) ? a[r][j] * sg : 0.0;   // (one operand carries the magnitude guard)
#pragma unroll
                for (int x = j; x < SWT; x++) part[x - j] += xv * a[r][x];
            }
            const bool owner = (tid == g - rb);                // holds row g in a[0][.]  (g - rb < STM_NB <= NTH)
            const int tpar = par;
            if (owner) {
#pragma unroll
                for (int x = j; x < SWT; x++) ps.top[tpar][x - j] = a[0][x];
            }
            TCY(1);
            block_reduce8<NTH>(ps, par, part, sum);
            TCY(2);
            const double alpha = ps.top[tpar][0];
            const double ss = sum[0];
            // dlarfg; ss == 0 (no active row below the diagonal, or all of them zero) gives H = I
            double bb, tau0, scal0, scals0;
            stm_larfg_guarded(alpha, ss, sg, isg, bb, tau0, scal0, scals0);
            const bool ident = (ss == 0.0);
            const double beta = ident ? alpha : bb;
            const bool dead = (k < ntol) && (fabs(beta) <= tol);    // (:1495-1544) column zeroed, g does not advance
            const bool upd = !ident && !dead;
            const double tau = upd ? tau0 : 0.0;
            const double scal = upd ? scal0 : 0.0, scals = upd ? scals0 : 0.0;
            TCY(3);
            double w[8];
#pragma unroll
            for (int x = 1; x < SWT - j; x++) w[x] = tau * (ps.top[tpar][x] + scals * sum[x]);    // 0 unless upd
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                const bool act = (i > g && i < t);
                const double v = act ? a[r][j] * scal : 0.0;    // (upd false: the entries are zero already, or dead)
                a[r][j] = act ? v : ((dead && i >= g) ? 0.0 : a[r][j]);
#pragma unroll
                for (int x = j + 1; x < SWT; x++) a[r][x] -= w[x - j] * v;
            }
            if (owner) {
#pragma unroll
                for (int x = j + 1; x < SWT; x++) a[0][x] -= w[x - j];
                a[0][j] = dead ? 0.0 : beta;
            }
            if (tid == 0) {          // (flushed to global memory before each publish: flush_cols below)
                ps.st_out[j] = dead ? 0 : t; ps.tau[j] = tau; ps.diag[j] = dead ? STM_BIGROW : g; ps.dead[j] = dead ? 1 : 0;
            }
            if (!dead) {
                iflops += (long long)(t - g) * (3 + 4 * (long long)(n - k - 1));
                ilen += (t - g);
                nlive += (tau != 0.0);
                tlast = t;
                g++;
            }
            if (k == npiv - 1) rank = g;                       // (:1604-1608) also taken on a dead last pivot
            TCY(4);
        }
        // ---- retire register column j to F ----
        {
            double *dst = F + (long long)k * ld;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + tid + NTH * r;
                if (i < tmax) st_agent(&dst[i], a[r][j]);       // write-through: read by the other groups of this launch
            }
        }
        TCY(5);
        TL(20 + j);
    }
    TSTAMP(9);
    const double flops = (double)iflops, lensum = (double)ilen;
    if (!prev_done) flush_cols(min(flushed, jdone), min(sw, jdone));
    // ---- sub-panel bookkeeping; the last sub-panel (or the group after one that ran out of rows) finalises ----
    const bool last = prev_done || b == ns - 1;
    const int nl_total = nl_before + nlive;
    if (tid == 0 && !prev_done) {
        st_agent(&num->g, g); st_agent(&num->rank, rank); st_agent(&num->done, done);
        st_agent(&num->flops, ch_fl + flops);
        st_agent(&pd->sg[b], gs); st_agent(&pd->st[b], tlast); st_agent(&pd->pt, tlast); st_agent(&pd->nlive, nl_total);
        st_agent(&pd->lensum, ls_before + lensum);
        if (done) st_agent(&pd->done_group, b);
    }
    if (!last) {
        publish_progress(&num->prog, STM_PROG * p + 2 * b + 2);
        TL(30);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the write-through column stores of every wave: acknowledged)
    __syncthreads();                                           // all stores of this workgroup are complete and visible
    // T of a panel with trailing columns is left to the trailing update: the row-parallel form builds it from a Gram
    // block per slab (k_upd_w; the last slab to arrive runs the recurrence), the one-workgroup form from the Gram matrix
    // it accumulates beside W (dev_update_block) -- same association, bit-identical T -- so the Gram pass over the
    // whole panel (one CU filling rows x 32 columns: 6-11 us) and the recurrence leave the critical path.  The rule
    // depends on this front alone (defer_ok: an update launch follows this panel launch; k2 < n: this front takes part
    // in it), so the results do not depend on which fronts share a level.
    const int defer_t = (defer_ok && k2 < n && tlast > g1) ? 1 : 0;   // (no live reflector: T = 0 is written here)
    if (tid == 0) {
        // (pd->mode stays 1: it belongs to the header, and a column group that starts late -- the finalising group is
        //  not always the last one -- must still find it there)
        pd->t_deferred = defer_t;
        pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2;
        num->flops_upd += 4.0 * (double)(n - k2) * (prev_done ? ls_before : ls_before + lensum);
    }
    if (defer_t) {
        publish_progress(&num->prog, STM_PROG * p + 2 * b + 2);
        return;
    }
    if (tid < STM_NB) {
        ps.diag[tid] = (tid < nbp) ? ld_agent(&pd->pdiag[tid]) : STM_BIGROW;    // (columns past a `done` point were reset there)
        ps.tau[tid] = (tid < nbp) ? ld_agent(&Tau[k1 + tid]) : 0.0;
    }
    __syncthreads();
    dev_gram_T<NTH>(F + (long long)k1 * ld, ld, g1, tlast, nbp, ps.diag, ps.tau, ps.G, ps.T, Tout, lds
#ifdef STMMQR_STAMPS
                    , ((dbg & 32) && dbgbuf && p == 1) ? dbgbuf + 16 + 64 * b + 40 : nullptr
#endif
                    );
    if (Tkeep)
        for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
            const int ai = e % STM_NB, bi = e / STM_NB;
            Tkeep[e] = (ai < nbp && bi < nbp && ai <= bi) ? ps.T[ai][bi] : 0.0;
        }
    TSTAMP(10);
    TL(31);
    // (when an earlier group ran out of rows the groups after mine are still storing their columns: the kernel
    //  boundary orders those stores before the trailing update)
    publish_progress(&num->prog, STM_PROG * p + 2 * b + 2);
#undef TSTAMP
#undef TL
#undef TLW
#undef TCY
}

// ------------------------------------------------------------------------------------------------
// qr_cpack: C = F(rank:, fp:) upper trapezoid -> packed column-major; coalesced on the
// packed side.  part/nparts as in dev_assemble.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_cpack(const DevCtx &c, const FrontSym &s, FrontNum *num, int part, int nparts)
{
    const int tid = threadIdx.x;
    const int rank = num->rank, fm = num->fm;
    const int cn = s.fn - s.fp;
    int cm = min(fm - rank, cn);
    if (cm < 0) cm = 0;
    if (part == 0 && tid == 0) num->cm = cm;
    if (cm <= 0 || cn <= 0) return;
    const long long ld = s.ld;
    const double *Fc = c.Farena + s.foff + rank + (long long)s.fp * ld;
    double *C = c.Carena + s.coff;
    const long long tri = (long long)cm * (cm + 1) / 2;
    const long long csize = tri + (long long)cm * (cn - cm);
    for (long long e = (long long)part * NT + tid; e < csize; e += (long long)nparts * NT) {
        int cj, ci;
        if (e < tri) {
            cj = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
            while ((long long)cj * (cj + 1) / 2 > e) cj--;
            while ((long long)(cj + 1) * (cj + 2) / 2 <= e) cj++;
            ci = (int)(e - (long long)cj * (cj + 1) / 2);
        } else {
            const long long r = e - tri;
            cj = cm + (int)(r / cm);
            ci = (int)(r % cm);
        }
        C[e] = Fc[ci + cj * ld];
    }
}

// ------------------------------------------------------------------------------------------------
// small fronts: one workgroup runs the whole front (all panels, all trailing updates, C pack)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_front_wg(DevCtx c, const int *__restrict__ flist, int lds_doubles)
{
    extern __shared__ double dyn_lds[];
    __shared__ double s_Tw[STM_NB * STM_NB];
    __shared__ PanelShared ps;
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    FrontNum *num = &c.fnum[f];
    double *F = c.Farena + s.foff;
    auto Tkeep = [&](int p) { return c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : (double *)nullptr; };
    for (int p = 0; p < s.npanels; p++) {
        if ((c.dbg & 64) || panel_rows(s, num, c.Stair + s.rp, p) > lds_doubles - 65)
            dev_panel<NT, true>(ps, s, num, F, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, s_Tw,
                                dyn_lds, lds_doubles, c.dbg, nullptr, Tkeep(p), c.sig);
        else
            dev_panel<NT, false>(ps, s, num, F, c.Stair + s.rp, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, s_Tw,
                                 dyn_lds, lds_doubles, c.dbg, nullptr, Tkeep(p), c.sig);
        const int k2 = min(s.fn, (p + 1) * STM_NB);
        const int ncb = (s.fn - k2 + BN - 1) / BN;
        const PanelDesc *pd = &num->pd[STM_PDI(p)];
        for (int cb = 0; cb < ncb; cb++)
            dev_update_block(F, s.ld, pd->pg1, pd->pt - pd->pg1, pd->pk1, pd->pnb, pd->pdiag, s_Tw, k2 + cb * BN,
                             min(BN, s.fn - (k2 + cb * BN)), dyn_lds);
        __syncthreads();
    }
    dev_cpack(c, s, num, 0, 1);
}

// ------------------------------------------------------------------------------------------------
// Wave-pipelined panel (short panels: at most STM_WP_ROWS rows).  ONE workgroup, one WAVE per group of WP_SW columns,
// every wave holds all the rows of its columns in registers (lane l: rows rb + l + 64 r).  The column step of qr_front
// (SparseQR_factorize.c:1434-1609: dlarfg, dlarf on the rest of the group) then needs no workgroup barrier and no LDS
// round trip: the 8 sums of a step are one DPP reduction inside the wave.  Finished columns go to an image of the panel
// in LDS; the waves after the owner apply them (dlarf, one wave reduction each) as they appear -- the hand-off is an LDS
// flag per column (release / acquire at workgroup scope), polled by the consumer.  A wave only ever waits for waves
// before it, all of them resident in the same workgroup: no bounded waits, no global-memory flags.
// Which panels come here is decided from the front's own rows when the panel starts (k_panel, mode 2), so the results
// do not depend on the fronts that share the step or the device.
// ------------------------------------------------------------------------------------------------
#define WP_SW 4                          // columns per wave: 8 waves = the 512 threads of k_panel
struct WaveShared {
    int ready[STM_NB];                   // column j of the panel is in the LDS image, its scalars below are valid
    int d[STM_NB], t[STM_NB];            // unit-diagonal row of reflector j (STM_BIGROW: none) / one past its last row
    double tau[STM_NB];
    // the scalars that travel along the chain of columns: state AFTER the last finished column
    int g, rank, done, tlast, nlive, jdone;
    long long iflops, ilen;
};

template <int RPT>
__device__ __forceinline__ void dev_wave_panel(PanelShared &ps, WaveShared &wsh, const FrontSym &s, FrontNum *num, PanelDesc *pd,
                                               double *F, int *St, double *Tau, char *Rdead, int p, int g1, int tmax, double tol,
                                               int ntol_global, double *Tout, double *lds, double *Tkeep, int defer_ok,
                                               const double *sigp)
{
    constexpr int SW = WP_SW, NTH = 64 * (STM_NB / WP_SW), RS = 64 * RPT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = num->fm, n = s.fn, npiv = s.fp;
    const long long ld = s.ld;
    const int k1 = p * STM_NB, k2 = min(n, k1 + STM_NB), nbp = k2 - k1;
    const int c0 = SW * w, sw = max(0, min(SW, nbp - c0));
    const int rb = g1;
    const double flops_before = num->flops;
    if (tid < STM_NB) {
        wsh.ready[tid] = 0; wsh.d[tid] = STM_BIGROW; wsh.t[tid] = 0; wsh.tau[tid] = 0.0;
        ps.stair[tid] = (tid < nbp) ? St[k1 + tid] : 0;
        ps.st_out[tid] = 0; ps.dead[tid] = 0;
    }
    if (tid == 0) {
        wsh.g = g1; wsh.rank = num->rank; wsh.done = 0; wsh.tlast = g1; wsh.nlive = 0; wsh.jdone = STM_NB;
        wsh.iflops = 0; wsh.ilen = 0;
    }
    double a[RPT][SW];
#pragma unroll
    for (int x = 0; x < SW; x++) {
        const double *src = F + (long long)(k1 + c0 + min(x, max(sw, 1) - 1)) * ld;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int i = rb + lane + 64 * r;
            const double val = src[min(i, tmax - 1)];
            a[r][x] = (x < sw && i < tmax) ? val : 0.0;
        }
    }
    __syncthreads();
    auto wait_col = [&](int j) {
        while (__hip_atomic_load(&wsh.ready[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    };
    const int ntol = min(ntol_global - s.col1, npiv);
    const double sg = sigp ? sigp[0] : 1.0, isg = sigp ? sigp[1] : 1.0;
    if (sw > 0) {
        // ---- the reflectors of the waves before mine, one dlarf each, as they appear ----
        for (int j = 0; j < c0; j++) {
            wait_col(j);
            const double tau = wsh.tau[j];
            const int d = wsh.d[j];
            if (tau == 0.0 || d >= STM_BIGROW) continue;
            const double *vs = lds + j * RS;
            double v[RPT], pv[8];
#pragma unroll
            for (int x = 0; x < 8; x++) pv[x] = 0.0;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int i = rb + lane + 64 * r;
                const double val = vs[lane + 64 * r];           // (rows beyond a column's staircase are zero)
                v[r] = (i < d) ? 0.0 : ((i == d) ? 1.0 : val);
#pragma unroll
                for (int x = 0; x < SW; x++) pv[x] += v[r] * a[r][x];
            }
            const double rw = wave_reduce8(pv);
            double wv[SW];
            wv[0] = tau * lane_bcast<red8_lane(0)>(rw); wv[1] = tau * lane_bcast<red8_lane(1)>(rw);
            wv[2] = tau * lane_bcast<red8_lane(2)>(rw); wv[3] = tau * lane_bcast<red8_lane(3)>(rw);
#pragma unroll
            for (int r = 0; r < RPT; r++) {
#pragma unroll
                for (int x = 0; x < SW; x++) a[r][x] -= wv[x] * v[r];
            }
        }
        if (c0 > 0) wait_col(c0 - 1);                            // (a skipped reflector above was still waited for)
        int g = wsh.g, rank = wsh.rank, done = wsh.done, tlast = wsh.tlast, nlive = wsh.nlive, jdone = wsh.jdone;
        long long iflops = wsh.iflops, ilen = wsh.ilen;
        // ---- my columns ----
#pragma unroll
        for (int x = 0; x < SW; x++) {
            if (x >= sw) continue;
            const int jp = c0 + x, k = k1 + jp;
            if (!done && g >= m) {
                // no rows left: remaining pivotal columns are dead, remaining columns are empty (:1444-1458)
                for (int kk = k + lane; kk < n; kk += 64) {
                    if (kk < npiv) { Rdead[kk] = (char)1; St[kk] = 0; }
                    else St[kk] = m;
                    Tau[kk] = 0.0;
                }
                for (int jj = jp + lane; jj < STM_NB; jj += 64) pd->pdiag[jj] = STM_BIGROW;
                done = 1;
                jdone = jp;
            }
            if (!done) {
                const int t = max(g + 1, ps.stair[jp]);
                double part[8];
#pragma unroll
                for (int e = 0; e < 8; e++) part[e] = 0.0;
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + lane + 64 * r;
                    const double xv = (i > g && i < t) ? a[r][x] * sg : 0.0;   // (one operand carries the magnitude guard)
#pragma unroll
                    for (int y = x; y < SW; y++) part[y - x] += xv * a[r][y];
                }
                const double rw = wave_reduce8(part);
                const double sum0 = lane_bcast<red8_lane(0)>(rw), sum1 = lane_bcast<red8_lane(1)>(rw);
                const double sum2 = lane_bcast<red8_lane(2)>(rw), sum3 = lane_bcast<red8_lane(3)>(rw);
                const double sum[4] = {sum0, sum1, sum2, sum3};
                const int ol = g - rb;                            // lane that holds row g in a[0][.]  (g - rb < STM_NB)
                double top[SW];
#pragma unroll
                for (int y = x; y < SW; y++)
                    top[y - x] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a[0][y]), ol),
                                                  __builtin_amdgcn_readlane(__double2loint(a[0][y]), ol));
                const double alpha = top[0], ss = sum[0];
                double bb, tau0, scal0, scals0;
                stm_larfg_guarded(alpha, ss, sg, isg, bb, tau0, scal0, scals0);
                const bool ident = (ss == 0.0);
                const double beta = ident ? alpha : bb;
                const bool dead = (k < ntol) && (fabs(beta) <= tol);    // (:1495-1544) column zeroed, g does not advance
                const bool upd = !ident && !dead;
                const double tau = upd ? tau0 : 0.0;
                const double scal = upd ? scal0 : 0.0, scals = upd ? scals0 : 0.0;
                double wv[SW];
#pragma unroll
                for (int y = 1; y < SW - x; y++) wv[y] = tau * (top[y] + scals * sum[y]);     // 0 unless upd
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + lane + 64 * r;
                    const bool act = (i > g && i < t);
                    const double v = act ? a[r][x] * scal : 0.0;
                    a[r][x] = act ? v : ((dead && i >= g) ? 0.0 : a[r][x]);
#pragma unroll
                    for (int y = x + 1; y < SW; y++) a[r][y] -= wv[y - x] * v;
                }
                if (lane == ol) {
#pragma unroll
                    for (int y = x + 1; y < SW; y++) a[0][y] -= wv[y - x];
                    a[0][x] = dead ? 0.0 : beta;
                }
                if (lane == 0) {
                    ps.st_out[jp] = dead ? 0 : t; ps.tau[jp] = tau; ps.diag[jp] = dead ? STM_BIGROW : g; ps.dead[jp] = dead ? 1 : 0;
                    wsh.tau[jp] = tau; wsh.d[jp] = dead ? STM_BIGROW : g; wsh.t[jp] = t;
                }
                if (!dead) {
                    iflops += (long long)(t - g) * (3 + 4 * (long long)(n - k - 1));
                    ilen += (t - g);
                    nlive += (tau != 0.0);
                    tlast = t;
                    g++;
                }
                if (k == npiv - 1) rank = g;                       // (:1604-1608) also taken on a dead last pivot
            }
            // ---- the column goes to F and, for the waves after mine, to the LDS image ----
            {
                double *dst = F + (long long)k * ld;
                double *vs = lds + jp * RS;
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int i = rb + lane + 64 * r;
                    if (i < tmax) dst[i] = a[r][x];
                    vs[lane + 64 * r] = a[r][x];
                }
            }
            if (x == sw - 1 || done) {
                // hand the chain to the next wave (after my last column; at once when the rows ran out: the columns after this
                // one carry no reflector, their owners only store)
                if (lane == 0) {
                    wsh.g = g; wsh.rank = rank; wsh.done = done; wsh.tlast = tlast; wsh.nlive = nlive; wsh.jdone = jdone;
                    wsh.iflops = iflops; wsh.ilen = ilen;
                }
            }
            if (lane == 0) {
                __hip_atomic_store(&wsh.ready[jp], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    // ---- bookkeeping of the panel (what the last column group of the pipeline does) ----
    const int g = wsh.g, tlast = wsh.tlast, jdone = wsh.jdone, done = wsh.done;
    const double lensum = (double)wsh.ilen;
    if (tid < min(nbp, jdone)) {
        const int kk = k1 + tid;
        St[kk] = ps.st_out[tid]; Tau[kk] = ps.tau[tid]; pd->pdiag[tid] = ps.diag[tid];
        if (ps.dead[tid]) Rdead[kk] = (char)1;
    }
    const int defer_t = (defer_ok && k2 < n && tlast > g1) ? 1 : 0;   // (as dev_tall_group: T of a panel with trailing columns
                                                                      //  is built by the update that follows)
    if (tid == 0) {
        num->g = g; num->rank = wsh.rank; num->done = done;
        num->flops = flops_before + (double)wsh.iflops;
        pd->sg[0] = g1; pd->st[0] = tlast; pd->pt = tlast; pd->nlive = wsh.nlive; pd->lensum = lensum;
        pd->done_group = done ? 0 : -1;
        pd->t_deferred = defer_t;
        pd->pk1 = k1; pd->pnb = nbp; pd->pc0 = k2;
        num->flops_upd += 4.0 * (double)(n - k2) * lensum;
    }
    if (defer_t) return;
    __syncthreads();                                               // pdiag / Tau / F of this workgroup: visible to all its threads
    if (tid < STM_NB) {
        ps.diag[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;  // (columns past a `done` point were reset there)
        ps.tau[tid] = (tid < nbp) ? Tau[k1 + tid] : 0.0;
    }
    __syncthreads();
    dev_gram_T<NTH>(F + (long long)k1 * ld, ld, g1, tlast, nbp, ps.diag, ps.tau, ps.G, ps.T, Tout, lds);
    if (Tkeep)
        for (int e = tid; e < STM_NB * STM_NB; e += NTH) {
            const int ai = e % STM_NB, bi = e / STM_NB;
            Tkeep[e] = (ai < nbp && bi < nbp && ai <= bi) ? ps.T[ai][bi] : 0.0;
        }
}

#define NTP 512               // threads of the large-front panel kernel (8 waves, <= 256 VGPRs each)

// ------------------------------------------------------------------------------------------------
// large fronts: panel and trailing update are separate launches (many workgroups per update)
// ------------------------------------------------------------------------------------------------
// the body of k_panel: column group b of panel p of front f (fi: the front's place in the launch's lists)
__device__ __forceinline__ void dev_k_panel(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int fi, int b,
                                            int nsub, int defer_ok, int lds_doubles, PanelShared &ps, WaveShared &wsh, double *dyn_lds)
{
    const int f = flist[fi];
    const int p = plist[fi];                                   // every front of a step is at its own panel
    __builtin_amdgcn_s_setprio(3);                             // (critical path: ahead of the side stream's update waves)
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || stm_use_ca(s, p, c.panel_algo, c.ca_min_rows)) return;      // (the Gram-based panel kernel takes those)
    FrontNum *num = &c.fnum[f];
    double *F = c.Farena + s.foff;
    double *T = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    int *St = c.Stair + s.rp;
    double *Tkeep = c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr;
    PanelDesc *pd = &num->pd[STM_PDI(p)];
    if ((c.dbg & 2048) && b == ((c.dbg >> 20) & 7)) {          // tests: column group (dbg >> 20) & 7 starts ~1 ms late
        for (int it = 0; it < 4000; it++) __builtin_amdgcn_s_sleep(100);
    }
    if ((c.dbg & 4096) && b > 0) {                              // tests: every column group but the first gives up at once
        if (threadIdx.x == 0) STM_SET_PERR(c, num);
        return;
    }
    bool tall = stm_tall_panel(s, p, c.tall_min) && !(c.dbg & 256);
    if (!tall && b > 0) return;
    if (tall) {
        const int k1 = p * STM_NB, k2 = min(s.fn, k1 + STM_NB), nbp = k2 - k1;
        int mode, g1, tmax, w;
        if (b == 0) {
            // group 0 decides the mode and publishes the panel-wide constants (the header) to the other groups
            const int was_done = num->done;
            g1 = num->g;
            tmax = min(num->fm, max(St[k2 - 1], g1 + nbp));
            w = (tmax - g1 > STM_TALL_XWIDE) ? 2 : (tmax - g1 > STM_TALL_WIDE) ? 4 : STM_SW;
            // mode 0: nothing to do or the whole panel is done below by this workgroup (it does not fit the register
            // image, or needs more groups than were launched: more rows than the full-rank estimate)
            // (the groups the PLAN launches for this front -- not the launch's, which other fronts may have raised)
            mode = (was_done || tmax - g1 > STM_TALL_MAX || (nbp + w - 1) / w > min(nsub, stm_tall_launches(s, p, c.tall_min))) ? 0 : 1;
            // mode 2: a short panel -- this workgroup alone, a wave per 4 columns (dev_wave_panel).  The condition belongs to
            // the front: the rows its staircase reaches now (in a sparse front the first panels are far shorter than the front:
            // 89 % of the pipeline panels of the xenon1 stand-in have at most 512 rows).  Measured and dropped: panels of up
            // to 1024 rows with 16 rows per lane and a 16-column ring image in LDS (default workload 120.6 -> 154 ms).
            // (the LDS test never fails for a launch sized by the host's rule, STM_NB * STM_WP_ROWS doubles for every launch with
            //  a pipeline panel: it keeps a smaller launch safe)
            if (!was_done && tmax - g1 <= STM_WP_ROWS && !(c.dbg & 16384) &&
                STM_NB * ((tmax - g1 <= 128) ? 128 : (tmax - g1 <= 256) ? 256 : 512) <= lds_doubles)
                mode = 2;
            if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0 && !was_done) {       // diagnosis: panels by their actual rows
                const int rws = tmax - g1;
                atomicAdd(&c.dbgbuf[32 + (rws <= 128 ? 0 : rws <= 256 ? 1 : rws <= 512 ? 2 : rws <= 1024 ? 3 : rws <= 2048 ? 4 : rws <= 4096 ? 5 : 6)], 1ull);
                if (rws <= 512 && stm_panel_rows_est(s, p) > STM_WP_ROWS) atomicAdd(&c.dbgbuf[39], 1ull);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                st_agent(&pd->mode, mode); st_agent(&pd->pg1, g1); st_agent(&pd->pt, g1); st_agent(&pd->tmax, tmax);
                st_agent(&pd->nlive, 0); st_agent(&pd->sw, w); st_agent(&pd->done_group, -1);
                if (was_done) { pd->pnb = 0; pd->t_deferred = 0; }
            }
            publish_progress(&num->hdr, p + 1);
            if (was_done) return;
        } else {
            if (!wait_progress(&num->hdr, p + 1)) { if (threadIdx.x == 0) STM_SET_PERR(c, num); return; }
            mode = ld_agent(&pd->mode); g1 = ld_agent(&pd->pg1); tmax = ld_agent(&pd->tmax); w = ld_agent(&pd->sw);
            if (mode != 1 || b * w >= nbp) return;
        }
        if (mode == 2) {
            const int rows = tmax - g1;
#define WAVE_ARGS ps, wsh, s, num, pd, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, g1, tmax, c.tol, c.ntol, T, dyn_lds, Tkeep, defer_ok, c.sig
            if (rows <= 128) dev_wave_panel<2>(WAVE_ARGS);
            else if (rows <= 256) dev_wave_panel<4>(WAVE_ARGS);
            else dev_wave_panel<8>(WAVE_ARGS);
#undef WAVE_ARGS
            return;
        }
        if (mode == 1) {
            const int rows = tmax - g1;
#define TALL_ARGS ps, s, num, pd, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, b, g1, tmax, c.tol, c.ntol, T, dyn_lds, c.dbg, c.dbgbuf, Tkeep, defer_ok, c.sig, c.abort
            if (w == 2) dev_tall_group<NTP, 16, 2>(TALL_ARGS);
            else if (w == 4) dev_tall_group<NTP, 8, 4>(TALL_ARGS);
            else if (rows <= NTP) dev_tall_group<NTP, 1, 8>(TALL_ARGS);
            else if (rows <= 2 * NTP) dev_tall_group<NTP, 2, 8>(TALL_ARGS);
            else dev_tall_group<NTP, 4, 8>(TALL_ARGS);
#undef TALL_ARGS
            return;
        }
    }
    // One workgroup does the whole panel.  The LDS it stages sub-panels in is a property of the FRONT (stm_front_lds: what
    // the front would get alone), never of the launch -- the sub-panel width, and with it the rounding, must not depend on
    // the fronts that share the step.  The host sizes the launch for the fronts PLANNED to come here; a panel that was
    // planned for the pipeline and fell back (more rows than the full-rank estimate, recovery of a timed-out wait) works
    // in place when the launch is smaller than that.
    const int lds_front = stm_front_lds(s);
    if ((c.dbg & 64) || lds_front > lds_doubles || panel_rows(s, num, St, p) > lds_front - 65)
        dev_panel<NTP, true>(ps, s, num, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, T, dyn_lds,
                             lds_front, c.dbg, c.dbgbuf, Tkeep, c.sig);
    else
        dev_panel<NTP, false>(ps, s, num, F, St, c.Tau + s.rp, c.Rdead + s.col1, p, c.tol, c.ntol, T, dyn_lds,
                              lds_front, c.dbg, c.dbgbuf, Tkeep, c.sig);
    if (threadIdx.x == 0) { pd->mode = 0; pd->t_deferred = 0; }
}

__global__ __launch_bounds__(NTP) void k_panel(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int nsub,
                                               int defer_ok, int lds_doubles)
{
    extern __shared__ double dyn_lds[];
    __shared__ PanelShared ps;
    __shared__ WaveShared wsh;
    const unsigned long long t0c = clock64(), t0w = wall_clock64();
    dev_k_panel(c, flist, plist, blockIdx.x, blockIdx.y, nsub, defer_ok, lds_doubles, ps, wsh, dyn_lds);
    if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0) {           // diagnosis: shader clock held during the panels (cycles / 100 MHz ticks)
        atomicAdd(&c.dbgbuf[44], clock64() - t0c); atomicAdd(&c.dbgbuf[45], wall_clock64() - t0w); atomicAdd(&c.dbgbuf[46], 1ull);
    }
}

__global__ __launch_bounds__(NT) void k_update(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0)
{
    extern __shared__ double dyn_lds[];
    const int f = flist[blockIdx.y];
    const int p = plist[blockIdx.y];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    const PanelDesc *pd = &c.fnum[f].pd[STM_PDI(p)];
    const int cbg = cb0 + (int)blockIdx.x * (1 + c.cbskip);
    const int c0 = pd->pc0 + cbg * BN;
    if (c0 >= s.fn) return;
    double *Tw = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    if (pd->t_deferred) {
        // T was left to the update: every column-block workgroup builds it from its own Gram matrix; the first one
        // stores it (T slot of the plan, kept T of the Q-apply)
        const bool first = (cbg == 0) || (c.cbskip > 0 && blockIdx.x == 0);   // (under a stride every plan keeps its own T)
        dev_update_block(c.Farena + s.foff, s.ld, pd->pg1, pd->pt - pd->pg1, pd->pk1, pd->pnb, pd->pdiag, nullptr, c0,
                         min(BN, s.fn - c0), dyn_lds, c.Tau + s.rp + pd->pk1, first ? Tw : nullptr,
                         (first && c.Tall) ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr);
        return;
    }
    dev_update_block(c.Farena + s.foff, s.ld, pd->pg1, pd->pt - pd->pg1, pd->pk1, pd->pnb, pd->pdiag, Tw, c0,
                     min(BN, s.fn - c0), dyn_lds);
}

// qr_larftb seam, method QR_QX: the pending block reflector of FrontNum::pd[0] applied WITHOUT the transpose,
// C <- (I - V T V') C (dlarfb 'L','N','F','C', SparseQR_factorize.c:1880-1886); T from k_larft
__global__ __launch_bounds__(NT) void k_update_n(DevCtx c, int f)
{
    extern __shared__ double dyn_lds[];
    const FrontSym s = c.fs[f];
    const PanelDesc *pd = &c.fnum[f].pd[0];
    const int c0 = pd->pc0 + (int)blockIdx.x * BN;
    if (c0 >= s.fn) return;
    dev_update_block<true>(c.Farena + s.foff, s.ld, pd->pg1, pd->pt - pd->pg1, pd->pk1, pd->pnb, pd->pdiag,
                           c.Tws + STM_TSLOT(c.tslot[f], 0) * STM_NB * STM_NB, c0, min(BN, s.fn - c0), dyn_lds);
}

// standalone T factor of the pending block reflector described by FrontNum (qr_larftb seam)
__global__ __launch_bounds__(NT) void k_larft(DevCtx c, int f)
{
    extern __shared__ double dyn_lds[];
    __shared__ PanelShared ps;
    int *s_diag = ps.diag;
    double *s_tau = ps.tau;
    double (*s_G)[STM_NB + 1] = ps.G;
    double (*s_T)[STM_NB + 1] = ps.T;
    const FrontSym s = c.fs[f];
    const PanelDesc *num = &c.fnum[f].pd[0];
    const int tid = threadIdx.x;
    if (tid < STM_NB) {
        s_diag[tid] = num->pdiag[tid];
        s_tau[tid] = (tid < num->pnb && num->pdiag[tid] != STM_BIGROW) ? c.Tau[s.rp + num->pk1 + tid] : 0.0;
    }
    __syncthreads();
    dev_gram_T<NT>(c.Farena + s.foff + (long long)num->pk1 * s.ld, s.ld, num->pg1, num->pt, num->pnb, s_diag, s_tau, s_G,
               s_T, c.Tws + STM_TSLOT(c.tslot[f], 0) * STM_NB * STM_NB, dyn_lds);
}

// ------------------------------------------------------------------------------------------------
// Row-parallel trailing update for tall panels: the phases of dev_update_block as two launches so that
// the rows are split over workgroups too (a 4000-row update has 126 column blocks x 16 row slabs instead of
// 126 workgroups that each walk 63 chunks):
//   k_upd_w : partial W1 = V(slab)' C(slab, cb)            grid (cb, slab, front)   -> Wp[front][cb][slab]
//   k_upd_c : W2 = T' sum_slabs W1 ;  C(slab, cb) -= V(slab) W2     grid (cb, slab, front)
// SLAB rows per slab; the summation order over slabs is fixed (deterministic results).
// ------------------------------------------------------------------------------------------------
#define SLAB STM_UPD_SLAB

// the body of k_upd_w: workgroup (cb, sl) of front fi of the launch's lists; ncbx = column blocks of the launch (the last one is
// the Gram block when with_gram)
template <bool PRE = false>        // PRE: the slab's four chunks requested at once (riders: two workgroups per CU, little else hides a round trip)
__device__ __forceinline__ void dev_k_upd_w(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                            int with_gram, double *Wp, const long long *__restrict__ wpoff, int *wcnt, int cb, int sl,
                                            int fi, int ncbx, double *dyn_lds)
{
    __shared__ int s_pd[STM_NB];
    const int f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    const PanelDesc *pd = &c.fnum[f].pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    // the front's slice of the workspace: (ncbf + 1) column-block slots of nslf slabs (both symbolic: the host sized it so)
    const int ncbf = stm_upd_ncb(s, p), nslf = stm_upd_nsl(s);
    // with_gram: the LAST column block of the launch is the Gram block (T may have been left to this kernel):
    // C = V, partial V'V per slab; the last slab workgroup of a front to arrive sums them in slab order and
    // builds T (dlarft recurrence) for k_upd_c
    const bool gram = with_gram && (cb == ncbx - 1);
    if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0) {                  // diagnosis: launched / useful workgroups
        atomicAdd(&c.dbgbuf[40], 1ull);
        const int c0d = gram ? pd->pk1 : pd->pc0 + (cb0 + cb * (1 + c.cbskip)) * BN;
        if (!((gram && !pd->t_deferred) || (!gram && cb0 + cb * (1 + c.cbskip) >= ncbf) || nbp <= 0 || mp <= 0 || c0d >= s.fn || sl * SLAB >= mp))
            atomicAdd(&c.dbgbuf[41], 1ull);
    }
    if (gram && !pd->t_deferred) return;
    if (!gram && cb0 + cb * (1 + c.cbskip) >= ncbf) return;
    const int c0 = gram ? pd->pk1 : pd->pc0 + (cb0 + cb * (1 + c.cbskip)) * BN;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl * SLAB >= mp) return;
    const int nc = gram ? nbp : min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Vs = dyn_lds, *Cs = Vs + STM_NB * VS;
    if (tid < STM_NB) s_pd[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;
    __syncthreads();
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    const double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc = {0, 0, 0, 0};
    const int rend = min(mp, (sl + 1) * SLAB);
    if constexpr (PRE) {
        UpdChunk ckp[SLAB / RB];
#pragma unroll
        for (int q = 0; q < SLAB / RB; q++) upd_chunk_load(ckp[q], Vg, Cg, ld, sl * SLAB + q * RB + (tid & 63), mp, nbp, nc, tid >> 6);
        const int pfrom = (nc == BN) ? upd_plain_from(s_pd, g1, lane) : STM_BIGROW;
#pragma unroll
        for (int q = 0; q < SLAB / RB; q++) {
            const int r0 = sl * SLAB + q * RB;
            if (r0 < rend) {
                upd_chunk_to_lds(ckp[q], r0 + (tid & 63), mp, nbp, nc, s_pd, g1, tid & 63, tid >> 6, Vs, Cs, gram, r0 > pfrom && r0 + RB <= mp);
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < RB / 4; kk++) {
                    const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
                    const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                }
                __syncthreads();
            }
        }
    } else {
    UpdChunk ck;
    upd_chunk_load(ck, Vg, Cg, ld, sl * SLAB + (tid & 63), mp, nbp, nc, tid >> 6);
    const int pfrom = (nc == BN) ? upd_plain_from(s_pd, g1, lane) : STM_BIGROW;
    for (int r0 = sl * SLAB; r0 < rend; r0 += RB) {
        upd_chunk_to_lds(ck, r0 + (tid & 63), mp, nbp, nc, s_pd, g1, tid & 63, tid >> 6, Vs, Cs, gram, r0 > pfrom && r0 + RB <= mp);
        __syncthreads();
        if (r0 + RB < rend) upd_chunk_load(ck, Vg, Cg, ld, r0 + RB + (tid & 63), mp, nbp, nc, tid >> 6);
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk++) {
            const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
            const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    }
    // (the Gram block's slot comes after the front's last column block; a pair / quad update front only ever brings column blocks 0 .. sweep-1
    //  here -- everything beyond goes through k_upd_w2, whose slots are packed more tightly: stm_pair_slots -- so its Gram block
    //  sits right behind those two)
    const int gslot = (c.ypoff && c.ypoff[f] >= 0) ? min(ncbf, c.sweep) : ncbf;
    double *W = Wp + wpoff[fi] + ((long long)(gram ? gslot : cb) * nslf + sl) * (STM_NB * BN);
    __shared__ int s_ticket;
    const int nsl = (mp + SLAB - 1) / SLAB;
    if (!gram) {
        // The partial W1 of this slab; the LAST slab workgroup of the column block to arrive (ticket) adds the partials in
        // slab order into slot 0, so that k_upd_c reads 8 KB per workgroup instead of every partial again (at 27 000 rows
        // 106 partials = 848 KB against 128 KB of V and C per slab: most of that kernel's traffic).  Hand-off as for the
        // Gram block below: write-through stores, every wave's vmcnt(0), the barrier, one lane's ticket.
        if (nsl == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc[r];
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[r]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + cb;          // (one counter per column block: the front's slice has
                                                                   //  at least ncbf + 1 blocks)
        if (tid == 0) {
            s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_ticket == nsl - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (s_ticket != nsl - 1) return;
        double *W0 = Wp + wpoff[fi] + ((long long)cb * nslf) * (STM_NB * BN);
        double v[STM_NB * BN / NT];
#pragma unroll
        for (int q = 0; q < STM_NB * BN / NT; q++) v[q] = stm_ordered_sum<true>(W0 + tid + q * NT, STM_NB * BN, nsl);   // fixed order
#pragma unroll
        for (int q = 0; q < STM_NB * BN / NT; q++) W0[tid + q * NT] = v[q];
        return;
    }
    // ---- Gram block: the last slab to arrive builds T ----
    // The partial G is stored write-through (as the panel pipeline's hand-offs: no L2 write-back before the ticket),
    // every wave waits for its stores, the barrier joins them, one lane takes the ticket; the last arriver acquires
    // with one lane.
#pragma unroll
    for (int r = 0; r < 4; r++) st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FrontNum *num = &c.fnum[f];
    if (tid == 0) {
        s_ticket = __hip_atomic_fetch_add(&num->gcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket == nsl - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&num->gcnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_ticket != nsl - 1) return;
    double *s_G = Vs;                                          // [row * WS + col] (the chunk images are free now)
    __shared__ double s_tau[STM_NB];
    const double *G0 = Wp + wpoff[fi] + ((long long)gslot * nslf) * (STM_NB * BN);
    if (tid < STM_NB) s_tau[tid] = (tid < nbp) ? c.Tau[s.rp + pd->pk1 + tid] : 0.0;
    for (int e = tid; e < STM_NB * BN; e += NT) {
        const double gsum = stm_ordered_sum<false>(G0 + e, STM_NB * BN, nsl);      // fixed order: deterministic
        s_G[(e / BN) * WS + (e % BN)] = gsum;                  // G(row, col) = v_row' v_col
    }
    __syncthreads();
    double *Tout = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    double *Tkeep = c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr;
    double (*s_Tb)[STM_NB + 1] = reinterpret_cast<double (*)[STM_NB + 1]>(Cs);            // (the C chunk image is free as well)
    dev_T_from_gram(reinterpret_cast<double (*)[STM_NB + 1]>(s_G), s_Tb, s_tau, nbp, threadIdx.x);
    for (int e = tid; e < STM_NB * STM_NB; e += NT) {
        const int a = e % STM_NB, b = e / STM_NB;
        const double tv = (a <= b && a < nbp && b < nbp) ? s_Tb[a][b] : 0.0;
        Tout[e] = tv;
        if (Tkeep) Tkeep[e] = tv;
    }
}

__global__ __launch_bounds__(NT) void k_upd_w(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                              int with_gram, double *Wp, const long long *__restrict__ wpoff, int *wcnt)
{
    extern __shared__ double dyn_lds[];
    dev_k_upd_w(c, flist, plist, cb0, with_gram, Wp, wpoff, wcnt, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, dyn_lds);
}

__global__ __launch_bounds__(NT) void k_upd_c(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                              const double *Wp, const long long *__restrict__ wpoff)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_pd[STM_NB];
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    const PanelDesc *pd = &c.fnum[f].pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    const int cb = blockIdx.x, sl = blockIdx.y;
    const int nslf = stm_upd_nsl(s);
    const int c0 = pd->pc0 + (cb0 + cb * (1 + c.cbskip)) * BN;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl * SLAB >= mp) return;
    // Very tall panels: one workgroup takes 2 or 4 slabs -- every workgroup of a column block sums the same nsl partial
    // W (nsl x 8 KB: twice a slab of V and C at 32 slabs), so fewer, longer workgroups read less per updated row.
    // (The rows of C are independent here: the arithmetic does not change.)
    const int nsl_all = (mp + SLAB - 1) / SLAB;
    const int spw = (nsl_all >= 32) ? 4 : (nsl_all >= 16) ? 2 : 1;      // (1 and 8 measured: 1 is 6% slower at 27 000 rows, 8 the same)
    if (sl % spw) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Vs = dyn_lds, *Cs = Vs + STM_NB * VS, *Ws = Cs + BN * VS;
    // (the prologue's W1 / T images live in the chunk images, which are first written after it: 42 KB of LDS per
    //  workgroup instead of 59, three workgroups per CU instead of two)
    double *s_W1 = Vs, *s_T = Cs;
    if (tid < STM_NB) s_pd[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;
    // the first chunk of V and C is requested before the W2 prologue so that its latency hides behind it
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int lrow = tid & 63, lcg = tid >> 6;
    UpdChunk ck;
    upd_chunk_load(ck, Vg, Cg, ld, sl * SLAB + lrow, mp, nbp, nc, lcg);
    // W2 = T' W1 (every slab workgroup of a column block recomputes it: 32^3 multiply-adds against one launch less per
    // panel); W1 = the slabs' partials added in slab order by k_upd_w
    {
        const double *W0 = Wp + wpoff[fi] + ((long long)cb * nslf) * (STM_NB * BN);
        const double *T = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
        for (int e = tid; e < STM_NB * BN; e += NT) {
            const double v = W0[e];                                 // (the slabs' partials were added by k_upd_w)
            s_W1[(e / BN) * WS + (e % BN)] = v;
            s_T[(e / STM_NB) * WS + (e % STM_NB)] = T[e];          // s_T[col][row] = T(row, col)
        }
        __syncthreads();
        const d4 w2 = dev_w2_tile(s_T, 1, WS, s_W1, wid, lane);      // (round 5: on the matrix cores, dev_w2_tile)
#pragma unroll
        for (int r = 0; r < 4; r++) Ws[(16 * (wid >> 1) + l4 + 4 * r) * WS + 16 * (wid & 1) + l15] = w2[r];
    }
    __syncthreads();
    const int rend = min(mp, (sl + spw) * SLAB);
    const int pfrom = upd_plain_from(s_pd, g1, tid & 63);
    // C never goes through LDS here: a thread keeps the eight entries of its row that it loaded, V.W2 comes back from the
    // MFMA layout through the (otherwise unused) C image and the thread subtracts and stores from registers -- two
    // barriers per chunk instead of three (the next chunk's V image is written after the second one, its product
    // image after the next first one: every reader of either is past by then)
    for (int r0 = sl * SLAB; r0 < rend; r0 += RB) {
        const int i = r0 + lrow;
        double cc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cc[q] = ck.c[q];
        upd_chunk_v_to_lds(ck, i, mp, nbp, s_pd, g1, lrow, lcg, Vs, r0 > pfrom && r0 + RB <= mp);
        __syncthreads();
        if (r0 + RB < rend) upd_chunk_load(ck, Vg, Cg, ld, i + RB, mp, nbp, nc, lcg);
        d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const double a = Vs[(4 * kk + l4) * VS + 16 * wid + l15];
            const double b0 = Ws[(4 * kk + l4) * WS + l15];
            const double b1 = Ws[(4 * kk + l4) * WS + 16 + l15];
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, u1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wid + l4 + 4 * r;
            Cs[l15 * VS + row] = u0[r];
            Cs[(16 + l15) * VS + row] = u1[r];
        }
        __syncthreads();
        if (i < mp) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int col = lcg * 8 + q;
                if (col < nc) Cg[i + col * ld] = cc[q] - Cs[col * VS + lrow];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The row-parallel update as ONE launch (options.fused_update): a workgroup keeps its 256 x 32 tile of C (and of V) in
// registers between the two phases, so C is read once and written once per panel (k_upd_w + k_upd_c read it twice: the
// update of a 27 000-row front is HBM-bound on exactly that) and a launch boundary + a prologue disappear from every
// step.  Between the phases the slab workgroups of a column block meet through global memory:
//   every slab   : partial W1 -> its slot (write-through), ticket
//   last arriver : W1 = sum of the partials in slab order, waits for T (FrontNum::tready, set by the Gram block's last
//                  slab in this launch or an earlier one), W2 = T' W1 -> slot 0, flag[column block] = epoch
//   every slab   : waits for the flag, C -= V W2 from its registers, stores C.
// Grid (slab, column block [0 = Gram block], front): the slab workgroups of a column block are consecutive in dispatch
// order and a workgroup only waits for workgroups of its own column block and for the Gram block (y = 0) of its front, so
// the earliest unfinished column block is always completely dispatched as long as the GPU holds gridDim.x workgroups
// (the host falls back to the two-launch form beyond 256 slabs).  The waits are bounded; one that runs out sets perr.
// Arithmetic = k_upd_w + k_upd_c exactly (same chunks, same MFMA sequence, same summation order): the same bits.
// ------------------------------------------------------------------------------------------------
// the body of k_upd_f: workgroup (slab sl, column block by [0 = the Gram block when with_gram]) of front fi of the launch's lists
__device__ __forceinline__ void dev_k_upd_f(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                            int with_gram, double *Wp, const long long *__restrict__ wpoff, int *wcnt, int *wflag,
                                            int epoch, int sl, int by, int fi, double *dyn_lds)
{
    __shared__ int s_pd[STM_NB];
    __shared__ int s_ticket, s_ok;
    const int f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    FrontNum *num = &c.fnum[f];
    const PanelDesc *pd = &num->pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    const bool gram = with_gram && by == 0;
    const int cb = by - (with_gram ? 1 : 0);                       // launch-relative column block
    const int ncbf = stm_upd_ncb(s, p), nslf = stm_upd_nsl(s);
    if (gram && !pd->t_deferred) return;
    if (!gram && cb0 + cb * (1 + c.cbskip) >= ncbf) return;
    const int c0 = gram ? pd->pk1 : pd->pc0 + (cb0 + cb * (1 + c.cbskip)) * BN;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl * SLAB >= mp) return;
    const int nc = gram ? nbp : min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int lrow = tid & 63, lcg = tid >> 6;
    double *Vs = dyn_lds, *Cs = Vs + STM_NB * VS, *Ws = Cs + BN * VS;
    if (tid < STM_NB) s_pd[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int r00 = sl * SLAB, rend = min(mp, (sl + 1) * SLAB);
    // the whole tile: every load of the workgroup is in flight at once, and the values stay for the second phase
    UpdChunk ck[SLAB / RB];
#pragma unroll
    for (int q = 0; q < SLAB / RB; q++) upd_chunk_load(ck[q], Vg, Cg, ld, r00 + q * RB + lrow, mp, nbp, nc, lcg);
    __syncthreads();
    // ---- phase 1: partial W1 = V(slab)' C(slab) (k_upd_w) ----
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < SLAB / RB; q++) {
        const int r0 = r00 + q * RB;
        if (r0 < rend) {
            upd_chunk_to_lds(ck[q], r0 + lrow, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs, gram);
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < RB / 4; kk++) {
                const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
                const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }
    const int nsl = (mp + SLAB - 1) / SLAB;
    const int gslot = (c.ypoff && c.ypoff[f] >= 0) ? min(ncbf, c.sweep) : ncbf;                         // (as in k_upd_w)
    double *Wslot = Wp + wpoff[fi] + ((long long)(gram ? gslot : cb) * nslf) * (STM_NB * BN);     // slot 0 of the column block
    double *W = Wslot + (long long)sl * (STM_NB * BN);
#pragma unroll
    for (int r = 0; r < 4; r++) st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double *Tslot = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    if (gram) {
        // ---- Gram block: the last slab to arrive builds T and raises tready (k_upd_w) ----
        if (tid == 0) {
            s_ticket = __hip_atomic_fetch_add(&num->gcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_ticket == nsl - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&num->gcnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (s_ticket != nsl - 1) return;
        double *s_G = Vs;
        __shared__ double s_tau[STM_NB];
        if (tid < STM_NB) s_tau[tid] = (tid < nbp) ? c.Tau[s.rp + pd->pk1 + tid] : 0.0;
        for (int e = tid; e < STM_NB * BN; e += NT) {
            const double gsum = stm_ordered_sum<true>(Wslot + e, STM_NB * BN, nsl);     // fixed order: deterministic
            s_G[(e / BN) * WS + (e % BN)] = gsum;
        }
        __syncthreads();
        double *Tkeep = c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr;
        double (*s_Tb)[STM_NB + 1] = reinterpret_cast<double (*)[STM_NB + 1]>(Cs);
        dev_T_from_gram(reinterpret_cast<double (*)[STM_NB + 1]>(s_G), s_Tb, s_tau, nbp, threadIdx.x);
        for (int e = tid; e < STM_NB * STM_NB; e += NT) {
            const int a = e % STM_NB, b = e / STM_NB;
            const double tv = (a <= b && a < nbp && b < nbp) ? s_Tb[a][b] : 0.0;
            st_agent(&Tslot[e], tv);
            if (Tkeep) Tkeep[e] = tv;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st_agent(&num->tready, epoch);
        return;
    }
    // ---- the column block's meeting point ----
    int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + cb;
    int *flag = wflag + wpoff[fi] / (STM_NB * BN) + cb;
    bool last = (nsl == 1);
    if (nsl > 1) {
        if (tid == 0) {
            s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_ticket == nsl - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        last = (s_ticket == nsl - 1);
    }
    double *s_W1 = Vs, *s_T = Cs;                                  // (the chunk images are free between the phases)
    if (last) {
        // (>=: with look-ahead the T of the NEXT panel may be announced while the side stream still applies this one)
        if (pd->t_deferred && !stm_wait_ge(&num->tready, epoch, c.abort, &s_ok)) { if (tid == 0) STM_SET_PERR(c, num); return; }
        for (int e = tid; e < STM_NB * BN; e += NT) {
            const double v = stm_ordered_sum<true>(Wslot + e, STM_NB * BN, nsl);        // fixed order: deterministic
            s_W1[(e / BN) * WS + (e % BN)] = v;
            s_T[(e / STM_NB) * WS + (e % STM_NB)] = ld_agent(&Tslot[e]);                // s_T[col][row] = T(row, col)
        }
        __syncthreads();
        // W2 = T' W1 (k_upd_c's prologue, done once per column block here)
        const d4 w2 = dev_w2_tile(s_T, 1, WS, s_W1, wid, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) Ws[(16 * (wid >> 1) + l4 + 4 * r) * WS + 16 * (wid & 1) + l15] = w2[r];
        if (nsl > 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) st_agent(&Wslot[(16 * (wid >> 1) + l4 + 4 * r) * BN + 16 * (wid & 1) + l15], w2[r]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_agent(flag, epoch);
        } else
            __syncthreads();
    } else {
        if (!stm_wait_ge(flag, epoch, c.abort, &s_ok)) { if (tid == 0) STM_SET_PERR(c, num); return; }
        for (int e = tid; e < STM_NB * BN; e += NT) Ws[(e / BN) * WS + (e % BN)] = ld_agent(&Wslot[e]);
        __syncthreads();
    }
    // ---- phase 2: C(slab) -= V(slab) W2 from the registers (k_upd_c) ----
#pragma unroll
    for (int q = 0; q < SLAB / RB; q++) {
        const int r0 = r00 + q * RB;
        if (r0 < rend) {
            const int i = r0 + lrow;
            upd_chunk_to_lds(ck[q], i, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
            __syncthreads();
            d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < STM_NB / 4; kk++) {
                const double a = Vs[(4 * kk + l4) * VS + 16 * wid + l15];
                const double b0 = Ws[(4 * kk + l4) * WS + l15];
                const double b1 = Ws[(4 * kk + l4) * WS + 16 + l15];
                u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, u1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * wid + l4 + 4 * r;
                Cs[l15 * VS + row] -= u0[r];
                Cs[(16 + l15) * VS + row] -= u1[r];
            }
            __syncthreads();
            if (i < mp) {
#pragma unroll
                for (int qq = 0; qq < 8; qq++) {
                    const int col = lcg * 8 + qq;
                    if (col < nc) Cg[i + col * ld] = Cs[col * VS + lrow];
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(NT, 2) void k_upd_f(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                              int with_gram, double *Wp, const long long *__restrict__ wpoff, int *wcnt,
                                              int *wflag, int epoch)
{
    extern __shared__ double dyn_lds[];
    dev_k_upd_f(c, flist, plist, cb0, with_gram, Wp, wpoff, wcnt, wflag, epoch, blockIdx.x, blockIdx.y, blockIdx.z, dyn_lds);
}

// ------------------------------------------------------------------------------------------------
// Passenger launches (options.lookahead = 2, the default).  The chain of a large front is panel(t) -> T(t) + update of column
// block 0 -> panel(t+1); the rest of update t (column blocks 1..) does not feed panel t+1.  Instead of a second stream (two
// cross-stream hand-offs of ~9 us per step) the two launches of that rest RIDE on the chain's own launches, as extra workgroups
// behind the chain's in dispatch order:
//   B(t)   = k_upd_fw : T(t) + block 0 (k_upd_f's workgroups)          + k_upd_w of blocks 1.. of step t
//   A(t+1) = k_panel_pc: panel(t+1) (k_panel's workgroups, dispatched first) + k_upd_c of blocks 1.. of step t
// Every workgroup does exactly what it does in the serial order (same bits); a launch lasts as long as its longest role.
// Dependencies: k_upd_w(t) needs V(t) [A(t)] and C after k_upd_c(t-1) [A(t)]; k_upd_c(t) needs its partial sums [B(t)] and T(t)
// [B(t)]; block 0 of step t+1 is block 1 of step t: complete after A(t+1), before B(t+1).
// ------------------------------------------------------------------------------------------------
// k_upd_c's tile by a 512-thread workgroup: the two 256-thread halves take alternate 64-row chunks of the slab; every row of C sees
// exactly the operations of k_upd_c.  lds: STM_NB * WS (W2) + per half the V image and the product image (BN * VS each).
#define STM_PC_LDS_DOUBLES (STM_NB * WS + 4 * BN * VS)
__device__ __forceinline__ void dev_upd_c_h2(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                             const double *Wp, const long long *__restrict__ wpoff, int fi, int cb, int sl0,
                                             int rspw, double *dyn_lds, int *s_pd)
{
    const int f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    const PanelDesc *pd = &c.fnum[f].pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    const int nslf = stm_upd_nsl(s);
    const int c0 = pd->pc0 + (cb0 + cb * (1 + c.cbskip)) * BN;
    // (a rider takes rspw consecutive slabs -- the launch's choice: the rows of C are independent, the arithmetic does not change --
    //  so that one prologue, during which nothing else runs on this CU, serves more rows)
    const int sl = sl0 * rspw, spw = rspw;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl * SLAB >= mp) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Ws = dyn_lds, *Vs = Ws + STM_NB * WS + half * (2 * BN * VS), *Cs = Vs + BN * VS;
    double *s_W1 = dyn_lds + STM_NB * WS, *s_T = s_W1 + BN * VS;       // (half 0's images: first written after the prologue)
    if (threadIdx.x < STM_NB) s_pd[threadIdx.x] = ((int)threadIdx.x < nbp) ? pd->pdiag[threadIdx.x] : STM_BIGROW;
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int lrow = tid & 63, lcg = tid >> 6;
    const int rbeg = sl * SLAB, rend = min(mp, (sl + spw) * SLAB);
    int r0 = rbeg + half * RB;
    // (a rider has the CU to itself -- the launch carries the panel's registers and LDS -- so nothing else hides its round trips:
    //  the first two chunks of the half are requested before the prologue, from then on two trips ahead)
    UpdChunk ck0, ck1;
    upd_chunk_load(ck0, Vg, Cg, ld, r0 + lrow, mp, nbp, nc, lcg);
    upd_chunk_load(ck1, Vg, Cg, ld, r0 + 2 * RB + lrow, mp, nbp, nc, lcg);
    {
        const double *W0 = Wp + wpoff[fi] + ((long long)cb * nslf) * (STM_NB * BN);
        const double *T = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
        for (int e = threadIdx.x; e < STM_NB * BN; e += 2 * NT) {
            const double v = W0[e];
            s_W1[(e / BN) * WS + (e % BN)] = v;
            s_T[(e / STM_NB) * WS + (e % STM_NB)] = T[e];          // s_T[col][row] = T(row, col)
        }
        __syncthreads();
        d4 w2 = {0, 0, 0, 0};
        if (half == 0) w2 = dev_w2_tile(s_T, 1, WS, s_W1, wid, lane);
        __syncthreads();                                           // (Ws does not alias the prologue images; the barrier orders
                                                                   //  the reads of s_W1 / s_T before half 0's first V image)
        if (half == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) Ws[(16 * (wid >> 1) + l4 + 4 * r) * WS + 16 * (wid & 1) + l15] = w2[r];
        }
    }
    const int pfrom = upd_plain_from(s_pd, g1, lane);
    const int nch = (rend - rbeg + RB - 1) / RB, trips = (nch + 1) / 2;
    auto trip = [&](UpdChunk &ck) {
        const bool valid = r0 < rend;
        const int i = r0 + lrow;
        double cc[8];
#pragma unroll
        for (int q = 0; q < 8; q++) cc[q] = ck.c[q];
        upd_chunk_v_to_lds(ck, i, mp, nbp, s_pd, g1, lrow, lcg, Vs, valid && r0 > pfrom && r0 + RB <= mp);
        __syncthreads();                                           // (the first trip: Ws too)
        if (r0 + 4 * RB < rend) upd_chunk_load(ck, Vg, Cg, ld, i + 4 * RB, mp, nbp, nc, lcg);      // (this buffer's next trip)
        d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const double a = Vs[(4 * kk + l4) * VS + 16 * wid + l15];
            const double b0 = Ws[(4 * kk + l4) * WS + l15];
            const double b1 = Ws[(4 * kk + l4) * WS + 16 + l15];
            u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, u0, 0, 0, 0);
            u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, u1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * wid + l4 + 4 * r;
            Cs[l15 * VS + row] = u0[r];
            Cs[(16 + l15) * VS + row] = u1[r];
        }
        __syncthreads();
        if (valid && i < mp) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int col = lcg * 8 + q;
                if (col < nc) Cg[i + col * ld] = cc[q] - Cs[col * VS + lrow];
            }
        }
        r0 += 2 * RB;
    };
    for (int j = 0; j < trips; j += 2) {
        trip(ck0);
        if (j + 1 < trips) trip(ck1);
    }
}

// A launch: the panel pipeline's workgroups (blockIdx.z = 0: front blockIdx.x, column group blockIdx.y) and, behind them in dispatch
// order, k_upd_c's tiles of the previous step's update beyond block 0 (blockIdx.z - 1 = front of THAT step's lists, blockIdx.x =
// column block, blockIdx.y = slab).
__global__ __launch_bounds__(NTP) void k_panel_pc(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int npan,
                                                  int nsub, int defer_ok, int lds_doubles, const int *__restrict__ uflist,
                                                  const int *__restrict__ uplist, int ucb0, const double *Wp,
                                                  const long long *__restrict__ uwpoff, int rspw)
{
    extern __shared__ double dyn_lds[];
    __shared__ PanelShared ps;
    __shared__ WaveShared wsh;
    if (blockIdx.z == 0) {
        if ((int)blockIdx.x >= npan || (int)blockIdx.y >= nsub) return;
        const unsigned long long t0c = clock64(), t0w = wall_clock64();
        dev_k_panel(c, flist, plist, blockIdx.x, blockIdx.y, nsub, defer_ok, lds_doubles, ps, wsh, dyn_lds);
        if ((c.dbg & 16) && c.dbgbuf && threadIdx.x == 0) {
            atomicAdd(&c.dbgbuf[44], clock64() - t0c); atomicAdd(&c.dbgbuf[45], wall_clock64() - t0w); atomicAdd(&c.dbgbuf[46], 1ull);
        }
        return;
    }
    dev_upd_c_h2(c, uflist, uplist, ucb0, Wp, uwpoff, (int)blockIdx.z - 1, blockIdx.x, blockIdx.y, rspw, dyn_lds, ps.stair);
}

// ------------------------------------------------------------------------------------------------
// T + column block 0 of a step in ONE launch with ONE meeting point (the chain's launch between two panels; k_upd_f with the
// Gram block needs two: the Gram slabs meet and build T, the block's slabs meet, wait for T, one of them forms T'W and hands it
// to the others -- ~25 us of dependent round trips for a 4000 x 32 block).  Here every slab workgroup of block 0 accumulates its
// partial V'V beside its partial V'C (one more MFMA accumulator on the V image it has staged anyway), stores both, takes a
// ticket; the last arriver raises the flag; then EVERY slab workgroup adds the partials in slab order, builds T (dev_T_from_gram)
// and T'W for itself -- redundant arithmetic instead of a second hand-off -- and applies from the registers.
// Arithmetic = k_upd_w (Gram block + block 0) + k_upd_c exactly: the same bits.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dev_k_upd_b0(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                             const long long *__restrict__ wpoff, int *wcnt, int *wflag, int epoch, int sl, int fi,
                                             double *dyn_lds)
{
    __shared__ int s_pd[STM_NB];
    __shared__ int s_ticket, s_ok;
    __shared__ double s_tau[STM_NB];
    const bool tl = (c.dbg & 32768) && c.dbgbuf && threadIdx.x == 0 && sl == 0;
    unsigned long long tl0 = tl ? wall_clock64() : 0;
#define B0TL(k) do { if (tl) { const unsigned long long t1 = wall_clock64(); atomicAdd(&c.dbgbuf[48 + (k)], t1 - tl0); tl0 = t1; } } while (0)
    const int f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    FrontNum *num = &c.fnum[f];
    const PanelDesc *pd = &num->pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    const int ncbf = stm_upd_ncb(s, p), nslf = stm_upd_nsl(s);
    if (ncbf <= 0) return;
    const int c0 = pd->pc0;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl * SLAB >= mp) return;
    const bool deferred = pd->t_deferred != 0;                     // T was left to the update by the panel kernel
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int lrow = tid & 63, lcg = tid >> 6;
    double *Vs = dyn_lds, *Cs = Vs + STM_NB * VS, *Ws = Cs + BN * VS;
    if (tid < STM_NB) {
        s_pd[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;
        s_tau[tid] = (tid < nbp) ? c.Tau[s.rp + pd->pk1 + tid] : 0.0;
    }
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int r00 = sl * SLAB, rend = min(mp, (sl + 1) * SLAB);
    UpdChunk ck[SLAB / RB];
#pragma unroll
    for (int q = 0; q < SLAB / RB; q++) upd_chunk_load(ck[q], Vg, Cg, ld, r00 + q * RB + lrow, mp, nbp, nc, lcg);
    const int nsl = (mp + SLAB - 1) / SLAB;
    const int gslot = (c.ypoff && c.ypoff[f] >= 0) ? min(ncbf, c.sweep) : ncbf;                         // (as in k_upd_w)
    double *Wslot = Wp + wpoff[fi];                                                                     // column block 0
    double *Gslot = Wp + wpoff[fi] + ((long long)gslot * nslf) * (STM_NB * BN);
    double *Tslot = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    __syncthreads();
    B0TL(0);
    // ---- phase 1: partial W1 = V(slab)' C(slab) and, for a deferred T, partial G = V(slab)' V(slab) ----
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc = {0, 0, 0, 0}, accg = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < SLAB / RB; q++) {
        const int r0 = r00 + q * RB;
        if (r0 < rend) {
            upd_chunk_to_lds(ck[q], r0 + lrow, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < RB / 4; kk++) {
                const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
                const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            if (deferred) {
#pragma unroll
                for (int kk = 0; kk < RB / 4; kk++) {
                    const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
                    const double b = Vs[(16 * ni + l15) * VS + 4 * kk + l4];      // (the Gram block of k_upd_w: C = V)
                    accg = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, accg, 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    double *s_G = Vs, *s_W1 = Vs + STM_NB * WS, *s_T = Cs;         // (the chunk images are free between the phases)
    B0TL(1);
    bool last = true;
    if (nsl > 1) {
        double *W = Wslot + (long long)sl * (STM_NB * BN), *G = Gslot + (long long)sl * (STM_NB * BN);
#pragma unroll
        for (int r = 0; r < 4; r++) st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[r]);
        if (deferred) {
#pragma unroll
            for (int r = 0; r < 4; r++) st_agent(&G[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], accg[r]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *cnt = wcnt + wpoff[fi] / (STM_NB * BN);
        int *flag = wflag + wpoff[fi] / (STM_NB * BN);
        if (tid == 0) {
            s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s_ticket == nsl - 1) {
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st_agent(flag, epoch);                             // (every partial was acknowledged before its ticket)
            }
        }
        __syncthreads();
        last = (s_ticket == nsl - 1);
        B0TL(2);
        if (!stm_wait_ge(flag, epoch, c.abort, &s_ok)) { if (tid == 0) STM_SET_PERR(c, num); return; }
        B0TL(3);
        for (int e = tid; e < STM_NB * BN; e += NT) {
            s_W1[(e / BN) * WS + (e % BN)] = stm_ordered_sum<true>(Wslot + e, STM_NB * BN, nsl);        // fixed order: deterministic
            if (deferred) s_G[(e / BN) * WS + (e % BN)] = stm_ordered_sum<true>(Gslot + e, STM_NB * BN, nsl);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            s_W1[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = acc[r];
            if (deferred) s_G[(16 * mi + l4 + 4 * r) * WS + 16 * ni + l15] = accg[r];
        }
    }
    __syncthreads();
    B0TL(4);
    double (*s_Tb)[STM_NB + 1] = reinterpret_cast<double (*)[STM_NB + 1]>(s_T);       // T(row, col) = s_Tb[row][col]
    if (deferred) {
        dev_T_from_gram(reinterpret_cast<double (*)[STM_NB + 1]>(s_G), s_Tb, s_tau, nbp, tid);
        if (last) {
            double *Tkeep = c.Tall ? c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB : nullptr;
            for (int e = tid; e < STM_NB * STM_NB; e += NT) {
                const int a = e % STM_NB, b = e / STM_NB;
                const double tv = (a <= b && a < nbp && b < nbp) ? s_Tb[a][b] : 0.0;
                Tslot[e] = tv;
                if (Tkeep) Tkeep[e] = tv;
            }
        }
    } else {
        for (int e = tid; e < STM_NB * STM_NB; e += NT) s_Tb[e % STM_NB][e / STM_NB] = Tslot[e];
        __syncthreads();
    }
    B0TL(5);
    // W2 = T' W1 (k_upd_c's prologue)
    {
        const d4 w2 = dev_w2_tile(s_T, WS, 1, s_W1, wid, lane);      // (T(q, l) = s_Tb[q][l]: zero below the diagonal and beyond nbp)
#pragma unroll
        for (int r = 0; r < 4; r++) Ws[(16 * (wid >> 1) + l4 + 4 * r) * WS + 16 * (wid & 1) + l15] = w2[r];
    }
    __syncthreads();
    B0TL(6);
    // ---- phase 2: C(slab) -= V(slab) W2 from the registers (k_upd_c) ----
#pragma unroll
    for (int q = 0; q < SLAB / RB; q++) {
        const int r0 = r00 + q * RB;
        if (r0 < rend) {
            const int i = r0 + lrow;
            upd_chunk_to_lds(ck[q], i, mp, nbp, nc, s_pd, g1, lrow, lcg, Vs, Cs);
            __syncthreads();
            d4 u0 = {0, 0, 0, 0}, u1 = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < STM_NB / 4; kk++) {
                const double a = Vs[(4 * kk + l4) * VS + 16 * wid + l15];
                const double b0 = Ws[(4 * kk + l4) * WS + l15];
                const double b1 = Ws[(4 * kk + l4) * WS + 16 + l15];
                u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, u0, 0, 0, 0);
                u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, u1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * wid + l4 + 4 * r;
                Cs[l15 * VS + row] -= u0[r];
                Cs[(16 + l15) * VS + row] -= u1[r];
            }
            __syncthreads();
            if (i < mp) {
#pragma unroll
                for (int qq = 0; qq < 8; qq++) {
                    const int col = lcg * 8 + qq;
                    if (col < nc) Cg[i + col * ld] = Cs[col * VS + lrow];
                }
            }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    B0TL(7);
    if (tl) atomicAdd(&c.dbgbuf[56], 1ull);
#undef B0TL
}

// k_upd_w's tiles as riders of the B launch: a workgroup takes rspw consecutive slabs of its column block (one descriptor chain for
// several tiles; the partial sum of every slab is formed and stored separately, exactly as by k_upd_w, so the ordered sums -- and
// the bits -- do not change), the loads run one chunk ahead across the slab boundaries, one ticket for all its slabs.
__device__ __forceinline__ void dev_upd_w_rider(const DevCtx &c, const int *__restrict__ flist, const int *__restrict__ plist, int cb0,
                                                double *Wp, const long long *__restrict__ wpoff, int *wcnt, int cb, int y, int rspw,
                                                int fi, double *dyn_lds)
{
    __shared__ int s_pd[STM_NB];
    __shared__ int s_ticket;
    const int f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels) return;
    const PanelDesc *pd = &c.fnum[f].pd[STM_PDI(p)];
    const int g1 = pd->pg1, mp = pd->pt - pd->pg1, nbp = pd->pnb;
    const int ncbf = stm_upd_ncb(s, p), nslf = stm_upd_nsl(s);
    if (cb0 + cb >= ncbf) return;
    const int c0 = pd->pc0 + (cb0 + cb) * BN;
    const int sl0 = y * rspw;
    if (nbp <= 0 || mp <= 0 || c0 >= s.fn || sl0 * SLAB >= mp) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    double *Vs = dyn_lds, *Cs = Vs + STM_NB * VS;
    if (tid < STM_NB) s_pd[tid] = (tid < nbp) ? pd->pdiag[tid] : STM_BIGROW;
    __syncthreads();
    const double *Vg = c.Farena + s.foff + g1 + (long long)pd->pk1 * ld;
    const double *Cg = c.Farena + s.foff + g1 + (long long)c0 * ld;
    const int mi = wid >> 1, ni = wid & 1;
    const int nsl = (mp + SLAB - 1) / SLAB;
    const int rend = min(mp, (sl0 + rspw) * SLAB);
    double *W = Wp + wpoff[fi] + ((long long)cb * nslf + sl0) * (STM_NB * BN);
    d4 acc = {0, 0, 0, 0};
    UpdChunk ck;
    upd_chunk_load(ck, Vg, Cg, ld, sl0 * SLAB + (tid & 63), mp, nbp, nc, tid >> 6);
    const int pfrom = (nc == BN) ? upd_plain_from(s_pd, g1, lane) : STM_BIGROW;
    int done = 0;
    for (int r0 = sl0 * SLAB; r0 < rend; r0 += RB) {
        upd_chunk_to_lds(ck, r0 + (tid & 63), mp, nbp, nc, s_pd, g1, tid & 63, tid >> 6, Vs, Cs, false, r0 > pfrom && r0 + RB <= mp);
        __syncthreads();
        if (r0 + RB < rend) upd_chunk_load(ck, Vg, Cg, ld, r0 + RB + (tid & 63), mp, nbp, nc, tid >> 6);
#pragma unroll
        for (int kk = 0; kk < RB / 4; kk++) {
            const double a = Vs[(16 * mi + l15) * VS + 4 * kk + l4];
            const double b = Cs[(16 * ni + l15) * VS + 4 * kk + l4];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
        if (((r0 + RB) % SLAB) == 0 || r0 + RB >= rend) {          // the slab's partial sum is complete
            if (nsl == 1) {
#pragma unroll
                for (int r = 0; r < 4; r++) W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[r]);
            }
            W += STM_NB * BN;
            acc = d4{0, 0, 0, 0};
            done++;
        }
    }
    if (nsl == 1) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + cb;
    if (tid == 0) {
        s_ticket = __hip_atomic_fetch_add(cnt, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket + done == nsl) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_ticket + done != nsl) return;
    double *W0 = Wp + wpoff[fi] + ((long long)cb * nslf) * (STM_NB * BN);
    double v[STM_NB * BN / NT];
#pragma unroll
    for (int q = 0; q < STM_NB * BN / NT; q++) v[q] = stm_ordered_sum<true>(W0 + tid + q * NT, STM_NB * BN, nsl);   // fixed order
#pragma unroll
    for (int q = 0; q < STM_NB * BN / NT; q++) W0[tid + q * NT] = v[q];
}

// B launch, one meeting point: block 0 + T by dev_k_upd_b0 (blockIdx.z < nfr, slab blockIdx.x, blockIdx.y = 0) and, behind it, the
// riders: k_upd_w's tiles of the column blocks beyond block 0 (as k_upd_fw).
__global__ __launch_bounds__(NT, 2) void k_upd_b0w(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int nfr,
                                                   int maxsl, int ncbrest, double *Wp, const long long *__restrict__ wpoff, int *wcnt,
                                                   int *wflag, int epoch, double *Wp2, int *wcnt2, int rspw)
{
    extern __shared__ double dyn_lds[];
    if ((int)blockIdx.z < nfr) {
        if ((int)blockIdx.x >= maxsl || blockIdx.y >= 1) return;
        dev_k_upd_b0(c, flist, plist, Wp, wpoff, wcnt, wflag, epoch, blockIdx.x, blockIdx.z, dyn_lds);
        return;
    }
    if ((int)blockIdx.x >= ncbrest || (int)blockIdx.y * rspw >= maxsl) return;
    dev_upd_w_rider(c, flist, plist, 1, Wp2, wpoff, wcnt2, blockIdx.x, blockIdx.y, rspw, (int)blockIdx.z - nfr, dyn_lds);
}

// B launch: T + column block 0 of the step (k_upd_f's workgroups: blockIdx.z < nfr, slab blockIdx.x, blockIdx.y = 0 the Gram block /
// 1 block 0) and, behind them, k_upd_w's tiles of the column blocks beyond block 0 (blockIdx.z - nfr = front, blockIdx.x = column
// block - 1, blockIdx.y = slab) into the passengers' workspace.
__global__ __launch_bounds__(NT, 2) void k_upd_fw(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, int nfr,
                                                  int maxsl, int ncbrest, double *Wp, const long long *__restrict__ wpoff, int *wcnt,
                                                  int *wflag, int epoch, double *Wp2, int *wcnt2)
{
    extern __shared__ double dyn_lds[];
    if ((int)blockIdx.z < nfr) {
        if ((int)blockIdx.x >= maxsl || blockIdx.y >= 2) return;
        dev_k_upd_f(c, flist, plist, 0, 1, Wp, wpoff, wcnt, wflag, epoch, blockIdx.x, blockIdx.y, blockIdx.z, dyn_lds);
        return;
    }
    if ((int)blockIdx.x >= ncbrest || (int)blockIdx.y >= maxsl) return;
    dev_k_upd_w<true>(c, flist, plist, 1, 0, Wp2, wpoff, wcnt2, blockIdx.x, blockIdx.y, (int)blockIdx.z - nfr, ncbrest, dyn_lds);
}

// ------------------------------------------------------------------------------------------------
// Pair update (large fronts, stm_use_pair): the block reflectors of TWO consecutive panels a = p-1 (even) and b = p applied
// in ONE sweep over the columns beyond panel p+1 -- the trailing update of a 27 000-row front is bound by its three
// passes over C per 32 columns (W = V'C reads it, C -= V W reads and writes it); two panels per sweep make that 1.5.
//   k_upd_w2 : partial  W1 = V1(slab)' C(slab),  W2 = V2(slab)' C(slab)      (the SAME C: before either application)
//              last column block: C := V1  ->  G21 = V2'V1
//              the last slab workgroup of a column block to arrive adds the partials in slab order (as k_upd_w)
//   k_upd_c2 : Y1 = T1' W1,  Y2 = T2' (W2 - G21 Y1)  -- which is V2'(C - V1 Y1), the second application's own W --,
//              C(slab) -= V1 Y1 + V2 Y2
// Exactly H_b' H_a' C in exact arithmetic; the rounding differs from two separate updates, so WHICH fronts take it is a
// property of the front alone (symbolic), like the choice of the panel kernel.  The columns of the next TWO panels are
// updated panel by panel (k_upd_w / k_upd_c on column blocks 0, 1 after an even panel, block 0 after an odd one): the
// panel factorizations need them.  Rows: [g1a, max(pta, ptb)); V1 / V2 are masked by their own diagonals and ends.
// ------------------------------------------------------------------------------------------------
struct UpdChunk2 { double v1[8], v2[8], c[8]; };
__device__ __forceinline__ void upd2_chunk_load(UpdChunk2 &ck, const double *V1g, const double *V2g, const double *Cg, long long ld,
                                                int i, int mp, int nb1, int nb2, int nc, int lcg)
{
    const int ic = min(i, mp - 1);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        ck.v1[q] = V1g[ic + (long long)min(col, nb1 - 1) * ld];
        ck.v2[q] = V2g[ic + (long long)min(col, max(nb2, 1) - 1) * ld];
        ck.c[q] = Cg[ic + (long long)min(col, nc - 1) * ld];
    }
}
template <int STRIDE = VS>
__device__ __forceinline__ void upd2_chunk_to_lds(const UpdChunk2 &ck, int i, int mp, int mp1, int mp2, int nb1, int nb2, int nc,
                                                  const int *s_pd1, const int *s_pd2, int g1, int lrow, int lcg, double *Vs1,
                                                  double *Vs2, double *Cs, bool c_is_v1)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int col = lcg * 8 + q;
        const int d1 = s_pd1[col] - g1, d2 = s_pd2[col] - g1;   // (BIGROW beyond nb: everything masked)
        const double v1 = (i < mp1 && col < nb1 && i >= d1) ? ((i == d1) ? 1.0 : ck.v1[q]) : 0.0;
        const double v2 = (i < mp2 && col < nb2 && i >= d2) ? ((i == d2) ? 1.0 : ck.v2[q]) : 0.0;
        Vs1[col * STRIDE + lrow] = v1;
        Vs2[col * STRIDE + lrow] = v2;
        Cs[col * STRIDE + lrow] = c_is_v1 ? v1 : ((i < mp && col < nc) ? ck.c[q] : 0.0);
    }
}
// LDS column stride of the chunk images of k_upd_w2: with 68 doubles (2 x 68 = 8 mod 64 dwords) the 16-byte operand reads below
// are conflict-free in every 16-lane group of ds_read_b128 (MI355X_MICROARCH.md, LDS: bank = (a / 4) mod 64, groups
// {0-3,12-15,20-27}, ...); the 8-byte reads of the earlier form were merged by the compiler into ds_read2_b64, which runs at half
// the LDS rate with a 32-bank modulus: two-way conflicts on the 66-double stride (profiles/r03_a_*: 39 % of the LDS cycles)
#define VS2 (RB + 4)
typedef double d2v __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// slabs of 256 rows one workgroup of the pair kernels takes (the pair's rows in slabs -> 1, 2 or 4)
__device__ __forceinline__ int stm_pair_spw(int nsl, int tune)
{
    if (tune & 15) return 1 << ((tune & 15) - 1);
    return (nsl >= 32) ? 4 : (nsl >= 16) ? 2 : 1;
}
// the two panel descriptions of a pair and what both kernels derive from them (uniform per workgroup)
struct PairGeom { int g1, mp, mp1, mp2, nb1, nb2, k1a, k1b, pc0; };
__device__ __forceinline__ bool pair_geom(const FrontNum *num, int p, PairGeom &G)
{
    const PanelDesc *pa = &num->pd[STM_PDI(p - 1)], *pb = &num->pd[STM_PDI(p)];
    G.nb1 = pa->pnb;
    if (G.nb1 <= 0) return false;                               // (panel a did nothing: then b did nothing either)
    G.nb2 = pb->pnb > 0 ? pb->pnb : 0;
    G.g1 = pa->pg1;
    G.mp1 = pa->pt - pa->pg1;
    G.mp2 = G.nb2 > 0 ? pb->pt - pa->pg1 : 0;
    G.mp = max(G.mp1, G.mp2);
    G.k1a = pa->pk1; G.k1b = G.nb2 > 0 ? pb->pk1 : pa->pk1;
    G.pc0 = pa->pc0 + 2 * STM_NB;                               // first column beyond panel p+1
    return G.mp > 0;
}

__global__ __launch_bounds__(NT) void k_upd_w2(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff, int *wcnt)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_pd1[STM_NB], s_pd2[STM_NB];
    __shared__ int s_ticket;
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !(p & 1)) return;
    FrontNum *num = &c.fnum[f];
    PairGeom G;
    if (!pair_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);   // pair column blocks: beyond block 0 of panel p
    if (ncbp <= 0) return;
    const int cb = blockIdx.x, sl = blockIdx.y;
    const bool gram = (cb == (int)gridDim.x - 1);                    // last block of the launch: G21 = V2'V1
    if (!gram && cb >= ncbp) return;
    if (gram && G.nb2 <= 0) return;
    const int c0 = gram ? G.k1a : G.pc0 + cb * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    // A workgroup takes `spw` consecutive slabs of its column block (a property of the PANEL PAIR, like k_upd_c2's rule): its
    // life is then 16 chunks instead of 4 behind the same descriptor chain, ticket and partial-W store (at one slab per workgroup
    // those were most of a workgroup's 25 us for 3.4 us of MFMA work), and there are `spw` times fewer partials to add.
    const int nsl = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_pair_spw(nsl, c.tune);
    if (sl % spw) return;
    const int nc = gram ? G.nb1 : min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // (the chunk images are read 16 bytes at a time: they start on a 16-byte boundary of the LDS, one spare double is allocated)
    double *Vs1 = dyn_lds + ((((unsigned)(uintptr_t)dyn_lds) >> 3) & 1), *Vs2 = Vs1 + STM_NB * VS2, *Cs = Vs2 + STM_NB * VS2;
    const PanelDesc *pa = &num->pd[STM_PDI(p - 1)], *pb = &num->pd[STM_PDI(p)];
    if (tid < STM_NB) {
        s_pd1[tid] = (tid < G.nb1) ? pa->pdiag[tid] : STM_BIGROW;
        s_pd2[tid] = (tid < G.nb2) ? pb->pdiag[tid] : STM_BIGROW;
    }
    __syncthreads();
    const double *Fb = c.Farena + s.foff + G.g1;
    const double *V1g = Fb + (long long)G.k1a * ld, *V2g = Fb + (long long)G.k1b * ld, *Cg = Fb + (long long)c0 * ld;
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    const int rend = min(G.mp, (sl + spw) * SLAB);
    UpdChunk2 ck;
    upd2_chunk_load(ck, V1g, V2g, Cg, ld, sl * SLAB + (tid & 63), G.mp, G.nb1, G.nb2, nc, tid >> 6);
    // The K index of an MFMA is a summation index: lane group l4 takes the rows 8 kk + 2 l4 and 8 kk + 2 l4 + 1 of a chunk for
    // two consecutive MFMAs, so that both operands of both come from ONE 16-byte LDS read each (three ds_read_b128 per four
    // MFMAs instead of six 8-byte reads).
    const d2v *A1p = reinterpret_cast<const d2v *>(Vs1 + (16 * mi + l15) * VS2 + 2 * l4);
    const d2v *A2p = reinterpret_cast<const d2v *>(Vs2 + (16 * mi + l15) * VS2 + 2 * l4);
    const d2v *Bp = reinterpret_cast<const d2v *>(Cs + (16 * ni + l15) * VS2 + 2 * l4);
    for (int r0 = sl * SLAB; r0 < rend; r0 += RB) {
        upd2_chunk_to_lds<VS2>(ck, r0 + (tid & 63), G.mp, G.mp1, G.mp2, G.nb1, G.nb2, nc, s_pd1, s_pd2, G.g1, tid & 63, tid >> 6, Vs1,
                               Vs2, Cs, gram);
        __syncthreads();
        if (r0 + RB < rend) upd2_chunk_load(ck, V1g, V2g, Cg, ld, r0 + RB + (tid & 63), G.mp, G.nb1, G.nb2, nc, tid >> 6);
#pragma unroll
        for (int kk = 0; kk < RB / 8; kk++) {
            const d2v a1 = A1p[4 * kk], a2 = A2p[4 * kk], b = Bp[4 * kk];
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, b.x, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.x, b.x, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, b.y, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2.y, b.y, acc2, 0, 0, 0);
        }
        __syncthreads();
    }
    // slot of (column block, slab group): two blocks, W1 then W2; the Gram block is column block ncbp
    const int ngrp = (nsl + spw - 1) / spw;
    double *W0 = Wp + wpoff[fi] + ((long long)(gram ? ncbp : cb) * stm_pair_slots(nslf, c.tune)) * (2 * STM_NB * BN);
    double *W = W0 + (long long)(sl / spw) * (2 * STM_NB * BN);
    if (ngrp == 1) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc1[r];
            W[STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc2[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        st_agent(&W[(16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc1[r]);
        st_agent(&W[STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc2[r]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + (gram ? ncbp : cb);
    if (tid == 0) {
        s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket == ngrp - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_ticket != ngrp - 1) return;
    double v[2 * STM_NB * BN / NT];
#pragma unroll
    for (int q = 0; q < 2 * STM_NB * BN / NT; q++) v[q] = stm_ordered_sum<true>(W0 + tid + q * NT, 2 * STM_NB * BN, ngrp);   // fixed order
#pragma unroll
    for (int q = 0; q < 2 * STM_NB * BN / NT; q++) W0[tid + q * NT] = v[q];
}

// Y of a pair, once per column block (between k_upd_w2 and k_upd_c2): Y1 = T1' W1, Y2 = T2' (W2 - G21 Y1) from the summed W of the
// block, the Gram block G21 of the same launch of k_upd_w2 and the two T factors; the NEGATED 64 x 32 Y goes to the front's Y area
// (DevCtx::Ypend, by absolute column block), in the layout k_upd_c2's lanes read their MFMA operand from.  (It used to be the prologue of EVERY workgroup of
// k_upd_c2: five 8 KB images through LDS and three 32-step triangular loops, 6-8 us in front of 13 us of MFMA work.)
__global__ __launch_bounds__(NT) void k_upd_y2(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff)
{
    extern __shared__ double dyn_lds[];
    const int fi = blockIdx.y, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !(p & 1)) return;
    FrontNum *num = &c.fnum[f];
    PairGeom G;
    const bool live = pair_geom(num, p, G);
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);
    const int cb = blockIdx.x;
    const int tid = threadIdx.x;
    if (!live) return;
    if (cb >= ncbp || G.pc0 + cb * BN >= s.fn) return;
    double *s_W1 = dyn_lds, *s_W2 = s_W1 + STM_NB * WS, *s_T1 = s_W2 + STM_NB * WS, *s_T2 = s_T1 + STM_NB * WS,
           *s_G = s_T2 + STM_NB * WS, *s_Y1 = s_G + STM_NB * WS;
    const int nslp = stm_pair_slots(nslf, c.tune);
    double *W0 = Wp + wpoff[fi] + ((long long)cb * nslp) * (2 * STM_NB * BN);
    const double *Gr = Wp + wpoff[fi] + ((long long)ncbp * nslp) * (2 * STM_NB * BN) + STM_NB * BN;     // W2 part of the Gram block
    const double *T1 = c.Tws + (long long)STM_TSLOT(c.tslot[f], p - 1) * STM_NB * STM_NB;
    const double *T2 = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    const bool has2 = G.nb2 > 0;
    for (int e = tid; e < STM_NB * BN; e += NT) {
        s_W1[(e / BN) * WS + (e % BN)] = W0[e];
        s_W2[(e / BN) * WS + (e % BN)] = has2 ? W0[STM_NB * BN + e] : 0.0;
        s_G[(e / BN) * WS + (e % BN)] = has2 ? Gr[e] : 0.0;                    // G21(a, b) = v2_a' v1_b
        s_T1[(e / STM_NB) * WS + (e % STM_NB)] = T1[e];                          // s_T[col][row] = T(row, col)
        s_T2[(e / STM_NB) * WS + (e % STM_NB)] = has2 ? T2[e] : 0.0;
    }
    __syncthreads();
    const int l = tid & 31, cg = tid >> 5;
    double y[4] = {0, 0, 0, 0};
    for (int q = 0; q <= l; q++) {                             // Y1 = T1' W1
        const double tq = s_T1[l * WS + q];
#pragma unroll
        for (int x = 0; x < 4; x++) y[x] += tq * s_W1[q * WS + cg * 4 + x];
    }
#pragma unroll
    for (int x = 0; x < 4; x++) s_Y1[l * WS + cg * 4 + x] = y[x];
    __syncthreads();
    double z[4];                                               // Z = W2 - G21 Y1  ( = V2'(C - V1 Y1) )
#pragma unroll
    for (int x = 0; x < 4; x++) z[x] = s_W2[l * WS + cg * 4 + x];
    for (int b = 0; b < STM_NB; b++) {
        const double gq = s_G[l * WS + b];
#pragma unroll
        for (int x = 0; x < 4; x++) z[x] -= gq * s_Y1[b * WS + cg * 4 + x];
    }
    __syncthreads();                                           // (everyone has read W2 before Z replaces it)
#pragma unroll
    for (int x = 0; x < 4; x++) s_W2[l * WS + cg * 4 + x] = z[x];
    __syncthreads();
    double y2[4] = {0, 0, 0, 0};
    for (int q = 0; q <= l; q++) {                             // Y2 = T2' Z
        const double tq = s_T2[l * WS + q];
#pragma unroll
        for (int x = 0; x < 4; x++) y2[x] += tq * s_W2[q * WS + cg * 4 + x];
    }
    double *Yo = c.Ypend + c.ypoff[f] + (long long)((G.pc0 + cb * BN) >> 5) * (2 * STM_NB * BN);          // by absolute column block
#pragma unroll
    for (int x = 0; x < 4; x++) {
        Yo[l * BN + cg * 4 + x] = -y[x];
        Yo[(STM_NB + l) * BN + cg * 4 + x] = -y2[x];
    }
}

__global__ __launch_bounds__(NT) void k_upd_c2(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist,
                                               const double *Wp, const long long *__restrict__ wpoff)
{
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || !(p & 1)) return;
    const FrontNum *num = &c.fnum[f];
    PairGeom G;
    if (!pair_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1;
    const int cb = blockIdx.x, sl = blockIdx.y;
    if (cb >= ncbp) return;
    const int c0 = G.pc0 + cb * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    const int nsl_all = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_pair_spw(nsl_all, c.tune);
    if (sl % spw) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const PanelDesc *pa = &num->pd[STM_PDI(p - 1)], *pb = &num->pd[STM_PDI(p)];
    const double *Fb = c.Farena + s.foff + G.g1;
    const double *__restrict__ V1g = Fb + (long long)G.k1a * ld, *__restrict__ V2g = Fb + (long long)G.k1b * ld;
    double *__restrict__ Cg = c.Farena + s.foff + G.g1 + (long long)c0 * ld;
    const double *__restrict__ Yn = c.Ypend + c.ypoff[f] + (long long)(c0 >> 5) * (2 * STM_NB * BN);      // -Y, [64][32] (k_upd_y2)
    // Round 3: the application runs WITHOUT LDS and without barriers.  The product is formed transposed,
    //     D(col, row) = C(row, col) - sum_k Y(k, col) V(row, k),
    // so that (i) C is the accumulator operand the MFMA starts from -- lane (l15, l4) holds D(col l4 + 4 r, row l15): for a
    // fixed r the 64 lanes touch 4 columns x 16 consecutive rows, i.e. four full 128-byte segments of the column-major front,
    // loaded from and stored to global memory directly in that layout -- and (ii) V is the B operand B(k, row): 16 consecutive
    // rows of 4 reflector columns per load, again whole segments.  Y (negated, from k_upd_y2) is the A operand and stays in
    // registers for the whole workgroup.  A wave takes every fourth 16-row tile of the workgroup's rows.
    //
    // Two forms of a tile.  INTERIOR tiles -- every row below all 64 unit diagonals, inside both panels' row ranges, a full
    // 32-column block: all but the first 64 rows of a pair, its last rows and the front's last column block -- run a pipeline
    // with NO predication: 24 unconditional loads of the next tile, 32 MFMAs, 8 unconditional stores.  That matters more than
    // the saved mask arithmetic: with a branch around any load or store of the loop the compiler no longer knows how many
    // memory operations are outstanding and waits for vmcnt(0) at the top of EVERY tile -- i.e. for the stores it has just
    // issued (measured: 325 ms per factorization of c5mid for this kernel, 210 ms without its stores, 165 ms with neither
    // loads nor stores).  The other tiles take the general form (masks, clamped loads, predicated stores), one at a time.
    double yn0[2 * STM_NB / 4], yn1[2 * STM_NB / 4];
#pragma unroll
    for (int kk = 0; kk < 2 * STM_NB / 4; kk++) {
        yn0[kk] = Yn[(4 * kk + l4) * BN + l15];
        yn1[kk] = Yn[(4 * kk + l4) * BN + 16 + l15];
    }
    // last row (relative to g1) that holds a unit diagonal; BIGROW: a dead reflector.  One load per lane and a wave reduction (round 4:
    // it was a loop of 64 dependent loads in front of every workgroup's tiles)
    int dmax;
    {
        const int q = lane & 31;
        const int d = (lane < 32) ? ((q < G.nb1) ? pa->pdiag[q] - G.g1 : STM_BIGROW) : ((q < G.nb2) ? pb->pdiag[q] - G.g1 : STM_BIGROW);
        dmax = wave_max_int(d);
    }
    const int rbeg = sl * SLAB, rend = min(G.mp, (sl + spw) * SLAB);
    const int ntile = (rend - rbeg + 15) >> 4;
    const int rfull = min(min(G.mp1, G.mp2), rend);     // rows below this are inside both panels
    // tiles [t_lo, t_hi) are interior (nothing if the column block is ragged or some reflector is dead)
    int t_lo = (dmax >= STM_BIGROW || nc < BN) ? ntile : max(0, (dmax + 1 - rbeg + 15) >> 4);
    int t_hi = (rfull - rbeg) >> 4;
    if (t_lo > ntile) t_lo = ntile;
    if (t_hi < t_lo) t_hi = t_lo;
    struct Tile { double c0[4], c1[4], v1[STM_NB / 4], v2[STM_NB / 4]; };
    // general form of one tile
    auto general_tile = [&](int tix) {
        const int i = rbeg + 16 * tix + l15;                                           // my row (relative to g1)
        const int row = min(i, G.mp - 1);                                              // (clamped: masked afterwards)
        Tile t;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            t.c0[r] = Cg[row + (long long)min(l4 + 4 * r, nc - 1) * ld];
            t.c1[r] = Cg[row + (long long)min(16 + l4 + 4 * r, nc - 1) * ld];
        }
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            t.v1[kk] = V1g[row + (long long)min(4 * kk + l4, G.nb1 - 1) * ld];
            t.v2[kk] = V2g[row + (long long)min(4 * kk + l4, max(G.nb2, 1) - 1) * ld];
        }
        d4 a0 = {t.c0[0], t.c0[1], t.c0[2], t.c0[3]}, a1 = {t.c1[0], t.c1[1], t.c1[2], t.c1[3]};
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const int col = 4 * kk + l4;
            const int d = (col < G.nb1) ? pa->pdiag[col] - G.g1 : STM_BIGROW;
            const double bv = (i < G.mp1 && i >= d) ? ((i == d) ? 1.0 : t.v1[kk]) : 0.0;
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[kk], bv, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[kk], bv, a1, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < STM_NB / 4; kk++) {
            const int col = 4 * kk + l4;
            const int d = (col < G.nb2) ? pb->pdiag[col] - G.g1 : STM_BIGROW;
            const double bv = (i < G.mp2 && i >= d) ? ((i == d) ? 1.0 : t.v2[kk]) : 0.0;
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[STM_NB / 4 + kk], bv, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[STM_NB / 4 + kk], bv, a1, 0, 0, 0);
        }
        if (i < G.mp) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (l4 + 4 * r < nc) Cg[i + (long long)(l4 + 4 * r) * ld] = a0[r];
                if (16 + l4 + 4 * r < nc) Cg[i + (long long)(16 + l4 + 4 * r) * ld] = a1[r];
            }
        }
    };
    int tix = wid;
    for (; tix < t_lo; tix += NW) general_tile(tix);
    if (tix < t_hi) {
        // interior tiles, four per trip: lane addresses advance by 64 rows per tile.  C comes from HBM and is requested THREE
        // tiles ahead (four register images used in turn -- the trip is unrolled so that none is ever copied: copying the
        // destination of a load in flight would wait for it), V (L2) one tile ahead and BEFORE the C loads of the same step:
        // the memory counter retires in order, so the wait for V(t) covers nothing younger -- C(t+2), the stores of tile
        // t-1, V(t+1) and C(t+3) stay in flight (a counted vmcnt(40)).  The scheduling barriers keep that issue order.
        // Requests beyond this wave's last tile of the trips are clamped to it (loaded again, never used); the 1-3 interior
        // tiles left after the last whole trip take the general form.
        const int nint = (t_hi - 1 - tix) / NW + 1, ntrip = nint >> 2;
        if (ntrip > 0) {
            const double *cp = Cg + (rbeg + 16 * tix + l15) + (long long)l4 * ld;
            const double *v1p = V1g + (rbeg + 16 * tix + l15) + (long long)l4 * ld;
            const double *v2p = V2g + (rbeg + 16 * tix + l15) + (long long)l4 * ld;
            const long long ld4 = 4 * ld;
            struct TC { double c0[4], c1[4]; };
            struct TV { double v1[STM_NB / 4], v2[STM_NB / 4]; };
            auto load_c = [&](TC &t, int off) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    t.c0[r] = cp[off + r * ld4];
                    t.c1[r] = cp[off + (4 + r) * ld4];
                }
            };
            auto load_v = [&](TV &t, int off) {
#pragma unroll
                for (int kk = 0; kk < STM_NB / 4; kk++) {
                    t.v1[kk] = v1p[off + kk * ld4];
                    t.v2[kk] = v2p[off + kk * ld4];
                }
            };
            TC cb4[4];
            TV vb2[2];
            const int step = 16 * NW, offlast = step * (4 * ntrip - 1);
            load_v(vb2[0], 0);
            load_c(cb4[0], 0);
            load_c(cb4[1], min(step, offlast));
            load_c(cb4[2], min(2 * step, offlast));
            for (int trip = 0; trip < ntrip; trip++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int off = step * (4 * trip + q);
                    load_v(vb2[(q + 1) & 1], min(off + step, offlast));
                    load_c(cb4[(q + 3) & 3], min(off + 3 * step, offlast));
                    __builtin_amdgcn_sched_barrier(0);
                    const TC &tc = cb4[q];
                    const TV &tv = vb2[q & 1];
                    d4 a0 = {tc.c0[0], tc.c0[1], tc.c0[2], tc.c0[3]}, a1 = {tc.c1[0], tc.c1[1], tc.c1[2], tc.c1[3]};
#pragma unroll
                    for (int kk = 0; kk < STM_NB / 4; kk++) {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[kk], tv.v1[kk], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[kk], tv.v1[kk], a1, 0, 0, 0);
                    }
#pragma unroll
                    for (int kk = 0; kk < STM_NB / 4; kk++) {
                        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn0[STM_NB / 4 + kk], tv.v2[kk], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(yn1[STM_NB / 4 + kk], tv.v2[kk], a1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    double *sp = const_cast<double *>(cp) + off;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        sp[r * ld4] = a0[r];
                        sp[(4 + r) * ld4] = a1[r];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tix += 4 * ntrip * NW;
        }
    }
    for (; tix < ntile; tix += NW) general_tile(tix);
}

// ------------------------------------------------------------------------------------------------
// Quad update (round 4, options.pair_update = 4): the block reflectors of FOUR consecutive panels q0 = p-3 .. q3 = p
// (p = 3 mod 4) in ONE sweep over the columns beyond panel p+1.  The sweeps of a 27 000-row front are bound by what they
// move per MFMA: a 16-row tile of k_upd_c2 brings 4 KB of C in, 4 KB out and 8 KB of V for 32 MFMAs; with four panels it is
// the same C for 64 MFMAs (0.75 instead of 1.5 passes over the trailing columns per panel).
//   k_upd_wq : partial W_i = V_i(slab)' C(slab), i = 1..4 (the same C: before any application); three more column blocks with
//              C := V_1, V_2, V_3 give the Gram blocks G_ij = V_i'V_j (i > j)
//   k_upd_yq : Y_i = T_i' (W_i - sum_{j<i} G_ij Y_j)   -- V_i' of C after the applications before it --, negated, to Ypend
//   k_upd_cq : C(slab) -= sum_i V_i Y_i
// Exactly H_3' H_2' H_1' H_0' C in exact arithmetic.  The columns of the next FOUR panels are updated panel by panel (column
// blocks 0 .. 3-r after panel r of a quad).  Which fronts take it is symbolic (stmmqr_plan::pair_front), as for the pair.
// ------------------------------------------------------------------------------------------------
#define QP 4                       // panels per sweep
#define QRB 32                     // rows per chunk of k_upd_wq (five chunk images: 46 KB of LDS, three workgroups per CU)
#define VSQ (QRB + 4)              // column stride of the chunk images: 16-byte operand reads conflict free (as VS2)
#define YSQ 48                     // row stride of -Y in k_upd_cq's LDS: the two 32-lane halves of an 8-byte read on disjoint banks
// slabs of 256 rows one workgroup of the quad kernels takes: as the pair's rule, and 8 from 64 slabs on (configs[4] stand-in, 203 slabs:
// 3434 -> 3370 ms; c5mid, 106 slabs: +-0; 8 everywhere: c5mid 580 -> 596.  tune >> 8: another threshold, measurement sweeps)
__device__ __forceinline__ int stm_quad_spw(int nsl, int tune)
{
    if (tune & 15) return 1 << ((tune & 15) - 1);
    return (nsl >= ((tune >> 8) ? (tune >> 8) : 64)) ? 8 : stm_pair_spw(nsl, 0);
}
struct QuadGeom { int g1, mp, pc0; int mpi[QP], nb[QP], k1[QP]; };
// (entry i of a per-panel array for a run-time i: selects over constant indices, so that the arrays stay in registers)
__device__ __forceinline__ int qsel(const int (&a)[QP], int i)
{
    int r = a[0];
#pragma unroll
    for (int q = 1; q < QP; q++) r = (i == q) ? a[q] : r;
    return r;
}
__device__ __forceinline__ bool quad_geom(const FrontNum *num, int p, QuadGeom &G)
{
    const PanelDesc *p0 = &num->pd[STM_PDI(p - (QP - 1))];
    if (p0->pnb <= 0) return false;                             // (the first panel did nothing: then none of them did)
    G.g1 = p0->pg1;
    G.pc0 = p0->pc0 + QP * STM_NB;                              // first column beyond panel p+1
    G.mp = 0;
#pragma unroll
    for (int i = 0; i < QP; i++) {
        const PanelDesc *pi = &num->pd[STM_PDI(p - (QP - 1) + i)];
        const int nb = pi->pnb > 0 ? pi->pnb : 0;
        G.nb[i] = nb;
        G.mpi[i] = nb > 0 ? pi->pt - G.g1 : 0;
        G.k1[i] = nb > 0 ? pi->pk1 : p0->pk1;
        G.mp = max(G.mp, G.mpi[i]);
    }
    return G.mp > 0;
}

__global__ __launch_bounds__(NT, 3) void k_upd_wq(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff, int *wcnt)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_pd[QP][STM_NB];
    __shared__ int s_ticket;
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || (p & (QP - 1)) != QP - 1) return;
    FrontNum *num = &c.fnum[f];
    QuadGeom G;
    if (!quad_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);   // sweep column blocks: beyond block 0 of panel p
    if (ncbp <= 0) return;
    const int cbx = blockIdx.x, sl = blockIdx.y;
    const int gj = cbx - ncbp;                                       // >= 0: Gram block, C := V_gj
    if (gj >= QP - 1) return;
    const bool gram = gj >= 0;
    if (gram && qsel(G.nb, gj + 1) <= 0) return;                     // (no panel behind it: nobody reads its products)
    const int c0 = gram ? qsel(G.k1, gj) : G.pc0 + cbx * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    const int nsl = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_quad_spw(nsl, c.tune);
    if (sl % spw) return;
    const int nc = gram ? qsel(G.nb, gj) : min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // five chunk images (the V of the four panels and C) of 32 rows: 46 KB, three workgroups per CU.  (Measured against it: chunks of 64
    // rows in two phases -- panels 0, 1 then 2, 3 against one C image, three images as k_upd_w2, 168 registers -- c5mid 607 -> 621 ms.)
    double *Vs = dyn_lds + ((((unsigned)(uintptr_t)dyn_lds) >> 3) & 1), *Cs = Vs + QP * STM_NB * VSQ;
    __shared__ int s_dm[2];
    if (tid < 2 * 64) {                                              // (the four panels' diagonals; their maximum: BIGROW if any is missing)
        const int i = tid >> 5, q = tid & 31;
        const int d = (q < qsel(G.nb, i)) ? num->pd[STM_PDI(p - (QP - 1) + i)].pdiag[q] : STM_BIGROW;
        s_pd[i][q] = d;
        const int dm = wave_max_int(d);
        if (lane == 0) s_dm[wid] = dm;
    }
    __syncthreads();
    // chunks below every unit diagonal, inside all four panels' rows, with full column blocks need no masks (all but the first 128 rows and
    // the last rows of a quad): their staging is a plain copy -- the masks are ~8 VALU instructions per element, 160 per thread and chunk
    const int dmax = max(s_dm[0], s_dm[1]);
    const int rfull = min(min(G.mpi[0], G.mpi[1]), min(G.mpi[2], G.mpi[3]));
    const bool plain_ok = dmax < STM_BIGROW && nc == BN;
    const double *Fb = c.Farena + s.foff + G.g1;
    const double *Cg = Fb + (long long)c0 * ld;
    const int mi = wid >> 1, ni = wid & 1;
    d4 acc[QP];
#pragma unroll
    for (int i = 0; i < QP; i++) acc[i] = (d4){0, 0, 0, 0};
    const int rend = min(G.mp, (sl + spw) * SLAB);
    const int lrow = tid & (QRB - 1), lcg = tid / QRB;              // staging: thread = (row of the chunk, group of 4 columns)
    double cv[QP][4], cc[4];
    auto chunk_load = [&](int i) {
        const int ic = min(i, G.mp - 1);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = lcg * 4 + q;
#pragma unroll
            for (int pi = 0; pi < QP; pi++) cv[pi][q] = Fb[ic + (long long)(G.k1[pi] + min(col, max(G.nb[pi], 1) - 1)) * ld];
            cc[q] = Cg[ic + (long long)min(col, nc - 1) * ld];
        }
    };
    auto chunk_to_lds = [&](int i) {
        const int r0c = i - lrow;                                    // first row of the chunk (uniform)
        if (plain_ok && r0c > dmax - G.g1 && r0c + QRB <= rfull) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int col = lcg * 4 + q;
#pragma unroll
                for (int pi = 0; pi < QP; pi++) Vs[(pi * STM_NB + col) * VSQ + lrow] = cv[pi][q];
                Cs[col * VSQ + lrow] = cc[q];                        // (a Gram block's C columns ARE V_gj's: loaded from the same place)
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int col = lcg * 4 + q;
            double vg = 0.0;
#pragma unroll
            for (int pi = 0; pi < QP; pi++) {
                const int d = s_pd[pi][col] - G.g1;                  // (BIGROW beyond nb: everything masked)
                const double v = (i < G.mpi[pi] && col < G.nb[pi] && i >= d) ? ((i == d) ? 1.0 : cv[pi][q]) : 0.0;
                Vs[(pi * STM_NB + col) * VSQ + lrow] = v;
                vg = (pi == gj) ? v : vg;
            }
            Cs[col * VSQ + lrow] = gram ? vg : ((i < G.mp && col < nc) ? cc[q] : 0.0);
        }
    };
    chunk_load(sl * SLAB + lrow);
    const d2v *Ap = reinterpret_cast<const d2v *>(Vs + (16 * mi + l15) * VSQ + 2 * l4);
    const d2v *Bp = reinterpret_cast<const d2v *>(Cs + (16 * ni + l15) * VSQ + 2 * l4);
    for (int r0 = sl * SLAB; r0 < rend; r0 += QRB) {
        chunk_to_lds(r0 + lrow);
        __syncthreads();
        if (r0 + QRB < rend) chunk_load(r0 + QRB + lrow);
#pragma unroll
        for (int kk = 0; kk < QRB / 8; kk++) {
            const d2v b = Bp[4 * kk];
#pragma unroll
            for (int pi = 0; pi < QP; pi++) {
                const d2v a = Ap[pi * (STM_NB * VSQ / 2) + 4 * kk];
                acc[pi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc[pi], 0, 0, 0);
                acc[pi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc[pi], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // slot of (column block, slab group): QP blocks W_1 .. W_4; the Gram blocks are column blocks ncbp .. ncbp + 2
    const int ngrp = (nsl + spw - 1) / spw;
    double *W0 = Wp + wpoff[fi] + ((long long)cbx * stm_quad_slots(nslf, c.tune)) * (QP * STM_NB * BN);
    double *W = W0 + (long long)(sl / spw) * (QP * STM_NB * BN);
    if (ngrp == 1) {
#pragma unroll
        for (int pi = 0; pi < QP; pi++)
#pragma unroll
            for (int r = 0; r < 4; r++) W[pi * STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15] = acc[pi][r];
        return;
    }
#pragma unroll
    for (int pi = 0; pi < QP; pi++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_agent(&W[pi * STM_NB * BN + (16 * mi + l4 + 4 * r) * BN + 16 * ni + l15], acc[pi][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *cnt = wcnt + wpoff[fi] / (STM_NB * BN) + cbx;
    if (tid == 0) {
        s_ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s_ticket == ngrp - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (s_ticket != ngrp - 1) return;
    double v[QP * STM_NB * BN / NT];
#pragma unroll
    for (int q = 0; q < QP * STM_NB * BN / NT; q++) v[q] = stm_ordered_sum<true>(W0 + tid + q * NT, QP * STM_NB * BN, ngrp);   // fixed order
#pragma unroll
    for (int q = 0; q < QP * STM_NB * BN / NT; q++) W0[tid + q * NT] = v[q];
}

// -Y of a quad, once per column block (between k_upd_wq and k_upd_cq)
__global__ __launch_bounds__(NT) void k_upd_yq(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist, double *Wp,
                                               const long long *__restrict__ wpoff)
{
    extern __shared__ double dyn_lds[];
    const int fi = blockIdx.y, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || (p & (QP - 1)) != QP - 1) return;
    FrontNum *num = &c.fnum[f];
    QuadGeom G;
    if (!quad_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1, nslf = stm_upd_nsl(s);
    const int cb = blockIdx.x, tid = threadIdx.x;
    if (cb >= ncbp || G.pc0 + cb * BN >= s.fn) return;
    double *s_Y = dyn_lds, *s_Z = s_Y + QP * STM_NB * WS, *s_T = s_Z + STM_NB * WS, *s_G = s_T + STM_NB * WS;
    const long long nslp = stm_quad_slots(nslf, c.tune);
    const double *W0 = Wp + wpoff[fi] + ((long long)cb * nslp) * (QP * STM_NB * BN);
    double *Yo = c.Ypend + c.ypoff[f] + (long long)((G.pc0 + cb * BN) >> 5) * (QP * STM_NB * BN);         // by absolute column block
    const int l = tid & 31, cg = tid >> 5;
#pragma unroll 1
    for (int i = 0; i < QP; i++) {
        double y[4] = {0, 0, 0, 0};
        if (qsel(G.nb, i) > 0) {                                // (uniform)
            double z[4];
#pragma unroll
            for (int x = 0; x < 4; x++) z[x] = W0[i * STM_NB * BN + l * BN + cg * 4 + x];
#pragma unroll 1
            for (int j = 0; j < i; j++) {                       // Z = W_i - sum_j G_ij Y_j  ( = V_i' of C after the applications 0 .. i-1 )
                const double *Gr = Wp + wpoff[fi] + ((long long)(ncbp + j) * nslp) * (QP * STM_NB * BN) + i * STM_NB * BN;   // G_ij(a, b) = v_ia' v_jb
                __syncthreads();
                for (int e = tid; e < STM_NB * BN; e += NT) s_G[(e / BN) * WS + (e % BN)] = Gr[e];
                __syncthreads();
                const double *Yj = s_Y + j * STM_NB * WS;
#pragma unroll
                for (int b = 0; b < STM_NB; b++) {
                    const double gq = s_G[l * WS + b];
#pragma unroll
                    for (int x = 0; x < 4; x++) z[x] -= gq * Yj[b * WS + cg * 4 + x];
                }
            }
            const double *T = c.Tws + STM_TSLOT(c.tslot[f], p - (QP - 1) + i) * STM_NB * STM_NB;
            __syncthreads();
#pragma unroll
            for (int x = 0; x < 4; x++) s_Z[l * WS + cg * 4 + x] = z[x];
            for (int e = tid; e < STM_NB * STM_NB; e += NT) s_T[(e / STM_NB) * WS + (e % STM_NB)] = T[e];     // s_T[col][row] = T(row, col)
            __syncthreads();
#pragma unroll
            for (int q = 0; q < STM_NB; q++) {                  // Y_i = T_i' Z  (the trip count as a predicate: the LDS reads up front)
                const double tq = s_T[l * WS + q];
                if (q <= l) {
#pragma unroll
                    for (int x = 0; x < 4; x++) y[x] += tq * s_Z[q * WS + cg * 4 + x];
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            s_Y[i * STM_NB * WS + l * WS + cg * 4 + x] = y[x];
            Yo[(i * STM_NB + l) * BN + cg * 4 + x] = -y[x];
        }
    }
}

__global__ __launch_bounds__(NT, 2) void k_upd_cq(DevCtx c, const int *__restrict__ flist, const int *__restrict__ plist)
{
    __shared__ double s_Y[QP * STM_NB * YSQ];
    const int fi = blockIdx.z, f = flist[fi], p = plist[fi];
    const FrontSym s = c.fs[f];
    if (p >= s.npanels || (p & (QP - 1)) != QP - 1) return;
    const FrontNum *num = &c.fnum[f];
    QuadGeom G;
    if (!quad_geom(num, p, G)) return;
    const int ncbp = stm_upd_ncb(s, p) - 1;
    const int cb = blockIdx.x, sl = blockIdx.y;
    if (cb >= ncbp) return;
    const int c0 = G.pc0 + cb * BN;
    if (c0 >= s.fn || sl * SLAB >= G.mp) return;
    const int nsl_all = (G.mp + SLAB - 1) / SLAB;
    const int spw = stm_quad_spw(nsl_all, c.tune);
    if (sl % spw) return;
    const int nc = min(BN, s.fn - c0);
    const long long ld = s.ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const double *__restrict__ Fb = c.Farena + s.foff + G.g1;
    double *__restrict__ Cg = c.Farena + s.foff + G.g1 + (long long)c0 * ld;
    const double *__restrict__ Yn = c.Ypend + c.ypoff[f] + (long long)(c0 >> 5) * (QP * STM_NB * BN);    // -Y, [128][32] (k_upd_yq)
    // As k_upd_c2: no chunk images and no barriers in the loop -- D(col, row) = C(row, col) - sum_k Y(k, col) V(row, k), C the
    // accumulator operand, V the B operand, both straight from global memory in whole 128-byte segments.  -Y is the A operand: 128 x 32
    // here, kept in LDS (in registers it would leave one wave per SIMD; measured, timing only: 1.52 x the time of k_upd_c2 for twice its
    // MFMAs in registers, 1.33 x from LDS at two waves).
    for (int e = tid; e < QP * STM_NB * BN; e += NT) s_Y[(e >> 5) * YSQ + (e & 31)] = Yn[e];
    int dmax = -1;                                      // last row (relative to g1) that holds a unit diagonal; BIGROW: a dead reflector
    int rfull = G.mp;
#pragma unroll
    for (int i = 0; i < QP / 2; i++) {                  // (one load per lane and a wave reduction for two panels)
        const int pi = 2 * i + (lane >> 5), q = lane & 31;
        const int d = (q < qsel(G.nb, pi)) ? num->pd[STM_PDI(p - (QP - 1) + pi)].pdiag[q] - G.g1 : STM_BIGROW;
        dmax = max(dmax, wave_max_int(d));
    }
#pragma unroll
    for (int i = 0; i < QP; i++) rfull = min(rfull, G.mpi[i]);
    __syncthreads();
    const double *y0p = s_Y + l4 * YSQ + l15, *y1p = y0p + 16;
    const int rbeg = sl * SLAB, rend = min(G.mp, (sl + spw) * SLAB);
    const int ntile = (rend - rbeg + 15) >> 4;
    rfull = min(rfull, rend);                           // rows below this are inside all four panels
    // tiles [t_lo, t_hi) are interior (nothing if the column block is ragged, a panel is short of 32 columns or some reflector is dead)
    int t_lo = (dmax >= STM_BIGROW || nc < BN) ? ntile : max(0, (dmax + 1 - rbeg + 15) >> 4);
    int t_hi = (rfull - rbeg) >> 4;
    if (t_lo > ntile) t_lo = ntile;
    if (t_hi < t_lo) t_hi = t_lo;
    t_lo = __builtin_amdgcn_readfirstlane(t_lo);        // (uniform in fact; dmax came through a wave reduction)
    t_hi = __builtin_amdgcn_readfirstlane(t_hi);
    // general form of one tile (masks, clamped loads, predicated stores)
    auto general_tile = [&](int tix) {
        const int i = rbeg + 16 * tix + l15;                                           // my row (relative to g1)
        const int row = min(i, G.mp - 1);                                              // (clamped: masked afterwards)
        d4 a0, a1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            a0[r] = Cg[row + (long long)min(l4 + 4 * r, nc - 1) * ld];
            a1[r] = Cg[row + (long long)min(16 + l4 + 4 * r, nc - 1) * ld];
        }
#pragma unroll 1
        for (int pi = 0; pi < QP; pi++) {
            const PanelDesc *pp = &num->pd[STM_PDI(p - (QP - 1) + pi)];
            const double *Vg = Fb + (long long)qsel(G.k1, pi) * ld;
            const int nbi = qsel(G.nb, pi), mpi = qsel(G.mpi, pi);
            double v[STM_NB / 4];
#pragma unroll
            for (int kk = 0; kk < STM_NB / 4; kk++) v[kk] = Vg[row + (long long)min(4 * kk + l4, max(nbi, 1) - 1) * ld];
#pragma unroll
            for (int kk = 0; kk < STM_NB / 4; kk++) {
                const int col = 4 * kk + l4;
                const int d = (col < nbi) ? pp->pdiag[col] - G.g1 : STM_BIGROW;
                const double bv = (i < mpi && i >= d) ? ((i == d) ? 1.0 : v[kk]) : 0.0;
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0p[(pi * STM_NB + 4 * kk) * YSQ], bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1p[(pi * STM_NB + 4 * kk) * YSQ], bv, a1, 0, 0, 0);
            }
        }
        if (i < G.mp) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (l4 + 4 * r < nc) Cg[i + (long long)(l4 + 4 * r) * ld] = a0[r];
                if (16 + l4 + 4 * r < nc) Cg[i + (long long)(16 + l4 + 4 * r) * ld] = a1[r];
            }
        }
    };
    for (int tix = wid; tix < t_lo; tix += NW) general_tile(tix);
    // Interior rows in SUPER tiles of 32 rows: lane l15 takes the rows 2 l15 and 2 l15 + 1, so that every load and store of the loop is a
    // 16-byte access (half the vector-memory instructions for the same bytes: the texture-address path was 71 % busy in k_upd_c2) -- the
    // even rows are one MFMA tile, the odd rows another, both fed from the same registers.  V comes in two halves (panels 0, 1 and
    // panels 2, 3 of the quad: 64 registers each), each requested again as soon as its MFMAs are issued; C one super tile ahead; two super
    // tiles per trip, counted waits.  Requests beyond the wave's last super tile are clamped to it (loaded again, never used).
    // (the buffer requests address 128 columns from one base with 32-bit offsets: fronts of more than 2^21 rows take the general tiles)
    const bool buf_ok = ld * (long long)(QP * STM_NB * sizeof(double)) < (1LL << 31) - (1 << 20);
    const int ns = buf_ok ? (t_hi - t_lo) >> 1 : 0;     // interior super tiles of the workgroup
    int sdone = 0;                                      // ... of this wave that the pipeline took
    if (ns > wid) {
        const int nmine = __builtin_amdgcn_readfirstlane((ns - 1 - wid) / NW + 1), ntrip = nmine >> 1;
        if (ntrip > 0) {
            // Buffer addressing: one resource descriptor per operand (base: this wave's first interior row of C's / V's first column),
            // ONE 32-bit lane offset for every request, the column and tile offsets scalar -- with 64-bit lane addresses the 40 address
            // pairs of a super tile do not fit beside the V halves and the C images (spills inside the loop; scratch reloads share the
            // memory counter).  Everything that feeds a scalar offset is made uniform for the compiler (readfirstlane), see DESIGN.md.
            const int widu = __builtin_amdgcn_readfirstlane(wid);
            const long long r0u = rbeg + 16 * t_lo + 32 * widu;
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(Cg + r0u), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rv =
                __builtin_amdgcn_make_buffer_rsrc((void *)const_cast<double *>(Fb + (long long)G.k1[0] * ld + r0u), 0, 0x7fffffff, 0x00020000);
            const int voff = (int)((2 * l15 + (long long)l4 * ld) * 8);
            const int ld4b = (int)(4 * ld * 8);                  // bytes between columns c and c + 4 (ld <= 2^26)
            struct TC { d2v c0[4], c1[4]; };
            struct TVh { d2v v[QP * STM_NB / 8]; };
            auto load_c = [&](TC &t, int off) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    t.c0[r] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rc, voff, off * 8 + r * ld4b, 0));
                    t.c1[r] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rc, voff, off * 8 + (4 + r) * ld4b, 0));
                }
            };
            auto load_vh = [&](TVh &t, int off, int h) {
#pragma unroll
                for (int kk = 0; kk < QP * STM_NB / 8; kk++)
                    t.v[kk] = __builtin_bit_cast(d2v, __builtin_amdgcn_raw_buffer_load_b128(rv, voff, off * 8 + (h * (QP * STM_NB / 8) + kk) * ld4b, 0));
            };
            TC cb2[2];
            TVh va, vb;
            const int step = 32 * NW, offlast = step * (2 * ntrip - 1);
            load_c(cb2[0], 0);                                   // (in the order of a step -- the scheduler would move C behind V --:
            __builtin_amdgcn_sched_barrier(0);                   //  the counted waits of the loop's first pass are then those of
            load_vh(va, 0, 0);                                   //  every other; the loop header takes the weaker of both)
            __builtin_amdgcn_sched_barrier(0);
            load_vh(vb, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            for (int trip = 0; trip < ntrip; trip++) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int off = step * (2 * trip + q), offn = min(off + step, offlast);
                    load_c(cb2[q ^ 1], offn);
                    __builtin_amdgcn_sched_barrier(0);
                    const TC &tc = cb2[q];
                    d4 a0e = {tc.c0[0].x, tc.c0[1].x, tc.c0[2].x, tc.c0[3].x}, a0o = {tc.c0[0].y, tc.c0[1].y, tc.c0[2].y, tc.c0[3].y};
                    d4 a1e = {tc.c1[0].x, tc.c1[1].x, tc.c1[2].x, tc.c1[3].x}, a1o = {tc.c1[0].y, tc.c1[1].y, tc.c1[2].y, tc.c1[3].y};
#pragma unroll
                    for (int kk = 0; kk < QP * STM_NB / 8; kk++) {
                        const double y0 = y0p[4 * YSQ * kk], y1 = y1p[4 * YSQ * kk];
                        a0e = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, va.v[kk].x, a0e, 0, 0, 0);
                        a1e = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, va.v[kk].x, a1e, 0, 0, 0);
                        a0o = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, va.v[kk].y, a0o, 0, 0, 0);
                        a1o = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, va.v[kk].y, a1o, 0, 0, 0);
                        if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);          // (keeps the -Y operands of at most four steps in registers)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_vh(va, offn, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < QP * STM_NB / 8; kk++) {
                        const double y0 = y0p[4 * YSQ * (QP * STM_NB / 8 + kk)], y1 = y1p[4 * YSQ * (QP * STM_NB / 8 + kk)];
                        a0e = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, vb.v[kk].x, a0e, 0, 0, 0);
                        a1e = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, vb.v[kk].x, a1e, 0, 0, 0);
                        a0o = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, vb.v[kk].y, a0o, 0, 0, 0);
                        a1o = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, vb.v[kk].y, a1o, 0, 0, 0);
                        if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_vh(vb, offn, 1);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double e0 = a0e[r], o0 = a0o[r], e1 = a1e[r], o1 = a1o[r];       // (scalars first: DESIGN.md, compiler findings)
                        const d2v t0 = {e0, o0}, t1 = {e1, o1};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t0), rc, voff, off * 8 + r * ld4b, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t1), rc, voff, off * 8 + (4 + r) * ld4b, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sdone = 2 * ntrip;
        }
        // this wave's super tiles that the pipeline did not take (at most one): as two tiles of the general form
        for (int q = sdone; q < nmine; q++) {
            general_tile(t_lo + 2 * (wid + q * NW));
            general_tile(t_lo + 2 * (wid + q * NW) + 1);
        }
    }
    for (int tix = t_lo + 2 * ns + wid; tix < ntile; tix += NW) general_tile(tix);
}

// T of the LAST panel of a front whose panel kernel left it pending (PanelDesc::t_deferred == 2: the Gram-based panel never
// builds T, and no trailing update follows the last panel).  Only the Q-apply on the resident factors reads it (DevCtx::Tall).
__device__ void dev_tlast(const DevCtx &c, int f, const FrontSym &s, FrontNum *num, double *scratch)
{
    __shared__ PanelShared ps;
    const int p = s.npanels - 1;
    if (p < 0) return;
    const PanelDesc *pd = &num->pd[STM_PDI(p)];
    if (pd->t_deferred != 2 || pd->pnb <= 0) return;
    const int tid = threadIdx.x;
    if (tid < STM_NB) {
        const int d = (tid < pd->pnb) ? pd->pdiag[tid] : STM_BIGROW;
        ps.diag[tid] = d;
        ps.tau[tid] = (d != STM_BIGROW) ? c.Tau[s.rp + pd->pk1 + tid] : 0.0;
    }
    __syncthreads();
    double *Tout = c.Tws + (long long)STM_TSLOT(c.tslot[f], p) * STM_NB * STM_NB;
    dev_gram_T<NT>(c.Farena + s.foff + (long long)pd->pk1 * s.ld, s.ld, pd->pg1, pd->pt, pd->pnb, ps.diag, ps.tau, ps.G, ps.T, Tout,
                   scratch);
    if (c.Tall) {
        double *Tkeep = c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB;
        for (int e = tid; e < STM_NB * STM_NB; e += NT) {
            const int a = e % STM_NB, b = e / STM_NB;
            Tkeep[e] = (a < pd->pnb && b < pd->pnb && a <= b) ? ps.T[a][b] : 0.0;
        }
    }
}

__global__ __launch_bounds__(NT) void k_cpack(DevCtx c, const int *__restrict__ flist,
                                              const int *__restrict__ nparts_list, int maxparts)
{
    extern __shared__ double dyn_lds[];
    const int fi = blockIdx.y;
    const int f = flist[fi];
    const FrontSym s = c.fs[f];
    if ((int)blockIdx.x == maxparts) {                         // the extra workgroup of every front: pending T of its last panel
        dev_tlast(c, f, s, &c.fnum[f], dyn_lds);
        return;
    }
    const int nparts = nparts_list[fi];
    if ((int)blockIdx.x >= nparts) return;
    dev_cpack(c, s, &c.fnum[f], blockIdx.x, nparts);
}

// ------------------------------------------------------------------------------------------------
// qr_rhpack, split in three: per-column lengths + offsets, offsets of the blocks (Post order = the
// reference's single shrunk stack), coalesced copy.  Layout: SURVEY.md A.6.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_rh_count(DevCtx c, const int *__restrict__ flist)
{
    __shared__ int s_scan[NW];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    FrontNum *num = &c.fnum[f];
    const int tid = threadIdx.x;
    const int fm = num->fm, n = s.fn, fp = s.fp;
    const int *St = c.Stair + s.rp;
    long long *off = c.Rhoff + s.rp;                   // (64-bit: the block of a 52 000 x 50 000 front has 2.2e9 entries)
    if (fm <= 0 || n <= 0) {
        for (int k = tid; k < n; k += NT) off[k] = 0;
        if (tid == 0) num->rsize = 0;
        return;
    }
    // pass 1: rm(k) = live pivots among columns 0..k (stored temporarily in off[])
    long long carry = 0;
    for (int base = 0; base < fp; base += NT) {
        const int k = base + tid;
        const int live = (k < fp && St[k] != 0) ? 1 : 0;
        int tot;
        const int incl = block_incl_scan(live, s_scan, &tot);
        if (k < fp) off[k] = carry + incl;
        carry += tot;
    }
    __syncthreads();
    const int rm = (int)carry;
    // pass 2: column lengths -> exclusive offsets
    carry = 0;
    for (int base = 0; base < n; base += NT) {
        const int k = base + tid;
        int len = 0;
        if (k < fp) {
            const int t = St[k];
            len = (t == 0) ? (int)off[k] : t;  // dead: rm so far (off[k] excludes k itself since live=0)
        } else if (k < n) {
            const int h = min(rm + (k - fp) + 1, fm);
            len = rm + max(St[k] - h, 0);
        }
        __syncthreads();
        int tot;
        const int incl = block_incl_scan(len, s_scan, &tot);
        if (k < n) off[k] = carry + incl - len;
        carry += tot;
    }
    if (tid == 0) {
        num->rsize = carry;
        if (c.rh_top) {
            // slab recycling (stmmqr_host.cpp, "timeline allocator"): the block gets its place in the R+H arena NOW, by a device-side
            // bump pointer -- its size depends on the numerical rank, the host never learns it before the end.  The arena holds
            // QRsym->maxstack doubles, the reference's own bound for all of R+H (SparseQR_analyze.c:1061-1161); should a block
            // not fit all the same, the overflow word is raised and nothing is copied (the host repeats without recycling).
            const long long at = (long long)atomicAdd((unsigned long long *)c.rh_top, (unsigned long long)carry);
            if (at + carry > c.rh_cap) { c.Rboff[f] = -1; atomicExch((int *)(c.rh_top + 1), 1); }
            else c.Rboff[f] = at;
        }
    }
}
// Slab recycling: the fronts that start at a step get their (reused) slabs zeroed before the assembly scatters into them
// (k_assemble writes every entry of S and of the children's contribution blocks exactly once and relies on zeros elsewhere;
// without recycling ONE memset of the whole arena does this).  grid (parts, fronts); 16-byte stores.
__global__ __launch_bounds__(256) void k_zero_slabs(DevCtx c, const int *__restrict__ flist)
{
    const int f = flist[blockIdx.y];
    const FrontSym s = c.fs[f];
    const long long n2 = ((long long)s.ld * s.fn) >> 1;             // ld is even: whole double2's
    double2 *F2 = reinterpret_cast<double2 *>(c.Farena + s.foff);
    const double2 z = {0.0, 0.0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) F2[i] = z;
}

// Slab recycling, resident-factor operations (Q-apply, solves): a front whose slab was given to another front after its packed
// R+H block had been staged is put back into front form in a scratch slab (zeros + the inverse of k_rh_copy); the kernels of
// SURVEY 8 (f1) then read it exactly as they read a front that was never packed.  `cs`: the FrontSym array of the scratch layout
// (foff = where this front lives in the scratch of its level).  grid (parts, fronts); phase 0 zero-fill, phase 1 scatter.
__global__ __launch_bounds__(NT) void k_rh_unpack(DevCtx c, const FrontSym *__restrict__ cs, const int *__restrict__ flist,
                                                  const char *__restrict__ kept, const double *__restrict__ RH, double *__restrict__ scratch,
                                                  int phase)
{
    const int f = flist[blockIdx.y];
    if (kept[f]) return;                                       // (still in front form in its own slab)
    const FrontSym s = cs[f];
    const FrontNum *num = &c.fnum[f];
    const int fm = num->fm, n = s.fn, fp = s.fp, rm = num->rank;
    double *F = scratch + s.foff;
    const long long ld = s.ld;
    if (phase == 0) {
        // zeros where the kernels may look: the front's ACTUAL rows (the slab's leading dimension is the symbolic bound under rank
        // detection, often twice as many), column by column
        const int rows2 = (min(fm, (int)ld) + 1) >> 1;
        const double2 z = {0.0, 0.0};
        const long long tot = (long long)rows2 * n;
        for (long long e = (long long)blockIdx.x * NT + threadIdx.x; e < tot; e += (long long)gridDim.x * NT) {
            const int k = (int)(e / rows2), i2 = (int)(e - (long long)k * rows2);
            reinterpret_cast<double2 *>(F + k * ld)[i2] = z;
        }
        return;
    }
    if (fm <= 0 || n <= 0 || c.Rboff[f] < 0) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int *St = c.Stair + s.rp;
    const long long *off = c.Rhoff + s.rp;
    const double *R = RH + c.Rboff[f];
    for (int k = blockIdx.x * NW + wid; k < n; k += gridDim.x * NW) {
        double *Fk = F + k * ld;
        const double *Rk = R + off[k];
        if (k < fp) {
            const int len = (int)(((k + 1 < n) ? off[k + 1] : num->rsize) - off[k]);
            for (int i = lane; i < len; i += 64) Fk[i] = Rk[i];
        } else {
            const int h = min(rm + (k - fp) + 1, fm);
            const int t = St[k];
            for (int i = lane; i < rm; i += 64) Fk[i] = Rk[i];
            for (int i = lane; i < t - h; i += 64) Fk[h + i] = Rk[rm + i];
        }
    }
}

// Slab recycling, download: the packed blocks sit in the arena in the order the fronts finished; the host wants the reference's
// layout (Post order, Rblock offsets = exclusive sums of the block sizes: `fin`).  One launch copies the part of every block that
// falls into the window [w0, w1) of that final layout into `out` (a bounce buffer the host then reads); fronts that kept their
// slab (`kept`: never packed on the device) are packed on the fly, column by column, with the same clipping.
__global__ __launch_bounds__(NT) void k_rh_window(DevCtx c, const int *__restrict__ flist, const long long *__restrict__ fin,
                                                  const char *__restrict__ kept, const double *__restrict__ RH, long long w0, long long w1,
                                                  double *__restrict__ out)
{
    const int f = flist[blockIdx.y];
    const FrontSym s = c.fs[f];
    const FrontNum *num = &c.fnum[f];
    const long long b0 = fin[f], b1 = b0 + num->rsize;
    if (b1 <= w0 || b0 >= w1 || num->rsize <= 0) return;
    if (!kept[f]) {
        if (c.Rboff[f] < 0) return;
        const double *src = RH + c.Rboff[f];
        const long long a = max(w0, b0), b = min(w1, b1);
        for (long long i = a + (long long)blockIdx.x * NT + threadIdx.x; i < b; i += (long long)gridDim.x * NT) out[i - w0] = src[i - b0];
        return;
    }
    const int fm = num->fm, n = s.fn, fp = s.fp, rm = num->rank;
    if (fm <= 0 || n <= 0) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long ld = s.ld;
    const double *F = c.Farena + s.foff;
    const int *St = c.Stair + s.rp;
    const long long *off = c.Rhoff + s.rp;
    auto copy = [&](long long dst0, const double *src, int len) {        // dst0: position in the final layout
        if (dst0 + len <= w0 || dst0 >= w1) return;
        const int i0 = (int)max(0LL, w0 - dst0), i1 = (int)min((long long)len, w1 - dst0);
        for (int i = i0 + lane; i < i1; i += 64) out[dst0 + i - w0] = src[i];
    };
    for (int k = blockIdx.x * NW + wid; k < n; k += gridDim.x * NW) {
        const double *Fk = F + k * ld;
        const long long d = b0 + off[k];
        if (k < fp) {
            const int len = (int)(((k + 1 < n) ? off[k + 1] : num->rsize) - off[k]);
            copy(d, Fk, len);
        } else {
            const int h = min(rm + (k - fp) + 1, fm);
            copy(d, Fk, rm);
            copy(d + rm, Fk + h, St[k] - h);
        }
    }
}

// single workgroup: Rboff[f] = offset of front f's block in Post order; total in *rh_total
__global__ __launch_bounds__(NT) void k_rh_scan(DevCtx c, const int *__restrict__ post, int nf, long long *rh_total, long long *outoff)
{
    __shared__ long long s_part[NT];
    const int tid = threadIdx.x;
    const int per = (nf + NT - 1) / NT;
    const int a = tid * per, b = min(nf, a + per);
    long long sum = 0;
    for (int q = a; q < b; q++) sum += c.fnum[post[q]].rsize;
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        long long run = 0;
        for (int t = 0; t < NT; t++) { const long long v = s_part[t]; s_part[t] = run; run += v; }
        *rh_total = run;
    }
    __syncthreads();
    long long run = s_part[tid];
    for (int q = a; q < b; q++) {
        const int f = post[q];
        outoff[f] = run;
        run += c.fnum[f].rsize;
    }
}

__global__ __launch_bounds__(NT) void k_rh_copy(DevCtx c, const int *__restrict__ flist,
                                                const int *__restrict__ nparts_list, double *__restrict__ RH)
{
    const int fi = blockIdx.y;
    const int nparts = nparts_list[fi];
    if ((int)blockIdx.x >= nparts) return;
    const int f = flist[fi];
    const FrontSym s = c.fs[f];
    const FrontNum *num = &c.fnum[f];
    const int fm = num->fm, n = s.fn, fp = s.fp, rm = num->rank;
    if (fm <= 0 || n <= 0) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long ld = s.ld;
    const double *F = c.Farena + s.foff;
    const int *St = c.Stair + s.rp;
    const long long *off = c.Rhoff + s.rp;
    if (c.Rboff[f] < 0) return;                              // (arena overflow: flagged by k_rh_count, the host repeats)
    double *R = RH + c.Rboff[f];
    // one wave per column; eight loads of a lane in flight before their stores (a plain copy loop waits for every load:
    // 0.85 TB/s on the default workload's 1.2 GB of factors)
    auto copy = [&](double *dst, const double *src, int len) {
        int i = lane;
        for (; i + 7 * 64 < len; i += 8 * 64) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[i + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; u++) dst[i + 64 * u] = v[u];
        }
        for (; i < len; i += 64) dst[i] = src[i];
    };
    for (int k = blockIdx.x * NW + wid; k < n; k += nparts * NW) {
        const double *Fk = F + k * ld;
        double *Rk = R + off[k];
        if (k < fp) {
            const int len = (int)(((k + 1 < n) ? off[k + 1] : num->rsize) - off[k]);
            copy(Rk, Fk, len);
        } else {
            const int h = min(rm + (k - fp) + 1, fm);
            const int t = St[k];
            copy(Rk, Fk, rm);
            copy(Rk + rm, Fk + h, t - h);
        }
    }
}

// ================================================================================================
// Q-apply and triangular solve on the factors that are still resident in HBM
// (reference: qr_private_Happly / QR_qmult, STMMQR/src/qr/SparseQR.c:1455-1790; qr_rsolve :2218-2517).
// The Householder vectors are read in place from the front arena (unit diagonal of the q-th live reflector of a front
// at front row q, entries below it up to the column's HStair), rows of a front are global rows through Hii.
// Work vector W: one entry per row of A, indexed by the row id of S = A(P,:)  (Hii holds exactly these ids).
// ================================================================================================
#ifndef QA_NT
#define QA_NT 512
#endif
#define QA_NW (QA_NT / 64)
// inclusive scan of one int per thread over NWV waves; *total = sum.  s_scan: NWV ints.
template <int NWV>
__device__ __forceinline__ int qa_incl_scan(int v, int *s_scan, int *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) s_scan[wid] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
        int sw = s_scan[w];
        if (w < wid) base += sw;
        tot += sw;
    }
    *total = tot;
    return x + base;
}
// live-reflector numbering of one front: dq[k] = number of live reflectors before column k, or -1 if column k has none
// (dead pivot column, or the rows ran out).  Mirrors the enumeration of the packed format (qr_rhpack :1691-1784).
__device__ void qa_number_reflectors(const FrontSym &s, int fm, const int *St, const double *Tau, int *dq, int *s_scan)
{
    const int tid = threadIdx.x;
    const int per = (s.fn + QA_NT - 1) / QA_NT;
    const int k0 = tid * per, k1 = min(s.fn, k0 + per);
    int cnt = 0;
    for (int k = k0; k < k1; k++) cnt += (k >= s.fp || St[k] != 0);
    int total;
    const int incl = qa_incl_scan<QA_NW>(cnt, s_scan, &total);
    int d = incl - cnt;
    for (int k = k0; k < k1; k++) {
        const bool live = (k >= s.fp || St[k] != 0);
        dq[k] = (live && d < fm && Tau[k] != 0.0) ? d : -1;     // (tau == 0: H = I, nothing to apply)
        d += live;
    }
    __syncthreads();
}

// method 0: x <- H_last ... H_1 x = Q'x (fronts leaves -> root, reflectors ascending); method 1: x <- Q x (reverse)
__global__ __launch_bounds__(QA_NT) void k_qapply(DevCtx c, const int *__restrict__ flist, int method, double *W, int *err)
{
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[QA_NW];
    __shared__ double s_red[QA_NW];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    const int tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    const double *Tau = c.Tau + s.rp;
    const int *Hi = c.Hii + s.hip;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    double *xs = dyn_lds;                               // [fm]
    int *dq = (int *)(xs + ((fm + 1) & ~1));            // [fn]
    for (int i = tid; i < fm; i += QA_NT) xs[i] = W[Hi[i]];
    qa_number_reflectors(s, fm, St, Tau, dq, s_scan);
    // one reflector after the other: v'x by a workgroup reduction, then the rank-1 update of the LDS-resident x
    // (requesting the next column ahead of the reduction was measured and did not pay: 21.9 -> 26 ms on the stand-in)
    const int kbeg = method ? s.fn - 1 : 0, kend = method ? -1 : s.fn, kinc = method ? -1 : 1;
    for (int k = kbeg; k != kend; k += kinc) {
        const int d = dq[k];
        if (d < 0) continue;
        const double tau = Tau[k];
        const int h = min(d + 1, fm), t = St[k];
        const double *v = F + (long long)k * ld;
        double part = 0;
        for (int i = h + tid; i < t; i += QA_NT) part += v[i] * xs[i];
        double sdot = block_sum<QA_NT>(part, s_red);
        sdot = (sdot + xs[d]) * tau;
        __syncthreads();                                // everyone has read xs[d]
        for (int i = h + tid; i < t; i += QA_NT) xs[i] -= sdot * v[i];
        if (tid == 0) xs[d] -= sdot;
        __syncthreads();
    }
    for (int i = tid; i < fm; i += QA_NT) W[Hi[i]] = xs[i];
    (void)err;
}

// Blocked form of k_qapply with the T factors the factorization kept (DevCtx::Tall): per panel of <= 32 reflectors
//     w = V'x  (one sweep over the panel rows, 32 dot products per thread, ONE workgroup reduction of 32 values),
//     y = T'w  (Q'x)  or  T w  (Q x),      x -= V y  (second sweep).
// V is read in place with the unit-diagonal / staircase mask (reflector j of the panel: diagonal row dq[k], entries up
// to HStair[k]); identity and dead columns have zero rows/columns in T and are masked out of V.
__global__ __launch_bounds__(QA_NT) void k_qapply_t(DevCtx c, const int *__restrict__ flist, int method, double *W, RhsBatch B)
{
    W += (long long)blockIdx.y * B.w;                      // (right-hand side blockIdx.y of the batch)
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[QA_NW];
    __shared__ int s_d[STM_NB], s_t[STM_NB];
    __shared__ double s_part[QA_NW][STM_NB], s_w[STM_NB], s_y[STM_NB], s_T[STM_NB][STM_NB + 1];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0 || s.qbig) return;         // (qbig: k_qbig_* below)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *St = c.Stair + s.rp;
    const double *Tau = c.Tau + s.rp;
    const int *Hi = c.Hii + s.hip;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    double *xs = dyn_lds;                               // [fm]
    int *dq = (int *)(xs + ((fm + 1) & ~1));            // [fn]
    for (int i = tid; i < fm; i += QA_NT) xs[i] = W[Hi[i]];
    qa_number_reflectors(s, fm, St, Tau, dq, s_scan);
    const int pbeg = method ? s.npanels - 1 : 0, pend = method ? -1 : s.npanels, pinc = method ? -1 : 1;
    for (int p = pbeg; p != pend; p += pinc) {
        const int k1 = p * STM_NB, nbp = min(STM_NB, s.fn - k1);
        if (tid < STM_NB) {
            const int d = (tid < nbp) ? dq[k1 + tid] : -1;
            s_d[tid] = (d >= 0) ? d : STM_BIGROW;
            s_t[tid] = (d >= 0) ? St[k1 + tid] : 0;
        }
        __syncthreads();
        int r0 = STM_BIGROW, r1 = 0;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) { r0 = min(r0, s_d[j]); r1 = max(r1, max(s_t[j], (s_d[j] < STM_BIGROW) ? s_d[j] + 1 : 0)); }
        if (r0 >= STM_BIGROW) { __syncthreads(); continue; }   // no live reflector in this panel (uniform)
        const double *Vp = F + (long long)k1 * ld;
        // T of this panel: requested now (coalesced), parked in LDS after the first sweep
        double treg[(STM_NB * STM_NB + QA_NT - 1) / QA_NT];
        {
            const double *T = c.Tall + (long long)(s.tpan + p) * STM_NB * STM_NB;
#pragma unroll
            for (int q = 0; q < (STM_NB * STM_NB + QA_NT - 1) / QA_NT; q++) treg[q] = T[min(tid + QA_NT * q, STM_NB * STM_NB - 1)];
        }
        // ---- w = V'x ----
        double acc[STM_NB];
#pragma unroll
        for (int j = 0; j < STM_NB; j++) acc[j] = 0;
        for (int i = r0 + tid; i < r1; i += QA_NT) {
            const double xi = xs[i];
#pragma unroll
            for (int j = 0; j < STM_NB; j++) {
                const double val = Vp[i + (long long)min(j, nbp - 1) * ld];        // unconditional, masked below
                const double v = (i > s_d[j] && i < s_t[j]) ? val : ((i == s_d[j]) ? 1.0 : 0.0);
                acc[j] += v * xi;
            }
        }
        {
            double part[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int x = 0; x < 8; x++) part[x] = acc[8 * q + x];
                const double rw = wave_reduce8(part);                             // lane l: total of value red8_idx(l)
                if (lane < 8) s_part[wid][8 * q + red8_idx(lane)] = rw;
            }
        }
#pragma unroll
        for (int q = 0; q < (STM_NB * STM_NB + QA_NT - 1) / QA_NT; q++) {
            const int e = tid + QA_NT * q;
            if (e < STM_NB * STM_NB) s_T[e % STM_NB][e / STM_NB] = treg[q];                // s_T[row][col], padded rows: no bank conflicts below
        }
        __syncthreads();
        if (tid < STM_NB) {
            double v = 0;
#pragma unroll
            for (int w = 0; w < QA_NW; w++) v += s_part[w][tid];
            s_w[tid] = v;
        }
        __syncthreads();
        // ---- y = T'w (Q'x) or T w (Q x); T upper triangular ----
        if (tid < STM_NB) {
            double y = 0;
            if (method == 0) { for (int q = 0; q <= tid; q++) y += s_T[q][tid] * s_w[q]; }
            else { for (int q = tid; q < STM_NB; q++) y += s_T[tid][q] * s_w[q]; }
            s_y[tid] = y;
        }
        __syncthreads();
        // ---- x -= V y ----
        for (int i = r0 + tid; i < r1; i += QA_NT) {
            double a = xs[i];
#pragma unroll
            for (int j = 0; j < STM_NB; j++) {
                const double val = Vp[i + (long long)min(j, nbp - 1) * ld];
                const double v = (i > s_d[j] && i < s_t[j]) ? val : ((i == s_d[j]) ? 1.0 : 0.0);
                a -= v * s_y[j];
            }
            xs[i] = a;
        }
        __syncthreads();
    }
    for (int i = tid; i < fm; i += QA_NT) W[Hi[i]] = xs[i];
}

// One front of the back substitution R x = y (fronts root -> leaves; reference qr_rsolve, SparseQR.c:2218-2470):
// y = the first rm rows of the front's slice of W (rm = live pivot columns), x of the non-pivotal columns comes from
// the ancestors, a dead pivot column gets x = 0 (basic solution), the live pivot columns form an rm x rm upper triangle
// whose row q is the q-th live column.
// ------------------------------------------------------------------------------------------------
// Q-apply for the large fronts (FrontSym::qbig): one workgroup streams a front's V at the fill rate of ONE CU
// (~45 GB/s: 10 ms for a 7818 x 7818 front), so the rows of such a front are split over workgroups (QB_ROWS rows
// each, one row per thread) and the panels become launches: launch k applies panel p_prev (x -= V y with y = T'w or
// T w, w = the slab partials of the previous launch summed in slab order) and forms the slab partials of
// w = V'x for panel p_next.  x lives in a device buffer Xf (gathered from / scattered to the work vector by
// k_qbig_prep / k_qbig_finish), the reflector numbering in Dq.
// ------------------------------------------------------------------------------------------------
#define QB_ROWS STM_QB_ROWS   // rows (= threads) of a step workgroup (128-row workgroups were measured: slower)
// All split fronts of a tree level advance together: blockIdx.y = index into the level's descriptor list
// (QbDesc: front, offsets of its slices of Xf / Dq / Wq, number of row slabs), launch k handles the k-th panel of each.
__global__ __launch_bounds__(QA_NT) void k_qbig_prep(DevCtx c, const QbDesc *__restrict__ qd, const double *W, double *Xf0, int *Dq0, RhsBatch B)
{
    W += (long long)blockIdx.y * B.w; Xf0 += (long long)blockIdx.y * B.xf;      // (the reflector numbering Dq0 is the same for every right-hand side)
    __shared__ int s_scan[QA_NW];
    const QbDesc d = qd[blockIdx.x];
    const FrontSym s = c.fs[d.f];
    const int fm = c.fnum[d.f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    const int *Hi = c.Hii + s.hip;
    double *Xf = Xf0 + d.xoff;
    for (int i = threadIdx.x; i < fm; i += QA_NT) Xf[i] = W[Hi[i]];
    qa_number_reflectors(s, fm, c.Stair + s.rp, c.Tau + s.rp, Dq0 + d.dqoff, s_scan);
}
__global__ __launch_bounds__(256) void k_qbig_finish(DevCtx c, const QbDesc *__restrict__ qd, double *W, const double *Xf0, RhsBatch B)
{
    W += (long long)blockIdx.z * B.w; Xf0 += (long long)blockIdx.z * B.xf;
    const QbDesc d = qd[blockIdx.y];
    const FrontSym s = c.fs[d.f];
    const int fm = c.fnum[d.f].fm;
    const int *Hi = c.Hii + s.hip;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < fm) W[Hi[i]] = Xf0[d.xoff + i];
}
__global__ __launch_bounds__(QB_ROWS) void k_qbig_step(DevCtx c, const QbDesc *__restrict__ qd, int k, int method, double *Xf0,
                                                     const int *Dq0, double *Wq0, RhsBatch B)
{
    Xf0 += (long long)blockIdx.z * B.xf; Wq0 += (long long)blockIdx.z * B.wq;
    __shared__ int s_d[2][STM_NB], s_t[2][STM_NB];
    __shared__ double s_part[QB_ROWS / 64][STM_NB], s_w[STM_NB], s_y[STM_NB], s_T[STM_NB][STM_NB + 1];
    const QbDesc qdd = qd[blockIdx.y];
    const int f = qdd.f, nslab = qdd.nslab;
    if ((int)blockIdx.x >= nslab) return;
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    // launch k: apply the (k-1)-th panel of the order, form the partials of the k-th (Q'x: ascending, Q x: descending)
    const int np = s.npanels;
    if (k > np) return;
    const int pp[2] = {(k >= 1) ? (method ? np - k : k - 1) : -1, (k < np) ? (method ? np - 1 - k : k) : -1};
    double *Xf = Xf0 + qdd.xoff;
    const int *Dq = Dq0 + qdd.dqoff;
    double *Wq = Wq0 + qdd.wqoff;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *St = c.Stair + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int i = blockIdx.x * QB_ROWS + tid, ic = min(i, fm - 1);
    // round trip 1: the reflector descriptions of both panels
    if (tid < 2 * STM_NB) {
        const int w = tid >> 5, j = tid & 31, p = pp[w];
        int d = -1, t = 0;
        if (p >= 0 && p * STM_NB + j < s.fn) { d = Dq[p * STM_NB + j]; t = St[p * STM_NB + j]; }
        s_d[w][j] = (d >= 0) ? d : STM_BIGROW;
        s_t[w][j] = (d >= 0) ? t : 0;
    }
    double x = (i < fm) ? Xf[i] : 0.0;
    __syncthreads();
    bool on[2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        int r0 = STM_BIGROW, r1 = 0;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) {
            r0 = min(r0, s_d[w][j]);
            r1 = max(r1, max(s_t[w][j], (s_d[w][j] < STM_BIGROW) ? s_d[w][j] + 1 : 0));
        }
        on[w] = (pp[w] >= 0 && r0 < STM_BIGROW && (int)blockIdx.x * QB_ROWS < r1 && ((int)blockIdx.x + 1) * QB_ROWS > r0);   // (uniform)
    }
    // round trip 2: everything both phases read, requested together (the launch is a chain of memory round trips)
    double v0[STM_NB] = {}, v1[STM_NB] = {}, treg[STM_NB * STM_NB / QB_ROWS], wsum = 0;
    if (on[0]) {
        const int nbp = min(STM_NB, s.fn - pp[0] * STM_NB);
        const double *Vp = F + (long long)(pp[0] * STM_NB) * ld;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) v0[j] = Vp[ic + (long long)min(j, nbp - 1) * ld];      // unconditional, masked below
        const double *T = c.Tall + (long long)(s.tpan + pp[0]) * STM_NB * STM_NB;
#pragma unroll
        for (int q = 0; q < STM_NB * STM_NB / QB_ROWS; q++) treg[q] = T[tid + QB_ROWS * q];
        if (tid < STM_NB) {
            const double *wp = Wq + (long long)(pp[0] & 1) * nslab * STM_NB;
            wsum += stm_ordered_sum<false>(wp + tid, STM_NB, nslab);                            // fixed order: deterministic
        }
    }
    if (on[1]) {
        const int nbp = min(STM_NB, s.fn - pp[1] * STM_NB);
        const double *Vp = F + (long long)(pp[1] * STM_NB) * ld;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) v1[j] = Vp[ic + (long long)min(j, nbp - 1) * ld];
    }
    if (on[0]) {
#pragma unroll
        for (int q = 0; q < STM_NB * STM_NB / QB_ROWS; q++) s_T[(tid + QB_ROWS * q) % STM_NB][(tid + QB_ROWS * q) / STM_NB] = treg[q];
        if (tid < STM_NB) s_w[tid] = wsum;
        __syncthreads();
        if (tid < STM_NB) {
            double y = 0;
            if (method == 0) { for (int q = 0; q <= tid; q++) y += s_T[q][tid] * s_w[q]; }
            else { for (int q = tid; q < STM_NB; q++) y += s_T[tid][q] * s_w[q]; }
            s_y[tid] = y;
        }
        __syncthreads();
        double a = x;
#pragma unroll
        for (int j = 0; j < STM_NB; j++) {
            const double v = (i > s_d[0][j] && i < s_t[0][j]) ? v0[j] : ((i == s_d[0][j]) ? 1.0 : 0.0);
            a -= v * s_y[j];
        }
        if (i < fm && a != x) Xf[i] = a;
        x = (i < fm) ? a : 0.0;
    }
    if (pp[1] >= 0) {
        double acc[STM_NB];
#pragma unroll
        for (int j = 0; j < STM_NB; j++) {
            const double v = (i < fm && i > s_d[1][j] && i < s_t[1][j]) ? v1[j] : ((i == s_d[1][j]) ? 1.0 : 0.0);
            acc[j] = on[1] ? v * x : 0.0;
        }
        {
            double part[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
#pragma unroll
                for (int xx = 0; xx < 8; xx++) part[xx] = acc[8 * q + xx];
                const double rw = wave_reduce8(part);                             // lane l: total of value red8_idx(l)
                if (lane < 8) s_part[wid][8 * q + red8_idx(lane)] = rw;
            }
        }
        __syncthreads();
        if (tid < STM_NB) {
            double v = 0;
#pragma unroll
            for (int w = 0; w < QB_ROWS / 64; w++) v += s_part[w][tid];
            Wq[((long long)(pp[1] & 1) * nslab + blockIdx.x) * STM_NB + tid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Grouped split Q-apply (round 4): FOUR panels (128 reflectors) of a large front per launch instead of one.  The launches of
// k_qbig_step are a chain (launch k needs the x of launch k-1) of ~10 us each whatever they do -- two dependent memory round trips --,
// 1473 of them per Q'b on the default workload.  The product of the four block reflectors of a group is ONE block reflector,
//     H_0 H_1 H_2 H_3 = I - V T4 V',    T4 = [T_0 X_01 X_02 X_03; 0 T_1 X_12 X_13; 0 0 T_2 X_23; 0 0 0 T_3]   (128 x 128),
// with the blocks X_ij of the compact WY recurrence ( [X_0j; ..; X_{j-1,j}] = -T4(0:j, 0:j) [G_0j; ..; G_{j-1,j}] T_j, G_ij = V_i'V_j ).
// T4 of every group of every split front is built ONCE per factorization, at the first Q-apply (k_qt4_number, k_qt4_build: the Gram
// matrix of the group's 128 columns by MFMA, then sixteen 32 x 32 block products), and kept in both layouts (T4 for Q x, its transpose
// for Q'x: the matrix-vector product of a launch reads it with the lanes along the output index).
// ------------------------------------------------------------------------------------------------
#define QG 4
#define QGN (QG * STM_NB)
#define QT4_DOUBLES (2 * QGN * QGN)                  // per group: T4 column-major, then T4 row-major
struct Qt4Item { int f, g; long long off, dqo; };    // T4 of group g of front f at T4all + off; the front's reflector numbering at Dq4 + dqo

__global__ __launch_bounds__(QA_NT) void k_qt4_number(DevCtx c, const int *__restrict__ fl, const long long *__restrict__ dqo, int *Dq4)
{
    __shared__ int s_scan[QA_NW];
    const int f = fl[blockIdx.x];
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    qa_number_reflectors(s, fm, c.Stair + s.rp, c.Tau + s.rp, Dq4 + dqo[blockIdx.x], s_scan);
}

#define QT4_VS 36
__global__ __launch_bounds__(512) void k_qt4_build(DevCtx c, const Qt4Item *__restrict__ items, const int *__restrict__ Dq4, double *T4all)
{
    extern __shared__ double lds[];                   // Gram phase: the chunk image [QGN][QT4_VS]; block phase: five 32 x 33 blocks
    __shared__ int s_d[QGN], s_t[QGN], s_rng[2][8], s_live[QG];
    const Qt4Item it = items[blockIdx.x];
    const int f = it.f, g = it.g;
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    double *T4c = T4all + it.off, *T4r = T4c + QGN * QGN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int k0 = g * QGN;
    if (tid < QGN) {
        const int k = k0 + tid;
        int d = -1, t = 0;
        if (fm > 0 && k < s.fn) { d = Dq4[it.dqo + k]; t = c.Stair[s.rp + k]; }
        s_d[tid] = (d >= 0) ? d : STM_BIGROW;
        s_t[tid] = (d >= 0) ? max(t, d + 1) : 0;
        const int lo = wave_max_int(-s_d[tid]), hi = wave_max_int(s_t[tid]);
        if (lane == 0) { s_rng[0][wid] = -lo; s_rng[1][wid] = hi; }
    }
    __syncthreads();
    if (tid < QG) {
        int lv = 0;
        for (int j = 0; j < STM_NB; j++) lv |= (s_d[STM_NB * tid + j] < STM_BIGROW);
        s_live[tid] = lv;
    }
    const int rmin = min(s_rng[0][0], s_rng[0][1]), rmax = min(fm, max(s_rng[1][0], s_rng[1][1]));
    if (rmin >= STM_BIGROW || rmax <= rmin) {                       // no live reflector in the group: T4 = 0
        for (int e = tid; e < QT4_DOUBLES; e += 512) st_agent(&T4c[e], 0.0);
        return;
    }
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    // ---- G = V'V (128 x 128) over the rows [rmin, rmax): wave w the tile row w ----
    d4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = (d4){0, 0, 0, 0};
    double *Vs = lds;
    const int srow = tid & 31, scg = tid >> 5;
    for (int r0 = rmin; r0 < rmax; r0 += 32) {
        const int i = r0 + srow, ic = min(i, fm - 1);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int col = scg * 8 + q, k = min(k0 + col, s.fn - 1);
            const double val = F[ic + (long long)k * ld];
            const int d = s_d[col], t = s_t[col];
            Vs[col * QT4_VS + srow] = (i == d) ? 1.0 : ((i > d && i < t) ? val : 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const double a = Vs[(16 * wid + l15) * QT4_VS + 4 * kk + l4];
#pragma unroll
            for (int tc = 0; tc < 8; tc++) {
                const double b = Vs[(16 * tc + l15) * QT4_VS + 4 * kk + l4];
                acc[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[tc], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // G to the T4c area (column-major; overwritten block column by block column below), write-through: read back by other waves
#pragma unroll
    for (int tc = 0; tc < 8; tc++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_agent(&T4c[(16 * wid + l4 + 4 * r) + QGN * (16 * tc + l15)], acc[tc][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- the blocks of T4 ----
    double *Ab = lds, *Bb = Ab + 32 * 33, *Yb = Bb + 32 * 33;        // Yb: three blocks
    const int oa = tid & 31, ob = tid >> 5;                          // my two outputs of a block product: (oa, ob), (oa, ob + 16)
    const int np = s.npanels;
    // T of panel 4 g + i (column-major); zero beyond the front's panels and for a panel without a live reflector -- the factorization
    // never wrote T of the panels behind the one where the rows ran out (the per-panel kernels never read it either)
    auto load_T = [&](int i, double *dst) {
        const int p = QG * g + i;
        const bool live = p < np && s_live[i];
        const double *T = c.Tall + (long long)(s.tpan + min(p, np - 1)) * STM_NB * STM_NB;
        for (int e = tid; e < 32 * 32; e += 512) dst[(e & 31) * 33 + (e >> 5)] = live ? T[e] : 0.0;           // dst[row][col]
    };
    auto load_blk = [&](int i, int j, double *dst) {                 // block (i, j) of the T4c area
        for (int e = tid; e < 32 * 32; e += 512) dst[(e & 31) * 33 + (e >> 5)] = ld_agent(&T4c[(32 * i + (e & 31)) + QGN * (32 * j + (e >> 5))]);
    };
    auto mm = [&](const double *A, const double *B, double &o0, double &o1) {      // += A B at my two outputs
#pragma unroll 8
        for (int q = 0; q < 32; q++) {
            const double a = A[oa * 33 + q];
            o0 += a * B[q * 33 + ob];
            o1 += a * B[q * 33 + ob + 16];
        }
    };
    for (int j = 1; j < QG; j++) {
        load_T(j, Bb);
        for (int i = 0; i < j; i++) {                                // Y_i = G_ij T_j
            __syncthreads();
            load_blk(i, j, Ab);
            __syncthreads();
            double y0 = 0, y1 = 0;
            mm(Ab, Bb, y0, y1);
            Yb[i * 32 * 33 + oa * 33 + ob] = y0;
            Yb[i * 32 * 33 + oa * 33 + ob + 16] = y1;
        }
        for (int i = 0; i < j; i++) {                                // X_ij = - sum_{l = i .. j-1} T4_il Y_l
            double x0 = 0, x1 = 0;
            for (int l = i; l < j; l++) {
                __syncthreads();
                if (l == i) load_T(i, Ab);
                else load_blk(i, l, Ab);
                __syncthreads();
                mm(Ab, Yb + l * 32 * 33, x0, x1);
            }
            st_agent(&T4c[(32 * i + oa) + QGN * (32 * j + ob)], -x0);
            st_agent(&T4c[(32 * i + oa) + QGN * (32 * j + ob + 16)], -x1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // diagonal blocks, zeros below them, and the row-major copy
    for (int i = 0; i < QG; i++) {
        __syncthreads();
        load_T(i, Ab);
        __syncthreads();
        for (int e = tid; e < 32 * 32; e += 512) st_agent(&T4c[(32 * i + (e & 31)) + QGN * (32 * i + (e >> 5))], Ab[(e & 31) * 33 + (e >> 5)]);
    }
    for (int e = tid; e < QGN * QGN; e += 512) {
        const int r = e & (QGN - 1), cc = e >> 7;
        if ((r >> 5) > (cc >> 5)) st_agent(&T4c[e], 0.0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = tid; e < QGN * QGN; e += 512) {
        const int cc = e & (QGN - 1), r = e >> 7;                    // T4r[c + 128 r] = T4(r, c)
        T4r[e] = ld_agent(&T4c[r + QGN * cc]);
    }
}

__global__ __launch_bounds__(QB_ROWS) void k_qbig_step4(DevCtx c, const QbDesc *__restrict__ qd, const long long *__restrict__ t4off, int k,
                                                      int method, double *Xf0, const int *Dq0, double *Wq0, const double *T4all, RhsBatch B)
{
    Xf0 += (long long)blockIdx.z * B.xf; Wq0 += (long long)blockIdx.z * B.wq4;
    __shared__ int s_d[2][QGN], s_t[2][QGN], s_rng[2][2][2];
    __shared__ double s_part[QB_ROWS / 64][QGN], s_w[QGN], s_y[QGN], s_yp[QB_ROWS / QGN][QGN];
    const QbDesc qdd = qd[blockIdx.y];
    const int f = qdd.f, nslab = qdd.nslab;
    if ((int)blockIdx.x >= nslab) return;
    const FrontSym s = c.fs[f];
    const int fm = c.fnum[f].fm;
    if (fm <= 0 || s.fn <= 0) return;
    const int ng = (s.npanels + QG - 1) / QG;
    if (k > ng) return;
    // launch k: apply the (k-1)-th group of the order, form the partials of the k-th (Q'x: ascending, Q x: descending)
    const int gg[2] = {(k >= 1) ? (method ? ng - k : k - 1) : -1, (k < ng) ? (method ? ng - 1 - k : k) : -1};
    double *Xf = Xf0 + qdd.xoff;
    const int *Dq = Dq0 + qdd.dqoff;
    double *Wq = Wq0 + (long long)qdd.wqoff * QG;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int *St = c.Stair + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int i = blockIdx.x * QB_ROWS + tid, ic = min(i, fm - 1);
    if (tid < 2 * QGN) {
        const int w = tid >> 7, j = tid & (QGN - 1), kc = gg[w] * QGN + j;
        int d = -1, t = 0;
        if (gg[w] >= 0 && kc < s.fn) { d = Dq[kc]; t = St[kc]; }
        s_d[w][j] = (d >= 0) ? d : STM_BIGROW;
        s_t[w][j] = (d >= 0) ? max(t, d + 1) : 0;
        const int lo = wave_max_int(-s_d[w][j]), hi = wave_max_int(s_t[w][j]);
        if (lane == 0) { s_rng[w][(tid >> 6) & 1][0] = -lo; s_rng[w][(tid >> 6) & 1][1] = hi; }
    }
    double x = (i < fm) ? Xf[i] : 0.0;
    __syncthreads();
    bool on[2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const int r0 = min(s_rng[w][0][0], s_rng[w][1][0]), r1 = max(s_rng[w][0][1], s_rng[w][1][1]);
        on[w] = (gg[w] >= 0 && r0 < STM_BIGROW && (int)blockIdx.x * QB_ROWS < r1 && ((int)blockIdx.x + 1) * QB_ROWS > r0);      // (uniform)
    }
    // Both groups' columns go through ONE pipeline of 16 sub-blocks of 16 columns (8 of the group to apply, then 8 of the group whose
    // partials are formed: their loads do not depend on x), four sub-blocks (64 loads per thread) in flight: a launch is a handful of
    // memory round trips whatever it does, so what counts is how many loads each of them carries.
    constexpr int SB = 16, NSB = QGN / SB, DEPTH = 4;
    auto load_v = [&](int idx, double (&v)[SB]) {       // sub-block idx of the pipeline (uniform: nothing is loaded for an idle phase)
        const int w = idx / NSB, sb = idx % NSB;
        if (!on[w]) return;
        const double *Vp = F + (long long)(gg[w] * QGN + sb * SB) * ld;
        const int nbp = s.fn - (gg[w] * QGN + sb * SB);
#pragma unroll
        for (int j = 0; j < SB; j++) v[j] = (nbp > 0) ? Vp[ic + (long long)min(j, nbp - 1) * ld] : 0.0;     // unconditional, masked below
    };
    double buf[DEPTH][SB];
#pragma unroll
    for (int q = 0; q < DEPTH; q++)
#pragma unroll
        for (int j = 0; j < SB; j++) buf[q][j] = 0.0;
    if (on[0]) {
        // y = T4' w (Q'x) or T4 w (Q x): lanes along the output index, four parts of 32 terms each -- requested first
        const double *M = T4all + t4off[blockIdx.y] + (long long)gg[0] * QT4_DOUBLES + (method ? 0 : QGN * QGN);
        const int o = tid & (QGN - 1), part = tid >> 7;
        double m[32];
#pragma unroll
        for (int q = 0; q < 32; q++) m[q] = M[o + QGN * (32 * part + q)];
        if (tid < QGN) s_w[tid] = stm_ordered_sum<false>(Wq + (long long)(gg[0] & 1) * nslab * QGN + tid, QGN, nslab);      // fixed order
#pragma unroll
        for (int q = 0; q < DEPTH; q++) load_v(q, buf[q]);
        __syncthreads();
        {
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
            for (int q = 0; q < 32; q += 4) {
                p0 += m[q] * s_w[32 * part + q];
                p1 += m[q + 1] * s_w[32 * part + q + 1];
                p2 += m[q + 2] * s_w[32 * part + q + 2];
                p3 += m[q + 3] * s_w[32 * part + q + 3];
            }
            s_yp[part][o] = (p0 + p1) + (p2 + p3);
        }
        __syncthreads();
        if (tid < QGN) s_y[tid] = (s_yp[0][tid] + s_yp[1][tid]) + (s_yp[2][tid] + s_yp[3][tid]);
        __syncthreads();
    } else {
#pragma unroll
        for (int q = 0; q < DEPTH; q++) load_v(q, buf[q]);
    }
    double a = x;
#pragma unroll
    for (int idx = 0; idx < 2 * NSB; idx++) {
        double (&v)[SB] = buf[idx % DEPTH];
        const int w = idx / NSB, sb = idx % NSB;
        if (idx == NSB) {                               // between the phases: x after the group that was applied
            if (on[0] && i < fm && a != x) Xf[i] = a;
            x = (i < fm) ? a : 0.0;
        }
        if (w == 0) {
            if (on[0]) {
#pragma unroll
                for (int j = 0; j < SB; j++) {
                    const int d = s_d[0][sb * SB + j], t = s_t[0][sb * SB + j];
                    const double vv = (i > d && i < t) ? v[j] : ((i == d) ? 1.0 : 0.0);
                    a -= vv * s_y[sb * SB + j];
                }
            }
        } else if (gg[1] >= 0) {
            double accv[SB];
#pragma unroll
            for (int j = 0; j < SB; j++) {
                const int d = s_d[1][sb * SB + j], t = s_t[1][sb * SB + j];
                const double vv = (i < fm && i > d && i < t) ? v[j] : ((i == d) ? 1.0 : 0.0);
                accv[j] = on[1] ? vv * x : 0.0;
            }
            double part[8];
#pragma unroll
            for (int q = 0; q < SB / 8; q++) {
#pragma unroll
                for (int xx = 0; xx < 8; xx++) part[xx] = accv[8 * q + xx];
                const double rw = wave_reduce8(part);                             // lane l: total of value red8_idx(l)
                if (lane < 8) s_part[wid][sb * SB + 8 * q + red8_idx(lane)] = rw;
            }
        }
        if (idx + DEPTH < 2 * NSB) load_v(idx + DEPTH, v);
    }
    if (gg[1] >= 0) {
        __syncthreads();
        if (tid < QGN) {
            double vsum = 0;
#pragma unroll
            for (int w = 0; w < QB_ROWS / 64; w++) vsum += s_part[w][tid];
            Wq[((long long)(gg[1] & 1) * nslab + blockIdx.x) * QGN + tid] = vsum;
        }
    }
}

#define RS_NT 1024               // the back substitution streams R through one workgroup: more loads in flight
__global__ __launch_bounds__(RS_NT) void k_rsolve(DevCtx c, const int *__restrict__ flist, const int *__restrict__ Rj,
                                                  const double *W, double *X, int *err, RhsBatch B)
{
    W += (long long)blockIdx.y * B.w; X += (long long)blockIdx.y * B.x;
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[RS_NT / 64];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const FrontNum nm = c.fnum[f];
    const int fp = s.fp, fn = s.fn, fm = nm.fm;
    if (fp <= 0 || s.qbig) return;                      // (qbig: k_rbig_* below)
    const int tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    const int *Hi = c.Hii + s.hip;
    const int *rj = Rj + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    double *acc = dyn_lds;                              // [fp]
    double *xo = acc + ((fp + 1) & ~1);                 // [fn - fp] x of the non-pivotal columns
    int *lc = (int *)(xo + ((fn - fp + 1) & ~1));       // [fp] live pivot columns, compact
    // ---- live pivot columns (HStair != 0 and a row left for the diagonal), dead ones: x = 0 ----
    int rm;
    {
        const int per = (fp + RS_NT - 1) / RS_NT;
        const int k0 = min(fp, tid * per), k1 = min(fp, k0 + per);
        int cnt = 0;
        for (int k = k0; k < k1; k++) cnt += (St[k] != 0);
        int total;
        const int incl = qa_incl_scan<RS_NT / 64>(cnt, s_scan, &total);
        int q = incl - cnt;
        for (int k = k0; k < k1; k++) {
            if (St[k] != 0 && q < fm) lc[q] = k;
            else if (St[k] == 0) X[s.col1 + k] = 0.0;
            q += (St[k] != 0);
        }
        rm = min(total, fm);
    }
    for (int k = fp + tid; k < fn; k += RS_NT) xo[k - fp] = X[rj[k]];
    __syncthreads();
    if (rm != nm.rank && tid == 0) atomicExch(err, 1);  // (cannot happen: same rule as the factorization)
    // acc = y - R12 x_others : thread per row, columns streamed (coalesced over the rows)
    for (int i = tid; i < rm; i += RS_NT) {
        double a0 = W[Hi[i]], a1 = 0, a2 = 0, a3 = 0;  // (four partial sums, 16 loads in flight: as k_rbig_init)
        int k = fp;
        for (; k + 16 <= fn; k += 16) {
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                a0 -= F[i + (long long)(k + u) * ld] * xo[k + u - fp];
                a1 -= F[i + (long long)(k + u + 1) * ld] * xo[k + u + 1 - fp];
                a2 -= F[i + (long long)(k + u + 2) * ld] * xo[k + u + 2 - fp];
                a3 -= F[i + (long long)(k + u + 3) * ld] * xo[k + u + 3 - fp];
            }
        }
        for (; k < fn; k++) a0 -= F[i + (long long)k * ld] * xo[k - fp];
        acc[i] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    // triangle: blocked back substitution over the compact list.  Per block of QS_NB live columns: the diagonal triangle
    // goes to LDS and one wave solves it there (no global latency inside the 32 dependent steps), then every thread
    // updates its rows of acc with the block's columns (coalesced over the rows, 32 independent loads in flight).
    constexpr int QS_NB = 32;
    __shared__ double s_tri[QS_NB][QS_NB + 1];
    __shared__ double s_x[QS_NB];
    for (int kb = ((max(rm, 1) - 1) / QS_NB) * QS_NB; kb >= 0 && rm > 0; kb -= QS_NB) {
        const int nb = min(QS_NB, rm - kb);
        for (int e = tid; e < QS_NB * QS_NB; e += RS_NT) {
            const int i = e % QS_NB, j = e / QS_NB;
            s_tri[i][j] = (i < nb && j < nb && i <= j) ? F[(kb + i) + (long long)lc[kb + j] * ld] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {
            // lane i owns row i of the triangle (i < nb): x_j for j = nb-1 .. 0
            const int i = tid;
            double a = (i < nb) ? acc[kb + i] : 0.0;
            for (int j = nb - 1; j >= 0; j--) {
                const double aj = __shfl(a, j, 64);
                const double xj = aj / s_tri[j][j];
                if (i < j) a -= s_tri[i][j] * xj;
                if (i == j) s_x[j] = xj;
            }
        }
        __syncthreads();
        if (tid < nb) X[s.col1 + lc[kb + tid]] = s_x[tid];
        for (int i = tid; i < kb; i += RS_NT) {
            double a = acc[i];
#pragma unroll 8
            for (int j = 0; j < nb; j++) a -= F[i + (long long)lc[kb + j] * ld] * s_x[j];
            acc[i] = a;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// R' x = b (qr_private_rtsolve, reference SparseQR.c:2522-2700), fronts leaves -> root, one workgroup per front.
// Every front carries a vector u over its fn columns: b minus the contributions of the rows solved so far.  A front
// starts with u = b on its pivotal columns and 0 elsewhere, ADDS its children's pass vectors (their u on the columns they
// hand up, mapped through Rjrel -- the assembly's column map, children in order: deterministic, no atomics), solves its
// triangle R11' x = u(pivots) forwards over the live pivot columns, and passes u(non-pivotal) - R12' x up.
//   Bp: b in R's column order (length n); U: the pass vectors, slot Rp[f] + k (rjsize); Xr: x in R's global row order
//   (rowbase[f] = rows of R above front f's).  Dead pivot columns have no equation (the squeezed R of the reference).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_NT) void k_rtsolve(DevCtx c, const int *__restrict__ flist, const double *__restrict__ Bp,
                                                   double *U, double *Xr, const int *__restrict__ rowbase, RhsBatch B)
{
    Bp += (long long)blockIdx.y * B.x; U += (long long)blockIdx.y * B.u; Xr += (long long)blockIdx.y * B.w;
    extern __shared__ double dyn_lds[];
    __shared__ int s_scan[RS_NT / 64];
    const int f = flist[blockIdx.x];
    const FrontSym s = c.fs[f];
    const FrontNum nm = c.fnum[f];
    const int fp = s.fp, fn = s.fn, fm = nm.fm;
    const int tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    double *u = dyn_lds;                                // [fn]
    double *x = u + ((fn + 1) & ~1);                    // [min(fp, fm)] the front's rows of x
    int *lc = (int *)(x + ((min(fp, max(fm, 1)) + 2) & ~1));   // [fp] live pivot columns, compact
    int rm;
    {
        const int per = (fp + RS_NT - 1) / RS_NT;
        const int k0 = min(fp, tid * per), k1 = min(fp, k0 + per);
        int cnt = 0;
        for (int k = k0; k < k1; k++) cnt += (St[k] != 0);
        int total;
        const int incl = qa_incl_scan<RS_NT / 64>(cnt, s_scan, &total);
        int q = incl - cnt;
        for (int k = k0; k < k1; k++) {
            if (St[k] != 0 && q < fm) lc[q] = k;
            q += (St[k] != 0);
        }
        rm = min(total, fm);
    }
    for (int k = tid; k < fn; k += RS_NT) u[k] = (k < fp) ? Bp[s.col1 + k] : 0.0;
    __syncthreads();
    for (int q = s.child0; q < s.child1; q++) {
        const int ch = c.Child[q];
        const FrontSym cs = c.fs[ch];
        const int pc = cs.rp + cs.fp, cn = cs.fn - cs.fp;
        for (int cj = tid; cj < cn; cj += RS_NT) u[c.Rjrel[pc + cj]] += U[pc + cj];     // (distinct targets within a child)
        __syncthreads();
    }
    // forward substitution over the live pivot columns in blocks of 32: (a) the block's right-hand sides lose the rows
    // solved before (32 columns x 32 row lanes per pass), (b) one wave solves the 32 x 32 lower triangle R' in LDS
    constexpr int QS_NB = 32;
    __shared__ double s_tri[QS_NB][QS_NB + 1];
    __shared__ double s_rhs[QS_NB];
    const int lane32 = tid & 31, grp = tid >> 5;                          // 32 groups of 32 lanes
    for (int kb = 0; kb < rm; kb += QS_NB) {
        const int nb = min(QS_NB, rm - kb);
        {
            const int j = grp;                                            // column of the block
            double a = 0;
            if (j < nb) {
                const double *col = F + (long long)lc[kb + j] * ld;
                for (int r = lane32; r < kb; r += 32) a += col[r] * x[r];
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane32 == 0 && j < nb) s_rhs[j] = u[lc[kb + j]] - a;
        }
        for (int e = tid; e < QS_NB * QS_NB; e += RS_NT) {
            const int i = e % QS_NB, j = e / QS_NB;                       // R(kb + i, column j of the block), i <= j
            s_tri[i][j] = (i < nb && j < nb && i <= j) ? F[(kb + i) + (long long)lc[kb + j] * ld] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {
            // lane j owns equation j: x_j = (rhs_j - sum_{i<j} R(i,j) x_i) / R(j,j)
            const int j = tid;
            double a = (j < nb) ? s_rhs[j] : 0.0;
            for (int i = 0; i < nb; i++) {
                const double ai = __shfl(a, i, 64);
                const double xi = ai / s_tri[i][i];
                if (j > i && j < nb) a -= s_tri[i][j] * xi;
                if (j == i) x[kb + i] = xi;
            }
        }
        __syncthreads();
    }
    // pass up: u(non-pivotal) -= R12' x ; a thread per column, the front's rows of R streamed
    for (int k = fp + tid; k < fn; k += RS_NT) {
        const double *col = F + (long long)k * ld;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int r = 0;
        for (; r + 4 <= rm; r += 4) { a0 += col[r] * x[r]; a1 += col[r + 1] * x[r + 1]; a2 += col[r + 2] * x[r + 2]; a3 += col[r + 3] * x[r + 3]; }
        for (; r < rm; r++) a0 += col[r] * x[r];
        U[s.rp + k] = u[k] - ((a0 + a1) + (a2 + a3));
    }
    for (int r = tid; r < rm; r += RS_NT) Xr[rowbase[f] + r] = x[r];
}

// scatter: out[perm[i]] = in[i]   gather: out[i] = in[perm[i]]   (perm == nullptr: identity)
// ------------------------------------------------------------------------------------------------
// Back substitution for the large fronts (FrontSym::qbig), rows split over workgroups as in k_qbig_*: prep (live pivot
// columns -> Lc, rm), init (acc = y - R12 x_others, a row per thread), then one launch per block of 32 live columns
// from the last to the first: every active workgroup solves the 32 x 32 triangle for itself (one wave, LDS) and updates
// its rows above the block; workgroup 0 stores x.  All split fronts of a level advance together (blockIdx.y).
// acc = the front's slice of Xf, Lc = its slice of Dq (the Q-apply has finished with both).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RS_NT) void k_rbig_prep(DevCtx c, const QbDesc *__restrict__ qd, double *X, int *Lc0, int *Rm, int *err, RhsBatch B)
{
    X += (long long)blockIdx.y * B.x;                      // (the live-column lists Lc0 / Rm are the same for every right-hand side)
    __shared__ int s_scan[RS_NT / 64];
    const QbDesc d = qd[blockIdx.x];
    const FrontSym s = c.fs[d.f];
    const FrontNum nm = c.fnum[d.f];
    const int fp = s.fp, fm = nm.fm, tid = threadIdx.x;
    const int *St = c.Stair + s.rp;
    int *lc = Lc0 + d.dqoff;
    const int per = (fp + RS_NT - 1) / RS_NT;
    const int k0 = min(fp, tid * per), k1 = min(fp, k0 + per);
    int cnt = 0;
    for (int k = k0; k < k1; k++) cnt += (St[k] != 0);
    int total;
    const int incl = qa_incl_scan<RS_NT / 64>(cnt, s_scan, &total);
    int q = incl - cnt;
    for (int k = k0; k < k1; k++) {
        if (St[k] != 0 && q < fm) lc[q] = k;
        else if (St[k] == 0) X[s.col1 + k] = 0.0;       // dead pivot column: basic solution
        q += (St[k] != 0);
    }
    if (tid == 0) {
        const int rm = min(total, fm);
        Rm[blockIdx.x] = rm;
        if (rm != nm.rank) atomicExch(err, 1);
    }
}
__global__ __launch_bounds__(STM_QB_ROWS) void k_rbig_init(DevCtx c, const QbDesc *__restrict__ qd, const int *__restrict__ Rj,
                                                           const double *W, const double *X, double *Acc0, const int *Rm, RhsBatch B)
{
    W += (long long)blockIdx.z * B.w; X += (long long)blockIdx.z * B.x; Acc0 += (long long)blockIdx.z * B.xf;
    // y - R12 x2 for the rows of the live pivot columns.  A workgroup takes 64 rows; its eight waves share the non-pivotal columns
    // (chunks of 16, wave w the chunks w, w + 8, ...) and their partial sums are added in wave order (round 4: a thread per row ran
    // through all the columns alone -- 250 dependent round trips on a front with 4000 of them, 276 us per launch).
    __shared__ double s_p[STM_QB_ROWS / 64][64];
    const QbDesc d = qd[blockIdx.y];
    const FrontSym s = c.fs[d.f];
    const int rm = Rm[blockIdx.y];
    const int lrow = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lrow;
    if (blockIdx.x * 64 >= rm) return;
    const int ic = min(i, rm - 1);
    const int *rj = Rj + s.rp;
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    // (coalesced over the rows; X[rj[k]] uniform; four partial sums and an unrolled body keep 16 loads in flight)
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const int nfull = (s.fn - s.fp) / 16;
    for (int ch = cl; ch < nfull; ch += STM_QB_ROWS / 64) {
        const int k = s.fp + 16 * ch;
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            a0 -= F[ic + (long long)(k + u) * ld] * X[rj[k + u]];
            a1 -= F[ic + (long long)(k + u + 1) * ld] * X[rj[k + u + 1]];
            a2 -= F[ic + (long long)(k + u + 2) * ld] * X[rj[k + u + 2]];
            a3 -= F[ic + (long long)(k + u + 3) * ld] * X[rj[k + u + 3]];
        }
    }
    if (cl == 0)
        for (int k = s.fp + 16 * nfull; k < s.fn; k++) a0 -= F[ic + (long long)k * ld] * X[rj[k]];
    s_p[cl][lrow] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (cl == 0 && i < rm) {
        double a = W[c.Hii[s.hip + i]];
#pragma unroll
        for (int w = 0; w < STM_QB_ROWS / 64; w++) a += s_p[w][lrow];
        Acc0[d.xoff + i] = a;
    }
}
__global__ __launch_bounds__(STM_QB_ROWS) void k_rbig_step(DevCtx c, const QbDesc *__restrict__ qd, int kstep, double *X, double *Acc0,
                                                           const int *Lc0, const int *Rm, RhsBatch B)
{
    X += (long long)blockIdx.z * B.x; Acc0 += (long long)blockIdx.z * B.xf;
    constexpr int QS_NB = 32;
    __shared__ double s_tri[QS_NB][QS_NB + 1];
    __shared__ double s_x[QS_NB];
    const QbDesc d = qd[blockIdx.y];
    const int rm = Rm[blockIdx.y];
    const int nblk = (rm + QS_NB - 1) / QS_NB;
    if (kstep >= nblk) return;
    const int kb = (nblk - 1 - kstep) * QS_NB, nb = min(QS_NB, rm - kb);
    const int sl = blockIdx.x, tid = threadIdx.x;
    if (sl != 0 && sl * STM_QB_ROWS >= kb) return;      // no row above the block (workgroup 0 always runs: it stores x)
    const FrontSym s = c.fs[d.f];
    const double *F = c.Farena + s.foff;
    const long long ld = s.ld;
    const int *lc = Lc0 + d.dqoff;
    double *acc = Acc0 + d.xoff;
    // my row's entries of the block's columns: requested before the triangle is solved
    const int i = sl * STM_QB_ROWS + tid;
    double rv[QS_NB];
#pragma unroll
    for (int j = 0; j < QS_NB; j++) rv[j] = F[min(i, max(kb - 1, 0)) + (long long)lc[kb + min(j, nb - 1)] * ld];
    double a = (i < kb) ? acc[i] : 0.0;
    for (int e = tid; e < QS_NB * QS_NB; e += STM_QB_ROWS) {
        const int ti = e % QS_NB, tj = e / QS_NB;
        s_tri[ti][tj] = (ti < nb && tj < nb && ti <= tj) ? F[(kb + ti) + (long long)lc[kb + tj] * ld] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        // lane r owns row r of the triangle (r < nb): x_j for j = nb-1 .. 0
        double t = (tid < nb) ? acc[kb + tid] : 0.0;
        for (int j = nb - 1; j >= 0; j--) {
            const double aj = __shfl(t, j, 64);
            const double xj = aj / s_tri[j][j];
            if (tid < j) t -= s_tri[tid][j] * xj;
            if (tid == j) s_x[j] = xj;
        }
    }
    __syncthreads();
    if (sl == 0 && tid < nb) X[s.col1 + lc[kb + tid]] = s_x[tid];
    if (i < kb) {
#pragma unroll
        for (int j = 0; j < QS_NB; j++) a -= ((j < nb) ? rv[j] : 0.0) * s_x[min(j, nb - 1)];
        acc[i] = a;
    }
}

__global__ __launch_bounds__(256) void k_perm(const double *__restrict__ in, const int *__restrict__ perm, double *out, int n,
                                               int scatter, long long sin, long long sout)
{
    in += (long long)blockIdx.y * sin; out += (long long)blockIdx.y * sout;      // (vector blockIdx.y of a batch)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = perm ? perm[i] : i;
    if (scatter) out[j] = in[i];
    else out[i] = in[j];
}

// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

int stm_launch_sigma(const double *Ax, int anz, unsigned long long *amaxbits, double *sig, hipStream_t st)
{
    CK(hipMemsetAsync(amaxbits, 0, sizeof(unsigned long long), st));
    if (anz > 0) {
        int grid = (anz + 255) / 256;
        if (grid > 1024) grid = 1024;
        hipLaunchKernelGGL(k_amax, dim3(grid), dim3(256), 0, st, Ax, anz, amaxbits);
    }
    hipLaunchKernelGGL(k_sigma, dim3(1), dim3(1), 0, st, (const unsigned long long *)amaxbits, sig);
    return (int)hipGetLastError();
}
int stm_launch_gather_sx(const double *Ax, const int *smap, double *Sx, int anz, hipStream_t st)
{
    if (anz <= 0) return 0;
    int grid = (anz + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_gather_sx, dim3(grid), dim3(256), 0, st, Ax, smap, Sx, anz);
    return (int)hipGetLastError();
}
int stm_launch_setup(const DevCtx &c, const int *flist, int nfr, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_setup, dim3(nfr), dim3(NT), 0, st, c, flist);
    return (int)hipGetLastError();
}
int stm_launch_assemble(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_assemble, dim3(maxparts, nfr), dim3(NT), 0, st, c, flist, nparts);
    return (int)hipGetLastError();
}
int stm_update_lds_bytes(void) { return (int)((2 * BN * VS + STM_NB * WS) * sizeof(double)); }
int stm_launch_front_wg(const DevCtx &c, const int *flist, int nfr, int lds_doubles, hipStream_t st)
{
    if (nfr <= 0) return 0;
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();
    hipLaunchKernelGGL(k_front_wg, dim3(nfr), dim3(NT), bytes, st, c, flist, (int)(bytes / sizeof(double)));
    return (int)hipGetLastError();
}
int stm_launch_panel(const DevCtx &c, const int *flist, const int *plist, int nfr, int nsub, int defer_ok, int lds_doubles, hipStream_t st)
{
    if (nfr <= 0) return 0;
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();   // in-panel MFMA update + Gram scratch
    // One workgroup per column group of the panel pipeline (blockIdx.y); fronts whose panel is not pipelined use group 0.
    // The column groups of a pipelined panel wait for each other, but only ever for groups with a smaller blockIdx.y of the
    // same front, i.e. for workgroups that the in-order dispatch has already started (the assumption of every
    // decoupled-look-back scan); the waits are bounded and a wait that runs out is recovered (stmmqr_factorize_device).
    // Oversubscribed launches (more workgroups than the GPU holds: groups start late) are exercised by the tests;
    // STMMQR_DBG bit 9 + STMMQR_CHUNK launch the fronts in chunks instead (tests).
    int K = nfr;
    if (nsub > 1 && (c.dbg & 512)) K = getenv("STMMQR_CHUNK") ? atoi(getenv("STMMQR_CHUNK")) : 1;
    if (K < 1) K = 1;
    for (int i = 0; i < nfr; i += K)
        hipLaunchKernelGGL(k_panel, dim3(nfr - i < K ? nfr - i : K, nsub), dim3(NTP), bytes, st, c, flist + i, plist + i, nsub, defer_ok,
                           (int)(bytes / sizeof(double)));
    return (int)hipGetLastError();
}
int stm_launch_update(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, hipStream_t st)
{
    if (nfr <= 0 || ncb <= 0) return 0;
    hipLaunchKernelGGL(k_update, dim3(ncb, nfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, flist, plist, cb0);
    return (int)hipGetLastError();
}
int stm_launch_update_split(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, double *Wp,
                            const long long *wpoff, int *wcnt, int with_gram, hipStream_t st)
{
    if (nfr <= 0 || ncb + (with_gram ? 1 : 0) <= 0 || maxsl <= 0) return 0;
    const size_t lds = (size_t)stm_update_lds_bytes();
    // (with_gram: one more column block, V'V for the fronts whose panel kernel left T to the update)
    // (k_upd_w stages V and C chunks only: 34 KB, four workgroups per CU)
    hipLaunchKernelGGL(k_upd_w, dim3(ncb + (with_gram ? 1 : 0), maxsl, nfr), dim3(NT), (size_t)(2 * BN * VS) * sizeof(double), st, c,
                       flist, plist, cb0, with_gram ? 1 : 0, Wp, wpoff, wcnt);
    if (ncb > 0) hipLaunchKernelGGL(k_upd_c, dim3(ncb, maxsl, nfr), dim3(NT), lds, st, c, flist, plist, cb0, (const double *)Wp, wpoff);
    return (int)hipGetLastError();
}
int stm_launch_update_fused(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, double *Wp,
                            const long long *wpoff, int *wcnt, int *wflag, int epoch, int with_gram, hipStream_t st)
{
    const int ny = ncb + (with_gram ? 1 : 0);
    if (nfr <= 0 || ny <= 0 || maxsl <= 0) return 0;
    hipLaunchKernelGGL(k_upd_f, dim3(maxsl, ny, nfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, flist, plist, cb0,
                       with_gram ? 1 : 0, Wp, wpoff, wcnt, wflag, epoch);
    return (int)hipGetLastError();
}
// Passenger launches (k_panel_pc / k_upd_fw above).  A: the panels of a step + k_upd_c of the column blocks ucb0 .. ucb0 + uncb - 1 of the
// fronts (uflist, uplist, uwpoff: the PREVIOUS step's lists) out of the passengers' workspace.
int stm_launch_panel_pc(const DevCtx &c, const int *flist, const int *plist, int nfr, int nsub, int defer_ok, int lds_doubles,
                        const int *uflist, const int *uplist, int unfr, int ucb0, int uncb, int umaxsl, const double *Wp,
                        const long long *uwpoff, hipStream_t st)
{
    if (unfr <= 0 || uncb <= 0 || umaxsl <= 0 || (c.dbg & 512)) {
        if (unfr > 0 && uncb > 0 && umaxsl > 0)
            hipLaunchKernelGGL(k_upd_c, dim3(uncb, umaxsl, unfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, uflist, uplist, ucb0, Wp, uwpoff);
        return stm_launch_panel(c, flist, plist, nfr, nsub, defer_ok, lds_doubles, st);
    }
    if (nfr <= 0 && !(nfr < 0)) {
        hipLaunchKernelGGL(k_upd_c, dim3(uncb, umaxsl, unfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, uflist, uplist, ucb0, Wp, uwpoff);
        return (int)hipGetLastError();
    }
    if (nfr < 0) nfr = 0;                                         // (measurements: the riders alone, in their rider form)
    size_t bytes = (size_t)lds_doubles * sizeof(double);
    if (bytes < (size_t)stm_update_lds_bytes()) bytes = stm_update_lds_bytes();
    const int lds_arg = (int)(bytes / sizeof(double));            // what k_panel would be told: its rules must not see the passengers
    if (bytes < (size_t)STM_PC_LDS_DOUBLES * sizeof(double)) bytes = (size_t)STM_PC_LDS_DOUBLES * sizeof(double);
    // slabs per rider workgroup: a rider has its CU to itself (the launch carries the panel's registers and LDS), so its descriptor
    // chain and prologue (~5 us) overlap with nothing; per slab ~2.4 us.  Rounds of ~240 workgroups (urows: the launch's tiles, a bound)
    int rspw = 1;
    {
        static int force = -1;
        if (force < 0) force = getenv("STMMQR_RSPW") ? atoi(getenv("STMMQR_RSPW")) : 0;
        double best = 1e30;
        for (int k = 1; k <= 16; k *= 2) {
            const long wgs = (long)unfr * uncb * ((umaxsl + k - 1) / k);
            const double t = (double)((wgs + 239) / 240) * (5.0 + 1.0 * k);
            if (t < best) { best = t; rspw = k; }
        }
        if (force > 0) rspw = force;
    }
    const int uy = (umaxsl + rspw - 1) / rspw;
    hipLaunchKernelGGL(k_panel_pc, dim3(nfr > uncb ? nfr : uncb, nsub > uy ? nsub : uy, 1 + unfr), dim3(NTP), bytes, st, c, flist,
                       plist, nfr, nsub, defer_ok, lds_arg, uflist, uplist, ucb0, Wp, uwpoff, rspw);
    return (int)hipGetLastError();
}
// B: T + column block 0 of the step's fronts in one fused launch (as stm_launch_update_fused(cb0 = 0, ncb = 1, with_gram)) + k_upd_w of
// their column blocks 1 .. ncb - 1 into the passengers' workspace (Wp2 / wcnt2)
int stm_launch_update_fw(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncb, int maxsl, double *Wp,
                         const long long *wpoff, int *wcnt, int *wflag, int epoch, double *Wp2, int *wcnt2, hipStream_t st)
{
    if (nfr <= 0 || maxsl <= 0) return 0;
    const int rest = ncb - 1;
    static int one = -1;
    if (one < 0) one = getenv("STMMQR_B0_ONE") ? atoi(getenv("STMMQR_B0_ONE")) : 1;       // (0: k_upd_f with its two meeting points)
    if (one) {
        // slabs per rider workgroup: two (measured on the default workload: 1 slab 110.8 ms, 2 109.7, 4 110.1, 8 113.0 -- these riders
        // share their CU with a second workgroup, so less of a descriptor chain is exposed than in the panel launch)
        static int rspw = -1;
        if (rspw < 0) rspw = getenv("STMMQR_RSPW_W") && atoi(getenv("STMMQR_RSPW_W")) > 0 ? atoi(getenv("STMMQR_RSPW_W")) : 2;
        const int uy = (maxsl + rspw - 1) / rspw;
        hipLaunchKernelGGL(k_upd_b0w, dim3(maxsl > rest ? maxsl : rest, rest > 0 ? uy : 1, rest > 0 ? 2 * nfr : nfr), dim3(NT),
                           (size_t)stm_update_lds_bytes(), st, c, flist, plist, nfr, maxsl, rest > 0 ? rest : 0, Wp, wpoff, wcnt, wflag, epoch,
                           Wp2, wcnt2, rspw);
        return (int)hipGetLastError();
    }
    if (rest <= 0) return stm_launch_update_fused(c, flist, plist, nfr, 0, 1, maxsl, Wp, wpoff, wcnt, wflag, epoch, 1, st);
    hipLaunchKernelGGL(k_upd_fw, dim3(maxsl > rest ? maxsl : rest, maxsl > 2 ? maxsl : 2, 2 * nfr), dim3(NT), (size_t)stm_update_lds_bytes(),
                       st, c, flist, plist, nfr, maxsl, rest, Wp, wpoff, wcnt, wflag, epoch, Wp2, wcnt2);
    return (int)hipGetLastError();
}
static int stm_anyorder(void)
{
    static int v = -1;
    if (v < 0) v = getenv("STMMQR_ANYORDER") ? atoi(getenv("STMMQR_ANYORDER")) : 0;
    return v;
}
// k_upd_w alone into the passengers' workspace (measurements: the riders as launches of their own)
int stm_launch_update_w(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, double *Wp,
                        const long long *wpoff, int *wcnt, hipStream_t st)
{
    if (nfr <= 0 || ncb <= 0 || maxsl <= 0) return 0;
    if (stm_anyorder())
        hipExtLaunchKernelGGL(k_upd_w, dim3(ncb, maxsl, nfr), dim3(NT), (size_t)(2 * BN * VS) * sizeof(double), st, nullptr, nullptr,
                              hipExtAnyOrderLaunch, c, flist, plist, cb0, 0, Wp, wpoff, wcnt);
    else
    hipLaunchKernelGGL(k_upd_w, dim3(ncb, maxsl, nfr), dim3(NT), (size_t)(2 * BN * VS) * sizeof(double), st, c, flist, plist, cb0, 0, Wp,
                       wpoff, wcnt);
    return (int)hipGetLastError();
}
// k_upd_c alone (the passengers' last phase when no panel launch follows)
int stm_launch_update_c(const DevCtx &c, const int *flist, const int *plist, int nfr, int cb0, int ncb, int maxsl, const double *Wp,
                        const long long *wpoff, hipStream_t st)
{
    if (nfr <= 0 || ncb <= 0 || maxsl <= 0) return 0;
    if (stm_anyorder())
        hipExtLaunchKernelGGL(k_upd_c, dim3(ncb, maxsl, nfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, nullptr, nullptr,
                              hipExtAnyOrderLaunch, c, flist, plist, cb0, Wp, wpoff);
    else
    hipLaunchKernelGGL(k_upd_c, dim3(ncb, maxsl, nfr), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, flist, plist, cb0, Wp, wpoff);
    return (int)hipGetLastError();
}
int stm_launch_update_pair(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncbp, int maxsl, double *Wp,
                           const long long *wpoff, int *wcnt, hipStream_t st)
{
    if (nfr <= 0 || ncbp <= 0 || maxsl <= 0) return 0;
    hipLaunchKernelGGL(k_upd_w2, dim3(ncbp + 1, maxsl, nfr), dim3(NT), (size_t)(3 * BN * VS2 + 2) * sizeof(double), st, c, flist, plist, Wp,
                       wpoff, wcnt);
    hipLaunchKernelGGL(k_upd_y2, dim3(ncbp, nfr), dim3(NT), (size_t)(6 * STM_NB * WS) * sizeof(double), st, c, flist, plist, Wp, wpoff);
    hipLaunchKernelGGL(k_upd_c2, dim3(ncbp, maxsl, nfr), dim3(NT), 0, st, c, flist, plist, (const double *)Wp, wpoff);
    return (int)hipGetLastError();
}
int stm_launch_update_quad(const DevCtx &c, const int *flist, const int *plist, int nfr, int ncbp, int maxsl, double *Wp,
                           const long long *wpoff, int *wcnt, hipStream_t st)
{
    if (nfr <= 0 || ncbp <= 0 || maxsl <= 0) return 0;
    hipLaunchKernelGGL(k_upd_wq, dim3(ncbp + QP - 1, maxsl, nfr), dim3(NT), (size_t)((QP + 1) * BN * VSQ + 2) * sizeof(double), st, c, flist,
                       plist, Wp, wpoff, wcnt);
    hipLaunchKernelGGL(k_upd_yq, dim3(ncbp, nfr), dim3(NT), (size_t)((QP + 3) * STM_NB * WS) * sizeof(double), st, c, flist, plist, Wp, wpoff);
    hipLaunchKernelGGL(k_upd_cq, dim3(ncbp, maxsl, nfr), dim3(NT), 0, st, c, flist, plist);
    return (int)hipGetLastError();
}
int stm_launch_update_notrans(const DevCtx &c, int f, int ncb, hipStream_t st)
{
    if (ncb <= 0) return 0;
    hipLaunchKernelGGL(k_update_n, dim3(ncb), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, f);
    return (int)hipGetLastError();
}
int stm_launch_larft(const DevCtx &c, int f, hipStream_t st)
{
    hipLaunchKernelGGL(k_larft, dim3(1), dim3(NT), (size_t)stm_update_lds_bytes(), st, c, f);
    return (int)hipGetLastError();
}
int stm_launch_cpack(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, hipStream_t st)
{
    if (nfr <= 0) return 0;
    // (+1: the workgroup that builds a pending T of the last panel, dev_tlast; its Gram scratch is the dynamic LDS)
    hipLaunchKernelGGL(k_cpack, dim3(maxparts + 1, nfr), dim3(NT), (size_t)(4 * 768) * sizeof(double), st, c, flist, nparts, maxparts);
    return (int)hipGetLastError();
}
int stm_launch_rh_count(const DevCtx &c, const int *flist, int nfr, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rh_count, dim3(nfr), dim3(NT), 0, st, c, flist);
    return (int)hipGetLastError();
}
int stm_launch_rh_scan(const DevCtx &c, const int *post, int nf, long long *rh_total, long long *outoff, hipStream_t st)
{
    hipLaunchKernelGGL(k_rh_scan, dim3(1), dim3(NT), 0, st, c, post, nf, rh_total, outoff);
    return (int)hipGetLastError();
}
int stm_launch_zero_slabs(const DevCtx &c, const int *flist, int nfr, int maxparts, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_zero_slabs, dim3(maxparts, nfr), dim3(256), 0, st, c, flist);
    return (int)hipGetLastError();
}
int stm_launch_rh_unpack(const DevCtx &c, const FrontSym *cs, const int *flist, int nfr, int maxparts, const char *kept, const double *RH,
                         double *scratch, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rh_unpack, dim3(maxparts, nfr), dim3(NT), 0, st, c, cs, flist, kept, RH, scratch, 0);
    hipLaunchKernelGGL(k_rh_unpack, dim3(maxparts, nfr), dim3(NT), 0, st, c, cs, flist, kept, RH, scratch, 1);
    return (int)hipGetLastError();
}
int stm_launch_rh_window(const DevCtx &c, const int *flist, int nfr, int maxparts, const long long *fin, const char *kept, const double *RH,
                         long long w0, long long w1, double *out, hipStream_t st)
{
    if (nfr <= 0 || w1 <= w0) return 0;
    hipLaunchKernelGGL(k_rh_window, dim3(maxparts, nfr), dim3(NT), 0, st, c, flist, fin, kept, RH, w0, w1, out);
    return (int)hipGetLastError();
}
int stm_launch_rh_copy(const DevCtx &c, const int *flist, const int *nparts, int nfr, int maxparts, double *RH,
                       hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rh_copy, dim3(maxparts, nfr), dim3(NT), 0, st, c, flist, nparts, RH);
    return (int)hipGetLastError();
}
int stm_launch_qapply(const DevCtx &c, const int *flist, int nfr, int method, double *W, int lds_bytes, int *err, hipStream_t st)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_qapply, dim3(nfr), dim3(QA_NT), (size_t)lds_bytes, st, c, flist, method, W, err);
    return (int)hipGetLastError();
}
int stm_launch_qapply_t(const DevCtx &c, const int *flist, int nfr, int method, double *W, int lds_bytes, hipStream_t st, int nb,
                        const RhsBatch &B)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_qapply_t, dim3(nfr, nb), dim3(QA_NT), (size_t)lds_bytes, st, c, flist, method, W, B);
    return (int)hipGetLastError();
}
// the split fronts of one level: prep, max(npanels) + 1 steps, finish
int stm_launch_qapply_big(const DevCtx &c, const QbDesc *qd, int nq, int max_npanels, int max_nslab, int max_fm, int method, double *W,
                          double *Xf, int *Dq, double *Wq, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nq <= 0 || max_npanels <= 0) return 0;
    hipLaunchKernelGGL(k_qbig_prep, dim3(nq, nb), dim3(QA_NT), 0, st, c, qd, (const double *)W, Xf, Dq, B);
    for (int k = 0; k <= max_npanels; k++)
        hipLaunchKernelGGL(k_qbig_step, dim3(max_nslab, nq, nb), dim3(QB_ROWS), 0, st, c, qd, k, method, Xf, (const int *)Dq, Wq, B);
    hipLaunchKernelGGL(k_qbig_finish, dim3((max_fm + 255) / 256, nq, nb), dim3(256), 0, st, c, qd, W, (const double *)Xf, B);
    return (int)hipGetLastError();
}
int stm_qt4_doubles(void) { return QT4_DOUBLES; }
// T4 of every group of every split front (items) -- once per factorization
int stm_launch_qt4_build(const DevCtx &c, const int *fl, const long long *dqo, int nfronts, const void *items, int nitems, int *Dq4, double *T4all,
                         hipStream_t st)
{
    if (nfronts <= 0 || nitems <= 0) return 0;
    hipLaunchKernelGGL(k_qt4_number, dim3(nfronts), dim3(QA_NT), 0, st, c, fl, dqo, Dq4);
    const size_t lds = sizeof(double) * (size_t)((QGN * QT4_VS > 5 * 32 * 33) ? QGN * QT4_VS : 5 * 32 * 33);
    hipLaunchKernelGGL(k_qt4_build, dim3(nitems), dim3(512), lds, st, c, (const Qt4Item *)items, (const int *)Dq4, T4all);
    return (int)hipGetLastError();
}
int stm_launch_qapply_big4(const DevCtx &c, const QbDesc *qd, const long long *t4off, int nq, int max_npanels, int max_nslab, int max_fm,
                           int method, double *W, double *Xf, int *Dq, double *Wq4, const double *T4all, hipStream_t st, int nb,
                           const RhsBatch &B)
{
    if (nq <= 0 || max_npanels <= 0) return 0;
    const int max_ng = (max_npanels + QG - 1) / QG;
    hipLaunchKernelGGL(k_qbig_prep, dim3(nq, nb), dim3(QA_NT), 0, st, c, qd, (const double *)W, Xf, Dq, B);
    for (int k = 0; k <= max_ng; k++)
        hipLaunchKernelGGL(k_qbig_step4, dim3(max_nslab, nq, nb), dim3(QB_ROWS), 0, st, c, qd, t4off, k, method, Xf, (const int *)Dq, Wq4, T4all, B);
    hipLaunchKernelGGL(k_qbig_finish, dim3((max_fm + 255) / 256, nq, nb), dim3(256), 0, st, c, qd, W, (const double *)Xf, B);
    return (int)hipGetLastError();
}
int stm_launch_rsolve(const DevCtx &c, const int *flist, int nfr, const int *Rj, const double *W, double *X, int lds_bytes,
                      int *err, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rsolve, dim3(nfr, nb), dim3(RS_NT), (size_t)lds_bytes, st, c, flist, Rj, W, X, err, B);
    return (int)hipGetLastError();
}
// back substitution of the split fronts of one level: prep, init, max ceil(fp / 32) steps
int stm_launch_rsolve_big(const DevCtx &c, const QbDesc *qd, int nq, int max_steps, int max_nslab, const int *Rj, const double *W,
                          double *X, double *Acc, int *Lc, int *Rm, int *err, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nq <= 0) return 0;
    hipLaunchKernelGGL(k_rbig_prep, dim3(nq, nb), dim3(RS_NT), 0, st, c, qd, X, Lc, Rm, err, B);
    hipLaunchKernelGGL(k_rbig_init, dim3(max_nslab * (STM_QB_ROWS / 64), nq, nb), dim3(STM_QB_ROWS), 0, st, c, qd, Rj, W, (const double *)X, Acc,
                       (const int *)Rm, B);
    for (int k = 0; k < max_steps; k++)
        hipLaunchKernelGGL(k_rbig_step, dim3(max_nslab, nq, nb), dim3(STM_QB_ROWS), 0, st, c, qd, k, X, Acc, (const int *)Lc, (const int *)Rm, B);
    return (int)hipGetLastError();
}
int stm_launch_rtsolve(const DevCtx &c, const int *flist, int nfr, const double *Bp, double *U, double *Xr, const int *rowbase,
                       int lds_bytes, hipStream_t st, int nb, const RhsBatch &B)
{
    if (nfr <= 0) return 0;
    hipLaunchKernelGGL(k_rtsolve, dim3(nfr, nb), dim3(RS_NT), (size_t)lds_bytes, st, c, flist, Bp, U, Xr, rowbase, B);
    return (int)hipGetLastError();
}
// One panel message of a shared front (stmmqr_host.cpp: panel_msg) packed / unpacked on the device in ONE launch: six byte ranges
// (the panel's columns, T, the front's Tau / Stair / Rdead ranges, its FrontNum) between their homes and a contiguous buffer.
// Six hipMemcpyAsync cost 6 x 5-8 us of device time per panel step; this is one kernel bound by the 2-13 MB of the columns.
struct MsgSeg { char *home; long long off, bytes; };
struct MsgSegs { MsgSeg s[6]; };
__global__ __launch_bounds__(256) void k_panel_msg(MsgSegs g, char *__restrict__ buf, int out)
{
    const MsgSeg sg = g.s[blockIdx.y];
    char *a = out ? buf + sg.off : sg.home;                   // destination
    const char *b = out ? sg.home : buf + sg.off;             // source
    const long long n16 = ((((uintptr_t)a | (uintptr_t)b) & 15) == 0) ? sg.bytes >> 4 : 0;
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, nt = (long long)gridDim.x * 256;
    for (long long i = t0; i < n16; i += nt) reinterpret_cast<float4 *>(a)[i] = reinterpret_cast<const float4 *>(b)[i];
    for (long long i = (n16 << 4) + t0; i < sg.bytes; i += nt) a[i] = b[i];
}
int stm_launch_panel_msg(void *const homes[6], const long long offs[6], const long long bytes[6], void *buf, int out, hipStream_t st)
{
    MsgSegs g;
    long long mx = 0;
    for (int q = 0; q < 6; q++) { g.s[q].home = (char *)homes[q]; g.s[q].off = offs[q]; g.s[q].bytes = bytes[q]; mx = bytes[q] > mx ? bytes[q] : mx; }
    const int gx = (int)((mx / 16 + 255) / 256 < 1 ? 1 : ((mx / 16 + 255) / 256 > 1024 ? 1024 : (mx / 16 + 255) / 256));
    hipLaunchKernelGGL(k_panel_msg, dim3(gx, 6), dim3(256), 0, st, g, (char *)buf, out);
    return (int)hipGetLastError();
}
int stm_launch_perm(const double *in, const int *perm, double *out, int n, int scatter, hipStream_t st, int nb, long long sin, long long sout)
{
    if (n <= 0 || nb <= 0) return 0;
    hipLaunchKernelGGL(k_perm, dim3((n + 255) / 256, nb), dim3(256), 0, st, in, perm, out, n, scatter, sin, sout);
    return (int)hipGetLastError();
}

int stm_configure_kernels(void)
{
    // allow the panel kernels to ask for up to 144 KiB of dynamic LDS (160 KiB per CU on gfx950)
    CK(hipFuncSetAttribute((const void *)k_front_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 122880));
    CK(hipFuncSetAttribute((const void *)k_panel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_panel_pc, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_update, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_update_n, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_upd_yq, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_qapply, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_qapply_t, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_rsolve, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_rtsolve, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    return 0;
}
