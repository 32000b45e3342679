// stmmqr_update.hip -- the trailing update of the large fronts, panel by panel (qr_larftb / dlarfb, SparseQR_factorize.c:1851-1904):
// k_update (one workgroup per column block), k_upd_w / k_upd_c (row-parallel), k_upd_f (fused), and the chain's block-0 launch with its
// riders (k_upd_b0w, k_upd_fw).  Shared device code: stmmqr_kdev.h.
#include <hip/hip_ext.h>
#include "stmmqr_kdev.h"


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
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
int stm_update_lds_bytes(void) { return (int)((2 * BN * VS + STM_NB * WS) * sizeof(double)); }
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
int stm_configure_update(void)
{
    CK(hipFuncSetAttribute((const void *)k_update, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)k_update_n, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    return 0;
}
