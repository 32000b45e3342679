// stmmqr_pack.hip -- qr_cpack / qr_rhpack and the slab recycling's own kernels (SparseQR_factorize.c:1639-1784, 597-701):
// k_cpack, k_rh_count / scan / copy / unpack / window, k_zero_slabs, k_panel_msg.  Shared device code: stmmqr_kdev.h.
#include "stmmqr_kdev.h"


// T of the LAST panel of a front whose panel kernel left it pending (PanelDesc::t_deferred == 2: the Gram-based panel never
// builds T, and no trailing update follows the last panel).  Only the Q-apply on the resident factors reads it (DevCtx::Tall).
__device__ void dev_tlast(const DevCtx &c, int f, const FrontSym &s, FrontNum *num, double *scratch)
{
    __shared__ PanelShared ps;
    const int p = s.npanels - 1;
    if (p < 0) return;
    // a front whose schedule stopped before its last panel (FrontSym::nsched) must have run out of rows by now; if it has not,
    // the host runs the factorization again on the full schedule (abort[2]; stmmqr_host.cpp, "how many panels")
    if (s.nsched < s.npanels) {
        if (threadIdx.x == 0 && !num->done && c.abort) c.abort[2] = 1;
        return;                                                // (the last panel never ran: nothing pending of it)
    }
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
// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------
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
// One panel message of a shared front (stmmqr_multi.cpp: panel_msg) packed / unpacked on the device in ONE launch: six byte ranges
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

// ---- subtree exchange (stmmqr_multi.cpp: stmmqr_factorize_exchange): the contribution block of a front travels between plans as
// ONE message of a size known at plan time -- [8 doubles header: fm, rank, cm, csize | the front's C slot (its symbolic bound) |
// fn - fp row ids as doubles] -- so neither side asks the device what was factorized before it posts the send / the receive:
// the kernel that packs reads the front's FrontNum on the device, the kernel that unpacks writes it (stmmqr_plan_import_front's
// state: fm, rank, cm, done = 1, g = rank), the block and the row ids.  One launch packs / unpacks up to 16 fronts. ----
__global__ __launch_bounds__(256) void k_front_msg(DevCtx c, StmFrontMsgs g, double *__restrict__ buf, int out)
{
    const StmFrontMsg m = g.m[blockIdx.y];
    const FrontSym s = c.fs[m.f];
    double *b = buf + m.off;
    FrontNum *num = &c.fnum[m.f];
    const int cn = s.fn - s.fp;
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, nt = (long long)gridDim.x * 256;
    if (out) {
        const int fm = num->fm, rank = num->rank, cm = num->cm;
        const long long csize = (long long)cm * (cm + 1) / 2 + (long long)cm * (cn - cm);
        if (t0 == 0) { b[0] = fm; b[1] = rank; b[2] = cm; b[3] = (double)csize; b[4] = b[5] = b[6] = b[7] = 0.0; }
        if (cm < 0 || cm > cn || csize > m.slot) { if (t0 == 0 && c.abort) c.abort[3] = 1; return; }
        const double *C = c.Carena + s.coff;
        for (long long i = t0; i < csize; i += nt) b[8 + i] = C[i];
        const int *rows = c.Hii + s.hip + rank;
        for (long long i = t0; i < cm; i += nt) b[8 + m.slot + i] = (double)rows[i];
        return;
    }
    const int fm = (int)b[0], rank = (int)b[1], cm = (int)b[2];
    const long long csize = (long long)cm * (cm + 1) / 2 + (long long)cm * (cn - cm);
    if (cm < 0 || cm > cn || rank < 0 || rank + cm > s.fm_ub || csize > m.slot || (long long)b[3] != csize) {
        if (t0 == 0 && c.abort) c.abort[3] = 1;          // (a message that does not fit the symbolic bounds: the factorization fails)
        return;
    }
    if (t0 == 0) {
        int *w = reinterpret_cast<int *>(num);
        for (int i = 0; i < (int)(sizeof(FrontNum) / sizeof(int)); i++) w[i] = 0;
        num->fm = fm; num->rank = rank; num->cm = cm; num->done = 1; num->g = rank;
    }
    double *C = c.Carena + s.coff;
    for (long long i = t0; i < csize; i += nt) C[i] = b[8 + i];
    int *rows = c.Hii + s.hip + rank;
    for (long long i = t0; i < cm; i += nt) rows[i] = (int)b[8 + m.slot + i];
}
int stm_launch_front_msg(const DevCtx &c, const StmFrontMsgs &g, int nmsg, long long max_slot, double *buf, int out, hipStream_t st)
{
    if (nmsg <= 0) return 0;
    long long gx = (max_slot + 256 * 8 - 1) / (256 * 8);
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(k_front_msg, dim3((unsigned)gx, nmsg), dim3(256), 0, st, c, g, buf, out);
    return (int)hipGetLastError();
}

// The packed contribution block of a SHARED front is complete only in the columns of the panels a plan owns (panel q of the front
// = columns [32 q, 32 q + 32), owned by place q mod nparts): those runs, back to back, are the message of place `part` to the
// group's first rank.  Where a run starts depends on cm, which only the device knows: workgroup row y finds its run by walking the
// panels (a few hundred at most).  The host sizes the message by the symbolic bound of cm (stm_front_cols_bound).
__global__ __launch_bounds__(256) void k_front_cols(DevCtx c, int f, int part, int nparts, double *__restrict__ buf, int out)
{
    const FrontSym s = c.fs[f];
    const long long cn = s.fn - s.fp, cm = c.fnum[f].cm;
    if (cm <= 0 || cm > cn) return;
    auto coff = [&](long long j) -> long long { return j < cm ? j * (j + 1) / 2 : cm * (cm + 1) / 2 + (j - cm) * cm; };
    long long pos = 0, a = 0, e = 0;
    int own = -1;
    for (long long q = s.fp / STM_NB; q * STM_NB < s.fn; q++) {
        if (q % nparts != part) continue;
        long long j0 = q * STM_NB - s.fp, j1 = (q + 1) * STM_NB - s.fp;
        j0 = j0 < 0 ? 0 : j0; j1 = j1 > cn ? cn : j1;
        if (j1 <= j0) continue;
        a = coff(j0); e = coff(j1);
        if (++own == (int)blockIdx.y) break;
        pos += e - a;
    }
    if (own != (int)blockIdx.y) return;
    double *C = c.Carena + s.coff + a;
    double *b = buf + pos;
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, nt = (long long)gridDim.x * 256;
    if (out) for (long long i = t0; i < e - a; i += nt) b[i] = C[i];
    else for (long long i = t0; i < e - a; i += nt) C[i] = b[i];
}
int stm_launch_front_cols(const DevCtx &c, int f, int part, int nparts, int nown, long long max_run, double *buf, int out, hipStream_t st)
{
    if (nown <= 0) return 0;
    long long gx = (max_run + 256 * 8 - 1) / (256 * 8);
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(k_front_cols, dim3((unsigned)gx, nown), dim3(256), 0, st, c, f, part, nparts, buf, out);
    return (int)hipGetLastError();
}
