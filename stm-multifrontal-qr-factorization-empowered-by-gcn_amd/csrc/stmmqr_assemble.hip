// stmmqr_assemble.hip -- the start of a front of the multifrontal-QR numeric phase (STMMQR/src/qr/SparseQR_factorize.c):
//   k_amax / k_sigma <- the magnitude guard of dlarfg / dnrm2 as ONE power of two per factorization
//   k_gather_sx      <- qr_stranspose2          (:755-785)   S = A(P,Q) values, pure gather
//   k_setup          <- qr_fsize + the integer half of qr_assemble (:1066-1145, :1239-1248, :1205)
//   k_assemble       <- qr_assemble             (:1151-1285) scatter of S rows and packed child C blocks
// Design notes (DESIGN.md has the long form): fronts are column-major with a fixed leading dimension; a front's rows are known only
// on the device (dead pivot columns change them), so every kernel reads FrontNum for its extents.
#include "stmmqr_kdev.h"


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
// ------------------------------------------------------------------------------------------------
// launchers (host side calls these; no HIP types leak into the C ABI)
// ------------------------------------------------------------------------------------------------

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
