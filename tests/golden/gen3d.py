"""Deterministic 3-D grid stand-ins for the SuiteSparse matrices that are absent from the reference checkout
(/root/reference/.MISSING_LARGE_BLOBS: xenon1, sme3Dc, 3D_51448_3D).  SURVEY.md 8(d) fixes the generators:

  xenon1 stand-in  : gen3d(36, 36, 38, stencil 27pt, dof 1, values U(-1,1), diag += 27, seed 0x58454E31)  n = 49 248
  sme3Dc stand-in  : gen3d(24, 24, 25, stencil 27pt, dof 3, seed 0x534D4533)                               n = 43 200

Every report that uses them says "stand-in".  Pure numpy; used by make_golden.py (build container) and by bench.py /
tests on the GPU box to regenerate the VALUES (the symbolic analysis of the pattern is a committed fixture).
"""
import numpy as np


def gen3d(nx, ny, nz, seed, stencil27=True, dof=1, longrange=0):
    """CSC (Ap, Ai, Ax) of an unsymmetric-valued 7/27-point operator on an nx*ny*nz grid, `dof` unknowns per grid point
    (every pair of neighbouring points couples all their unknowns: dof x dof blocks); longrange > 0 adds that many
    random couplings per grid point to points anywhere in the grid (no small separators: the top fronts of the
    frontal tree become a large fraction of n -- the structure of SURVEY.md 8(d)'s stand-in for 3D_51448_3D)."""
    rng = np.random.default_rng(seed)
    idx = np.arange(nx * ny * nz).reshape(nx, ny, nz)
    rows, cols = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if not stencil27 and abs(dx) + abs(dy) + abs(dz) > 1:
                    continue
                xs = slice(max(0, -dx), nx - max(0, dx)); xt = slice(max(0, dx), nx - max(0, -dx))
                ys = slice(max(0, -dy), ny - max(0, dy)); yt = slice(max(0, dy), ny - max(0, -dy))
                zs = slice(max(0, -dz), nz - max(0, dz)); zt = slice(max(0, dz), nz - max(0, -dz))
                rows.append(idx[xs, ys, zs].ravel()); cols.append(idx[xt, yt, zt].ravel())
    r = np.concatenate(rows); c = np.concatenate(cols)
    if longrange > 0:
        npts = idx.size
        r2 = np.repeat(np.arange(npts), longrange)
        c2 = rng.integers(0, npts, npts * longrange)
        r = np.concatenate([r, r2]); c = np.concatenate([c, c2])
        _, first = np.unique(c * npts + r, return_index=True)      # (a random coupling may repeat a stencil entry)
        first.sort()
        r, c = r[first], c[first]
    if dof > 1:
        a, b = np.meshgrid(np.arange(dof), np.arange(dof), indexing="ij")
        r = (r[:, None] * dof + a.ravel()[None, :]).ravel()
        c = (c[:, None] * dof + b.ravel()[None, :]).ravel()
    v = rng.uniform(-1, 1, r.size)
    v[r == c] += (27 if stencil27 else 7) * dof
    n = idx.size * dof
    o = np.lexsort((r, c))
    r, c, v = r[o], c[o], v[o]
    Ap = np.zeros(n + 1, np.int64)
    np.add.at(Ap, c + 1, 1)
    Ap = np.cumsum(Ap)
    return n, n, Ap, r.astype(np.int64), v


STANDINS = {
    # name: (nx, ny, nz, seed, 27-point?, ordering for the reference run: 2 = METIS)
    "xenon1_standin": (36, 36, 38, 0x58454E31, True, 2),
    # the same matrix with the driver's DEFAULT ordering (qrtest without an ordering argument = COLAMD, qrtest.c:155-169):
    # the ordering BASELINE.md's published totals were measured with
    "xenon1_colamd_standin": (36, 36, 38, 0x58454E31, True, -1),
    "grid20_standin": (20, 20, 20, 0x58454E31, True, 2),
    # BASELINE configs[3] (sme3Dc.mtx, absent): 3 unknowns per grid point, ~81 nnz per row
    "sme3dc_standin": (24, 24, 25, 0x534D4533, True, 2, 3),
    # the STRUCTURE of BASELINE configs[4] (3D_51448_3D.mtx, absent; SURVEY.md 8d: 7-point + random long-range couplings,
    # 13 nnz per row) at 1/6.5 of its size: n = 8000, top front 0.4 n, 4.4e11 flops (the full size, n = 52 022, is
    # ~1.2e14 flops and ~20 GB of factors: 47 minutes for the reference on one core)
    "c5mini_standin": (20, 20, 20, 0x33445F35, False, 2, 1, 6),
    # ... at half the full size: n = 27 000, root front 27 000 x 25 974
    "c5mid_standin": (30, 30, 30, 0x33445F35, False, 2, 1, 6),
    # ... and at FULL size (gen3d(37, 37, 38), n = 52 022): 1.156e14 flops, root front 52 022 x 49 959
    # (2.6e9 entries), 17.6 GB of packed factors; 41 minutes for the reference on one core.  Generated only on request
    # (make_golden.py c5_standin: ~75 minutes, 21 GB of host memory)
    "c5_standin": (37, 37, 38, 0x33445F35, False, 2, 1, 6),
}


def standin_matrix(name):
    nx, ny, nz, seed, s27, _ = STANDINS[name][:6]
    dof = STANDINS[name][6] if len(STANDINS[name]) > 6 else 1
    longrange = STANDINS[name][7] if len(STANDINS[name]) > 7 else 0
    return gen3d(nx, ny, nz, seed, s27, dof, longrange)
