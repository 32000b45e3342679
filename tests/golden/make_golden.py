#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference):
    make -C oracle ref                       # gcc on the reference sources where they lie
    python tests/golden/make_golden.py       # drives oracle/_ref/refdump, writes tests/golden/*.npz

Each fixture records what crosses the hot-path seam qr_factorize()
(STMMQR/include/SparseQR.h:127-135): the matrix handed to it (after singleton removal,
SparseQR.c:146-329), tol/ntol, the whole qr_symbolic, and the reference's qr_numeric outputs.
Large cases store the packed R+H stacks as per-front sketches (norm + two weighted sums over the
uniquely determined entries, see stmmqr_testlib.determined_mask);
small cases store them in full.  Inputs of the synthetic cases are generated here from fixed seeds.
"""
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from stmmqr_testlib import determined_sketch, rrow_signature_of_block  # noqa: E402

REFDUMP = ROOT / "oracle" / "_ref" / "refdump"
REFDATA = Path("/root/reference/Data")
FULL_STACK_LIMIT = 40000       # doubles; above this only sketches are stored
VALUES_LIMIT = 100000          # nnz; above this the values are regenerated from the seeded generator


def parse_dump(path):
    """records of refdump: name[32], type char, count int64, payload -- read one array at a time (a stack is tens of GB
    at the largest size: no second copy)"""
    out = {}
    with open(path, "rb") as f:
        while True:
            head = f.read(41)
            if len(head) < 41:
                break
            name = head[:32].split(b"\0")[0].decode()
            ty = chr(head[32])
            (cnt,) = struct.unpack_from("<q", head, 33)
            dt = {"b": np.int8, "q": np.int64}.get(ty, np.float64)
            out[name] = np.fromfile(f, dt, cnt)
    return out


def write_mtx(path, m, n, rows, cols, vals):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{m} {n} {len(vals)}\n")
        for i, j, v in zip(rows, cols, vals):
            f.write(f"{i + 1} {j + 1} {v:.17g}\n")


# ---------------------------------------------------------------------------
# synthetic inputs (fixed seeds)
# ---------------------------------------------------------------------------
def _coo_from_dense(D):
    r, c = np.nonzero(D)
    return D.shape[0], D.shape[1], r, c, D[r, c]


def syn_dense6x4():
    rng = np.random.default_rng(101)
    return _coo_from_dense(rng.standard_normal((6, 4)))


def syn_wide5x8():
    rng = np.random.default_rng(102)
    D = rng.standard_normal((5, 8)) * (rng.random((5, 8)) < 0.6)
    D[0, :] += 1.0
    return _coo_from_dense(D)


def syn_dupcol():
    """30x20 sparse with columns 7 and 13 exact copies of columns 3 and 5 -> two dead pivots."""
    rng = np.random.default_rng(103)
    D = rng.standard_normal((30, 20)) * (rng.random((30, 20)) < 0.3)
    D[np.arange(20), np.arange(20)] += 3.0
    D[:, 7] = D[:, 3]
    D[:, 13] = D[:, 5]
    return _coo_from_dense(D)


def syn_emptycol():
    rng = np.random.default_rng(104)
    D = rng.standard_normal((16, 12)) * (rng.random((16, 12)) < 0.4)
    D[np.arange(12), np.arange(12)] += 2.0
    D[:, 4] = 0.0
    return _coo_from_dense(D)


def syn_chain():
    """lower-bidiagonal + a few far entries: chain-shaped tree."""
    n = 40
    rng = np.random.default_rng(105)
    D = np.zeros((n + 3, n))
    for j in range(n):
        D[j, j] = 2.0 + rng.random()
        D[j + 1, j] = -1.0 + 0.1 * rng.standard_normal()
        D[j + 3, j] = 0.5 * rng.standard_normal()
    return _coo_from_dense(D)


def syn_star():
    """arrow matrix: many leaves joined by one dense last column/row block."""
    n = 48
    rng = np.random.default_rng(106)
    D = np.zeros((n + 6, n))
    for j in range(n - 4):
        D[j, j] = 3.0 + rng.random()
        D[j, n - 4:] = 0.3 * rng.standard_normal(4)
    D[n - 4:, n - 4:] = rng.standard_normal((10, 4))
    D[n - 4:, : n - 4] = 0.2 * rng.standard_normal((10, n - 4)) * (rng.random((10, n - 4)) < 0.3)
    return _coo_from_dense(D)


def syn_rand60x40():
    rng = np.random.default_rng(107)
    D = rng.standard_normal((60, 40)) * (rng.random((60, 40)) < 0.12)
    D[np.arange(40), np.arange(40)] += 2.0
    return _coo_from_dense(D)


def _grid(dims, seed, stencil_full):
    rng = np.random.default_rng(seed)
    import itertools
    idx = np.arange(int(np.prod(dims))).reshape(dims)
    rows, cols, vals = [], [], []
    offs = [o for o in itertools.product(*[(-1, 0, 1)] * len(dims))
            if stencil_full or sum(abs(x) for x in o) <= 1]
    for p in itertools.product(*[range(d) for d in dims]):
        i = idx[p]
        for o in offs:
            q = tuple(a + b for a, b in zip(p, o))
            if all(0 <= a < d for a, d in zip(q, dims)):
                j = idx[q]
                v = rng.uniform(-1, 1)
                if i == j:
                    v += len(offs)
                rows.append(i); cols.append(j); vals.append(v)
    n = idx.size
    return n, n, np.array(rows), np.array(cols), np.array(vals)


def syn_grid2d():
    return _grid((14, 14), 108, False)


def syn_grid3d():
    return _grid((7, 7, 7), 109, True)


def syn_rankdef_grid():
    """2-D grid operator with three duplicated columns and one zero column (rank detection inside a real tree)."""
    m, n, r, c, v = _grid((12, 12), 110, False)
    D = np.zeros((m, n)); D[r, c] = v
    D[:, 50] = D[:, 20]; D[:, 90] = D[:, 21]; D[:, 130] = 2.0 * D[:, 77]; D[:, 5] = 0
    return _coo_from_dense(D)


SYNTH = {
    "syn_dense6x4": syn_dense6x4, "syn_wide5x8": syn_wide5x8, "syn_dupcol": syn_dupcol,
    "syn_emptycol": syn_emptycol, "syn_chain": syn_chain, "syn_star": syn_star,
    "syn_rand60x40": syn_rand60x40, "syn_grid2d": syn_grid2d, "syn_grid3d": syn_grid3d,
    "syn_rankdef_grid": syn_rankdef_grid,
}
# (fixture name, source, ordering, tolmode)
# every matrix of the reference's own test list (STMMQR/test.txt:1-16) that is present under Data/ (9 of 16; the other seven are
# in .MISSING_LARGE_BLOBS and have stand-ins, gen3d.py).  Their values cannot be regenerated, so the fixture keeps the matrix.
REAL = [("bcsstk14", "bcsstk14.mtx", -1, "d"), ("epb1", "epb1.mtx", -1, "d"),
        ("lns_3937", "lns_3937.mtx", -1, "d"),
        ("dwt_992", "dwt_992.mtx", -1, "d"), ("reorientation_8", "reorientation_8.mtx", -1, "d"),
        ("cvxqp3", "cvxqp3.mtx", -1, "d"), ("t2d_q9", "t2d_q9.mtx", -1, "d"), ("bayer10", "bayer10.mtx", -1, "d"),
        ("ex18", "ex18.mtx", -1, "d")]


def compact(name, d, keep_values=False):
    """dump dict -> fixture dict (ints to int32 where they fit, stacks -> sketches when large)."""
    out = {}
    ns = int(d["num_ns"][0])
    assert ns == 1, "fixtures are generated with SPQR_grain = 1 (one stack)"
    stack = d["num_Stack_0"]
    nf = int(d["sym_nf"][0])
    post = d["sym_Post"][:nf]
    offs = d["num_Rblock_off"]
    order_offs = offs[post]
    ends = np.append(order_offs[1:], len(stack))
    sk = np.zeros((nf, 3)); rsize = np.zeros(nf, np.int64); sigs = {}
    Rp, Super, Hm, HStair = d["sym_Rp"], d["sym_Super"], d["num_Hm"], d["num_HStair"]
    for f, a, b in zip(post, order_offs, ends):
        fn, fp = Rp[f + 1] - Rp[f], Super[f + 1] - Super[f]
        sk[f] = determined_sketch(stack[a:b], HStair[Rp[f]:Rp[f + 1]], fp, fn, Hm[f])
        rsize[f] = b - a
        sigs[f] = rrow_signature_of_block(stack[a:b], HStair[Rp[f]:Rp[f + 1]], fp, fn, Hm[f])
    out["num_rrow_sig"] = np.concatenate([sigs[f] for f in range(nf)]) if nf else np.zeros((0, 3))
    out["num_rh_sketch"] = sk
    out["num_rh_size"] = rsize
    big = d["in_Ax"].size > VALUES_LIMIT and not keep_values
    for k, a in d.items():
        if k.startswith("num_Stack_"):
            if len(a) <= FULL_STACK_LIMIT:
                out["num_Stack"] = a
            continue
        if big and k in ("A_p", "A_i", "A_x", "in_Ap", "in_Ai", "in_Ax", "num_HTau", "solve_x"):
            continue                      # stand-in: pattern + values come from tests/golden/gen3d.py
        if a.dtype == np.int64 and a.size and np.abs(a).max() < 2**31:
            a = a.astype(np.int32)
        out[k] = a
    return out


def run(name, mtx, ordering, tolmode, keep_values=False):
    with tempfile.TemporaryDirectory() as td:
        binp = Path(td) / "dump.bin"
        env = {"MKL_THREADING_LAYER": "SEQUENTIAL", "PATH": "/usr/bin:/bin"}
        r = subprocess.run([str(REFDUMP), str(mtx), str(ordering), "1", tolmode, str(binp), "1"],
                           capture_output=True, text=True, env=env)
        if r.returncode != 0:
            raise RuntimeError(f"{name}: refdump failed\n{r.stdout}\n{r.stderr}")
        d = parse_dump(binp)
    fx = compact(name, d, keep_values)
    np.savez_compressed(HERE / f"{name}.npz", **fx)
    sz = (HERE / f"{name}.npz").stat().st_size
    print(f"{name:18s} m={d['in_m'][0]:6d} n={d['in_n'][0]:6d} nf={d['sym_nf'][0]:5d} rank={d['num_rank'][0]:6d} "
          f"flops={d['flopcount'][0]:.4g} res={d['res'][0]:.1e} -> {sz / 1024:.0f} KiB")


def run_standin(name):
    from gen3d import STANDINS, standin_matrix
    m, n, Ap, Ai, Ax = standin_matrix(name)
    with tempfile.TemporaryDirectory() as td:
        p = Path(td) / f"{name}.mtx"
        cols = np.repeat(np.arange(n), np.diff(Ap))
        with open(p, "w") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, len(Ax)))
            np.savetxt(f, np.c_[Ai + 1, cols + 1, Ax], fmt="%d %d %.17g")
        run(name, p, STANDINS[name][5], "d")


def main():
    if not REFDUMP.exists():
        sys.exit("build the reference first: make -C oracle ref")
    only = set(sys.argv[1:])
    from gen3d import STANDINS
    for name in STANDINS:
        # (c5_standin: 75 minutes and 21 GB of host memory -- only when it is asked for by name)
        if name in only or (not only and name != "c5_standin" and not (HERE / f"{name}.npz").exists()):
            run_standin(name)
    if only and only <= set(STANDINS):
        return
    with tempfile.TemporaryDirectory() as td:
        for name, gen in SYNTH.items():
            if only and name not in only:
                continue
            m, n, r, c, v = gen()
            p = Path(td) / f"{name}.mtx"
            write_mtx(p, m, n, r, c, v)
            run(name, p, -1, "d")
    for name, fn, ordering, tolmode in REAL:
        if only and name not in only:
            continue
        run(name, REFDATA / fn, ordering, tolmode, keep_values=True)


if __name__ == "__main__":
    main()
