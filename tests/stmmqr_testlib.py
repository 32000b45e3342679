"""Shared test helpers (TEST INFRASTRUCTURE).

* ``Oracle``      -- ctypes binding of oracle/libstmmqr_oracle.so, the CPU restatement of
                     STMMQR/src/qr/SparseQR_factorize.c (see oracle/stmmqr_oracle.h).
* ``load_golden`` -- committed fixtures tests/golden/*.npz, produced from the REAL reference by
                     tests/golden/make_golden.py (oracle/_ref/refdump).
* sketches / comparison helpers shared by the CPU and GPU parity tests.

Nothing here reads /root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
ORACLE_DIR = ROOT / "oracle"

I64 = np.int64
c_long_p = C.POINTER(C.c_long)
c_double_p = C.POINTER(C.c_double)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_long_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


# --------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------
SYM_ARRAYS = ["Sp", "Sj", "Qfill", "PLinv", "Sleft", "Parent", "Child", "Childp", "Super", "Rp", "Rj",
              "Post", "Hip", "Fm", "Cm"]
SYM_SCALARS = ["m", "n", "anz", "nf", "maxfn", "rjsize", "do_rank_detection", "maxstack", "hisize", "keepH",
               "ntasks", "ns"]


BIG_FIXTURES = ("xenon1_standin", "xenon1_colamd_standin", "sme3dc_standin", "c5mini_standin", "c5mid_standin", "c5_standin")      # full BASELINE size: too slow for the scalar CPU oracle, GPU tests only


# The reference's own test list (STMMQR/test.txt:1-16) as far as its files are present under /root/reference/Data: 9 of 16.
REFERENCE_TEST_MATRICES = ("dwt_992", "lns_3937", "bcsstk14", "epb1", "reorientation_8", "cvxqp3", "t2d_q9", "bayer10", "ex18")
# ... two of them are too heavy for the scalar CPU oracle inside every parametrized test (2e10 / 1.9e11 flops: 7 s / 100 s per oracle
# run): they have their own tests (tests/test_oracle_golden.py once each on the CPU, tests/test_gpu_factorize.py::test_reference_inputs)
HEAVY_REAL = ("reorientation_8", "cvxqp3")
# ---- tolerances of floating-point comparisons on ill-conditioned inputs: DERIVED from the factor, one constant for every input ----
# Two correct QR factorizations of A differ in R by about eps * cond(A) * ||R|| (rows whose pivot sits near tol are only determined
# relative to ||A||: the reference's own driver prints res = 1e0 ... 1e3 for five of its nine test matrices), and a triangular solve
# amplifies the rounding differences of two correct Q'b by cond(R).  cond(R) is estimated from the packed factor under test by a
# few random probes (cond_probe: a LOWER estimate, typically within sqrt(n) of cond_2), and every comparison below is allowed
# TOL_C * eps * cond_probe -- the same constant for bcsstk14 (cond 1e7) and ex18 (cond 1e12); no per-matrix table.
# (Rounds 1-4 kept two hand-set tables here, "measured differences x 100": a tolerance chosen from the observed difference cannot fail.)
EPS = float(np.finfo(np.float64).eps)
TOL_C = 64.0     # (largest observed difference / (eps * cond_probe) over the fixtures: 5.1 -- bcsstk14, whose probes underestimate its condition most)


def cond_probe(orc, S, N, nprobe=3, seed=7):
    """Lower estimate of cond_2(R) of the packed factor N (oracle-style Numeric): (max ||R z|| / ||z||) * (max ||R^-1 y|| / ||y||) over
    a few random vectors on the live columns / rows; each factor is a lower bound of the norm it probes."""
    rng = np.random.default_rng(seed)
    n, rank = S.n, int(N.c.rank)
    if rank == 0:
        return 1.0
    dead = np.asarray(N.Rdead[:n]) != 0
    nr = ni = 0.0
    for _ in range(nprobe):
        z = rng.standard_normal(n)
        z[dead] = 0.0
        nr = max(nr, float(np.linalg.norm(orc.rmult(S, N, z)) / max(np.linalg.norm(z), 1e-300)))
        y = np.zeros(S.m)
        y[:rank] = rng.standard_normal(rank)
        ni = max(ni, float(np.linalg.norm(orc.rsolve(S, N, y)) / np.linalg.norm(y)))
    return max(nr * ni, 1.0)


def solve_tol(kappa, floor=1e-9):
    """allowed ||x - x_ref|| / ||x_ref|| between two correct solutions through factors of condition `kappa`"""
    return max(floor, TOL_C * EPS * kappa)


def rrow_excess(got, ref, kappa, rel=1e-10):
    """R-row signatures (|diag|, ||row||, |<row, w>|) of two factorizations: the largest |difference| in units of what is allowed --
    `rel` of the row's own norm, or TOL_C * eps * kappa of the LARGEST row norm, whichever is larger.  <= 1 passes."""
    if ref.size == 0:
        return 0.0
    allowed = np.maximum(rel * np.maximum(ref[:, 1:2], 1e-300), TOL_C * EPS * kappa * np.max(ref[:, 1], initial=1e-300))
    return float(np.max(np.abs(got - ref) / allowed, initial=0.0))


def golden_names(include_big: bool = False):
    return sorted(p.stem for p in GOLDEN.glob("*.npz") if include_big or (p.stem not in BIG_FIXTURES and p.stem not in HEAVY_REAL))


def load_golden(name: str) -> dict:
    """Return a dict of numpy arrays; integer arrays are widened to int64."""
    z = np.load(GOLDEN / f"{name}.npz")
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype.kind in "iu" and a.dtype != np.int8:
            a = a.astype(I64)
        out[k] = np.ascontiguousarray(a)
    if "in_Ax" not in out:
        # large stand-in: pattern and values come from the seeded generator (tests/golden/gen3d.py)
        import sys
        sys.path.insert(0, str(GOLDEN))
        from gen3d import standin_matrix
        m, n, Ap, Ai, Ax = standin_matrix(name)
        assert m == scalar(out, "in_m") and n == scalar(out, "in_n") and Ax.size == scalar(out, "sym_anz")
        out["in_Ap"], out["in_Ai"], out["in_Ax"] = Ap, Ai, Ax
    return out


def scalar(g: dict, key: str):
    return g[key].reshape(-1)[0].item()


def sketch_weights(n: int):
    i = np.arange(n, dtype=np.float64)
    return np.sin(0.7 * i + 0.3), np.cos(1.3 * i + 0.1)


def block_sketch(v: np.ndarray) -> np.ndarray:
    """(||v||_2, <v,w1>, <v,w2>) -- order-sensitive fingerprint of one packed block."""
    if v.size == 0:
        return np.zeros(3)
    if v.size <= (1 << 24):
        w1, w2 = sketch_weights(v.size)
        return np.array([np.linalg.norm(v), float(v @ w1), float(v @ w2)])
    # very large blocks: the same sums in pieces (no index / weight arrays of the block's size)
    ss = d1 = d2 = 0.0
    for a in range(0, v.size, 1 << 24):
        x = v[a:a + (1 << 24)]
        i = np.arange(a, a + x.size, dtype=np.float64)
        ss += float(x @ x); d1 += float(x @ np.sin(0.7 * i + 0.3)); d2 += float(x @ np.cos(1.3 * i + 0.1))
    return np.array([np.sqrt(ss), d1, d2])


def determined_mask(Stair, fp: int, fn: int, fm: int) -> np.ndarray:
    """Boolean mask over one packed R+H block (qr_rhpack layout, SparseQR_factorize.c:1691-1784) selecting
    the entries that are uniquely determined by A: everything in the pivotal columns (R and H) and the R rows
    of the non-pivotal columns.  The Householder vectors of NON-pivotal columns only rotate the contribution
    block; where the symbolic structure over-estimates the numerical rank they are built from rounding noise
    and legitimately differ between two correct implementations, so they are checked functionally
    (A = QR, Q'Q = I) instead of element by element."""
    parts = []
    rm = 0
    for k in range(fp):
        t = int(Stair[k])
        if t == 0:
            parts.append(np.ones(rm, bool))
        else:
            if rm < fm:
                rm += 1
            parts.append(np.ones(t, bool))
    h = rm
    for k in range(fp, fn):
        t = int(Stair[k])
        h = min(h + 1, fm)
        parts.append(np.ones(rm, bool))
        parts.append(np.zeros(max(t - h, 0), bool))
    return np.concatenate(parts) if parts else np.zeros(0, bool)


def front_R(block, Stair, fp: int, fn: int, fm: int) -> np.ndarray:
    """Dense rm-by-fn R part of one packed R+H block (rows = live pivots of the front)."""
    cols = []
    rm = 0
    p = 0
    for k in range(fp):
        t = int(Stair[k])
        if t == 0:
            cols.append((p, rm)); p += rm
        else:
            if rm < fm:
                rm += 1
            cols.append((p, rm)); p += t
    h = rm
    for k in range(fp, fn):
        t = int(Stair[k])
        h = min(h + 1, fm)
        cols.append((p, rm)); p += rm + max(t - h, 0)
    assert p == block.size, (p, block.size)
    R = np.zeros((rm, fn))
    for k, (a, r) in enumerate(cols):
        R[:r, k] = block[a:a + r]
    return R


def rrow_signature(R: np.ndarray) -> np.ndarray:
    """Per R row: (|first nonzero-position entry| i.e. |diagonal|, ||row||, |<row, w>|).  Invariant under the
    row sign flips that are the only freedom of R = chol(A'A) (sign of R_kk = -sign(alpha), and alpha depends
    on how the rounding-noise rows of a child's contribution block happened to be rotated)."""
    rm, fn = R.shape
    if rm == 0:
        return np.zeros((0, 3))
    w1, _ = sketch_weights(fn)
    # the diagonal of row i is its first structurally stored entry = largest-|.| leading entry position:
    lead = np.array([np.flatnonzero(R[i])[0] if np.any(R[i]) else 0 for i in range(rm)])
    return np.stack([np.abs(R[np.arange(rm), lead]), np.linalg.norm(R, axis=1), np.abs(R @ w1)], axis=1)


HUGE_BLOCK = 1 << 27          # entries of a packed R+H block from which the streaming forms below are used


def rrow_signature_of_block(block, Stair, fp: int, fn: int, fm: int) -> np.ndarray:
    """rrow_signature(front_R(...)) without the dense rm x fn matrix: the columns of the packed block are walked once and
    per-row accumulators kept (a 50 000 x 50 000 front would need 20 GB dense).  Same definition: |first structurally
    stored nonzero of the row|, ||row||, |<row, w1>|; the sums run over the columns in order."""
    if block.size < HUGE_BLOCK:
        return rrow_signature(front_R(block, Stair, fp, fn, fm))
    w1, _ = sketch_weights(fn)
    lead = None
    rm = 0
    p = 0
    cols = []
    for k in range(fp):
        t = int(Stair[k])
        if t == 0:
            cols.append((p, rm)); p += rm
        else:
            if rm < fm:
                rm += 1
            cols.append((p, rm)); p += t
    h = rm
    for k in range(fp, fn):
        t = int(Stair[k])
        h = min(h + 1, fm)
        cols.append((p, rm)); p += rm + max(t - h, 0)
    assert p == block.size, (p, block.size)
    if rm == 0:
        return np.zeros((0, 3))
    lead = np.zeros(rm); have = np.zeros(rm, bool); ss = np.zeros(rm); dot = np.zeros(rm)
    for k, (a, r) in enumerate(cols):
        if r == 0:
            continue
        x = block[a:a + r]
        ss[:r] += x * x
        dot[:r] += x * w1[k]
        new = (~have[:r]) & (x != 0)
        if new.any():
            idx = np.flatnonzero(new)
            lead[idx] = x[idx]; have[idx] = True
    return np.stack([np.abs(lead), np.sqrt(ss), np.abs(dot)], axis=1)


def determined_sketch(block, Stair, fp: int, fn: int, fm: int) -> np.ndarray:
    """block_sketch(determined_part(...)) without the mask / the copy for very large blocks (same sums, column by column)."""
    if block.size < HUGE_BLOCK:
        return block_sketch(determined_part(block, Stair, fp, fn, fm))
    ss = d1 = d2 = 0.0
    pos = 0                       # index among the determined entries
    p = 0
    rm = 0

    def take(x):
        nonlocal ss, d1, d2, pos
        if x.size:
            i = np.arange(pos, pos + x.size, dtype=np.float64)
            ss += float(x @ x); d1 += float(x @ np.sin(0.7 * i + 0.3)); d2 += float(x @ np.cos(1.3 * i + 0.1))
            pos += x.size
    for k in range(fp):
        t = int(Stair[k])
        if t == 0:
            take(block[p:p + rm]); p += rm
        else:
            if rm < fm:
                rm += 1
            take(block[p:p + t]); p += t
    h = rm
    for k in range(fp, fn):
        t = int(Stair[k])
        h = min(h + 1, fm)
        take(block[p:p + rm]); p += rm + max(t - h, 0)
    assert p == block.size, (p, block.size)
    return np.array([np.sqrt(ss), d1, d2])


def determined_part(block, Stair, fp, fn, fm):
    mk = determined_mask(Stair, fp, fn, fm)
    assert mk.size == block.size, (mk.size, block.size)
    return block[mk]


# --------------------------------------------------------------------------------------
# oracle binding
# --------------------------------------------------------------------------------------
class OrcSymbolic(C.Structure):
    _fields_ = [(k, C.c_long) for k in
                ["m", "n", "anz", "nf", "maxfn", "rjsize", "hisize", "maxstack", "do_rank_detection"]] + \
               [(k, c_long_p) for k in
                ["Sp", "Sj", "Qfill", "PLinv", "Sleft", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip"]]


class OrcNumeric(C.Structure):
    _fields_ = [("Stack", c_double_p), ("Rblock_off", c_long_p), ("Rdead", C.c_char_p),
                ("HStair", c_long_p), ("HTau", c_double_p), ("Hii", c_long_p), ("HPinv", c_long_p),
                ("Hm", c_long_p), ("Hr", c_long_p), ("Cm", c_long_p),
                ("rank", C.c_long), ("rank1", C.c_long), ("maxfrank", C.c_long), ("maxfm", C.c_long),
                ("rh_total", C.c_long), ("flopcount", C.c_double),
                ("Csave", c_double_p), ("Csave_off", c_long_p),
                ("t_assemble", C.c_double), ("t_front", C.c_double), ("t_pack", C.c_double)]


class OrcChunk(C.Structure):
    _fields_ = [("fchunk", C.c_long), ("small", C.c_long), ("minchunk", C.c_long), ("minchunk_ratio", C.c_long)]


def build_oracle(force: bool = False) -> Path:
    so = ORACLE_DIR / "libstmmqr_oracle.so"
    src = ORACLE_DIR / "stmmqr_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(ORACLE_DIR), "port"], stdout=subprocess.DEVNULL)
    return so


class Symbolic:
    """Owns the numpy arrays behind an OrcSymbolic."""

    def __init__(self, g: dict, prefix: str = "sym_"):
        self.arr = {k: np.ascontiguousarray(g[prefix + k], dtype=I64) for k in SYM_ARRAYS if prefix + k in g}
        self.sc = {k: int(scalar(g, prefix + k)) for k in SYM_SCALARS if prefix + k in g}
        if self.arr.get("Qfill") is not None and self.arr["Qfill"].size == 0:
            self.arr["Qfill"] = None
        s = OrcSymbolic()
        for k in ["m", "n", "anz", "nf", "maxfn", "rjsize", "hisize", "maxstack", "do_rank_detection"]:
            setattr(s, k, self.sc[k])
        for k in ["Sp", "Sj", "Qfill", "PLinv", "Sleft", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip"]:
            setattr(s, k, _ip(self.arr.get(k)))
        self.c = s

    def __getattr__(self, k):
        if k in ("arr", "sc", "c"):
            raise AttributeError(k)
        if k in self.sc:
            return self.sc[k]
        if k in self.arr:
            return self.arr[k]
        raise AttributeError(k)


class Numeric:
    """Caller-allocated outputs of orc_factorize."""

    def __init__(self, S: Symbolic, save_c: bool = False):
        nf, n, m = S.nf, S.n, S.m
        self.Stack = np.zeros(max(S.maxstack, 1))
        self.Rblock_off = np.zeros(max(nf, 1), I64)
        self.Rdead = np.zeros(max(n, 1), np.int8)
        self.HStair = np.zeros(max(S.rjsize, 1), I64)
        self.HTau = np.zeros(max(S.rjsize, 1))
        self.Hii = np.zeros(max(S.hisize, 1), I64)
        self.HPinv = np.zeros(max(m, 1), I64)
        self.Hm = np.zeros(max(nf, 1), I64)
        self.Hr = np.zeros(max(nf, 1), I64)
        self.Cm = np.zeros(max(nf, 1), I64)
        self.Csave = np.zeros(max(S.maxstack, 1)) if save_c else None
        self.Csave_off = np.zeros(max(nf, 1), I64) if save_c else None
        c = OrcNumeric()
        c.Stack = _dp(self.Stack); c.Rblock_off = _ip(self.Rblock_off)
        c.Rdead = C.cast(self.Rdead.ctypes.data, C.c_char_p)
        c.HStair = _ip(self.HStair); c.HTau = _dp(self.HTau); c.Hii = _ip(self.Hii)
        c.HPinv = _ip(self.HPinv); c.Hm = _ip(self.Hm); c.Hr = _ip(self.Hr); c.Cm = _ip(self.Cm)
        c.Csave = _dp(self.Csave); c.Csave_off = _ip(self.Csave_off)
        self.c = c

    # packed R+H block of front f (needs rsize: distance to the next block in postorder)
    def rh_blocks(self, S: Symbolic):
        order = S.Post[: S.nf]
        offs = self.Rblock_off[order]
        ends = np.append(offs[1:], self.c.rh_total)
        out = {}
        for f, a, b in zip(order, offs, ends):
            out[int(f)] = self.Stack[a:b]
        return out


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(str(build_oracle()))
        L = self.lib
        L.orc_factorize.restype = C.c_int
        L.orc_factorize.argtypes = [C.POINTER(OrcSymbolic), c_long_p, c_long_p, c_double_p, C.c_double, C.c_long,
                                    C.POINTER(OrcChunk), C.POINTER(OrcNumeric)]
        L.orc_front.restype = C.c_long
        L.orc_front.argtypes = [C.c_long, C.c_long, C.c_long, C.c_double, C.c_long, C.POINTER(OrcChunk),
                                c_double_p, c_long_p, C.c_char_p, c_double_p, c_double_p, c_double_p]
        L.orc_larfg.restype = C.c_double
        L.orc_larfg.argtypes = [C.c_long, c_double_p, c_double_p]
        L.orc_larftb.restype = None
        L.orc_larftb.argtypes = [C.c_int, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, c_double_p, c_double_p,
                                 c_double_p, c_double_p]
        L.orc_cpack.restype = C.c_long
        L.orc_cpack.argtypes = [C.c_long] * 4 + [c_double_p, c_double_p]
        L.orc_rhpack.restype = C.c_long
        L.orc_rhpack.argtypes = [C.c_long] * 3 + [c_long_p, c_double_p, c_double_p, c_long_p]
        L.orc_fcsize.restype = C.c_long
        L.orc_fcsize.argtypes = [C.c_long] * 4
        L.orc_qmult.restype = None
        L.orc_qmult.argtypes = [C.c_int, C.POINTER(OrcSymbolic), C.POINTER(OrcNumeric), c_double_p, c_double_p]
        L.orc_rmult.restype = None
        L.orc_rmult.argtypes = [C.POINTER(OrcSymbolic), C.POINTER(OrcNumeric), c_double_p, c_double_p]
        L.orc_rsolve.restype = C.c_int
        L.orc_rsolve.argtypes = [C.POINTER(OrcSymbolic), C.POINTER(OrcNumeric), c_double_p, c_double_p]
        L.orc_stranspose2.restype = None
        L.orc_stranspose2.argtypes = [C.c_long, C.c_long, c_long_p, c_long_p, c_double_p, c_long_p, c_long_p,
                                      c_long_p, c_double_p, c_long_p]

    @staticmethod
    def chunk(fchunk=32, small=5000, minchunk=4, ratio=4):
        return OrcChunk(fchunk, small, minchunk, ratio)

    def factorize(self, S: Symbolic, Ap, Ai, Ax, tol, ntol, chunk=None, save_c=False) -> Numeric:
        N = Numeric(S, save_c)
        ch = chunk or self.chunk()
        Ap = np.ascontiguousarray(Ap, I64); Ai = np.ascontiguousarray(Ai, I64)
        Ax = np.ascontiguousarray(Ax, np.float64)
        rc = self.lib.orc_factorize(C.byref(S.c), _ip(Ap), _ip(Ai), _dp(Ax), float(tol), int(ntol), C.byref(ch),
                                    C.byref(N.c))
        if rc != 0:
            raise RuntimeError(f"orc_factorize failed: {rc}")
        return N

    def front(self, F, Stair, npiv, tol, ntol, chunk=None):
        """F: (m,n) Fortran-ordered float64, modified in place. Returns (rank, Tau, Rdead, flops)."""
        m, n = F.shape
        assert F.flags.f_contiguous
        ch = chunk or self.chunk()
        Tau = np.zeros(max(n, 1)); Rdead = np.zeros(max(npiv, 1), np.int8)
        W = np.zeros(max(ch.fchunk, 1) * max(n, 1) + 64)
        fl = C.c_double(0)
        r = self.lib.orc_front(m, n, npiv, float(tol), int(ntol), C.byref(ch), _dp(F), _ip(Stair),
                               C.cast(Rdead.ctypes.data, C.c_char_p), _dp(Tau), _dp(W), C.byref(fl))
        return int(r), Tau[:n], Rdead[:npiv], fl.value

    # ---- checkers ----
    def qmult(self, method, S: Symbolic, N: Numeric, x):
        x = np.array(x, dtype=np.float64, copy=True)
        w = np.zeros(S.m)
        self.lib.orc_qmult(method, C.byref(S.c), C.byref(N.c), _dp(x), _dp(w))
        return x

    def rmult(self, S: Symbolic, N: Numeric, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.zeros(S.m)
        self.lib.orc_rmult(C.byref(S.c), C.byref(N.c), _dp(x), _dp(y))
        return y

    def rsolve(self, S: Symbolic, N: Numeric, y):
        y = np.ascontiguousarray(y, np.float64)
        x = np.zeros(S.n)
        rc = self.lib.orc_rsolve(C.byref(S.c), C.byref(N.c), _dp(y), _dp(x))
        if rc != 0:
            raise RuntimeError("rank deficient")
        return x


# --------------------------------------------------------------------------------------
# residual checks on a factorization held as (Symbolic, Numeric-like)
# --------------------------------------------------------------------------------------
def csc_matvec(m, Ap, Ai, Ax, x):
    y = np.zeros(m)
    n = len(Ap) - 1
    cols = np.repeat(np.arange(n), np.diff(Ap))
    np.add.at(y, Ai, Ax * x[cols])
    return y


def aqr_probe_error(orc: Oracle, S: Symbolic, N: Numeric, Ap, Ai, Ax, nprobe=4, seed=7, live_only=False):
    """max over random x of ||A*E*x - Q*(R*x)|| / (||A||_F ||x||)   (SURVEY.md 8d parity metric).
    live_only: x is zero on the dead pivot columns (Rdead) -- on a rank-deficient factorization A E = Q R holds on the
    live columns to working accuracy (a dead column is dropped, it generates no reflector), on the dead ones only to tol."""
    rng = np.random.default_rng(seed)
    m, n = S.m, S.n
    q = S.Qfill if S.Qfill is not None else np.arange(n)
    anorm = np.linalg.norm(Ax) or 1.0
    worst = 0.0
    for _ in range(nprobe):
        x = rng.standard_normal(n)
        if live_only:
            x[np.asarray(N.Rdead[:n]) != 0] = 0.0
        xa = np.zeros(n)
        xa[q] = x                      # (A E) x = A (E x)
        y1 = csc_matvec(m, Ap, Ai, Ax, xa)
        y2 = orc.qmult(1, S, N, orc.rmult(S, N, x))
        worst = max(worst, np.linalg.norm(y1 - y2) / (anorm * np.linalg.norm(x)))
    return worst


def numeric_from_gpu(S: Symbolic, G) -> Numeric:
    """Wrap the arrays of a package QRNumeric (GPU result) in the oracle's Numeric so the oracle's checkers
    (qmult / rmult / rsolve, rh_blocks) can read them."""
    N = Numeric.__new__(Numeric)
    N.Stack = np.ascontiguousarray(G.Stack, np.float64)
    N.Rblock_off = np.ascontiguousarray(G.Rblock_off, I64)
    N.Rdead = np.ascontiguousarray(G.Rdead, np.int8)
    N.HStair = np.ascontiguousarray(G.HStair, I64)
    N.HTau = np.ascontiguousarray(G.HTau, np.float64)
    N.Hii = np.ascontiguousarray(G.Hii, I64)
    N.HPinv = np.ascontiguousarray(G.HPinv, I64)
    N.Hm = np.ascontiguousarray(G.Hm, I64)
    N.Hr = np.ascontiguousarray(G.Hr, I64)
    N.Cm = np.zeros(max(S.nf, 1), I64)
    N.Csave = None; N.Csave_off = None
    c = OrcNumeric()
    c.Stack = _dp(N.Stack); c.Rblock_off = _ip(N.Rblock_off)
    c.Rdead = C.cast(N.Rdead.ctypes.data, C.c_char_p)
    c.HStair = _ip(N.HStair); c.HTau = _dp(N.HTau); c.Hii = _ip(N.Hii)
    c.HPinv = _ip(N.HPinv); c.Hm = _ip(N.Hm); c.Hr = _ip(N.Hr); c.Cm = _ip(N.Cm)
    c.rank = G.rank; c.rank1 = G.rank1; c.maxfrank = G.maxfrank; c.maxfm = G.maxfm; c.rh_total = G.rh_total
    N.c = c
    return N


# ---------------------------------------------------------------------------------------------------------------------
# multi-process tests: every child is ended whatever happens (a hung rank must not be left holding a GPU or a port)
def run_ranks(target, world: int, args: tuple, timeout: float):
    """Start `world` spawned processes `target(rank, world, port, *args, q)`, wait for all of them until ONE common deadline,
    and return what rank 0 put into the queue.  In a `finally` every child still alive is terminated, then killed; a rank that
    failed or did not finish is reported with its exit code (never a bare hang, never a left-over process)."""
    import socket
    import time
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    deadline = time.monotonic() + timeout
    try:
        for p in procs:
            p.start()
        result = None
        while time.monotonic() < deadline:
            if result is None:
                try:
                    result = q.get(timeout=0.2)            # drain early: a child blocks in exit while its queue item is unread
                except Exception:
                    pass
            codes = [p.exitcode for p in procs]
            if any(c not in (None, 0) for c in codes):      # one rank died: the others would wait for it for ever
                break
            if all(c == 0 for c in codes):
                break
        codes = [p.exitcode for p in procs]
        assert all(c == 0 for c in codes), f"rank exit codes {codes} (None = still running after {timeout:.0f} s or ended early)"
        if result is None:
            result = q.get(timeout=5)
        return result
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(5)
            if p.is_alive():
                p.kill()
                p.join(5)


def finish_ranks(dist, ok: bool):
    """End of a rank's work: the ranks meet once more only when THIS rank succeeded (a barrier in a `finally` turns one rank's
    exception into everybody's hang); the process group is destroyed either way."""
    try:
        if ok:
            dist.barrier()
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass
