"""GPU parity of the inner seams (SparseQR.h:145-268) against the CPU oracle on seeded dense inputs.
Each seam runs the same kernels as the full factorization (csrc/stmmqr_seams.cpp)."""
import importlib

import numpy as np
import pytest

from stmmqr_testlib import I64, front_R, rrow_signature

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
    assert p.device_count() >= 1
    return p


def make_front(m, n, seed, stair_kind="ramp"):
    rng = np.random.default_rng(seed)
    if stair_kind == "full":
        St = np.full(n, m, I64)
    elif stair_kind == "steps":     # irregular increments (1..3 rows per column), as the fronts of epb1 have
        inc = rng.integers(1, 4, n)
        St = np.minimum(m, 2 + np.cumsum(inc)).astype(I64)
    else:  # SURVEY.md 8d microbench staircase
        St = np.minimum(m, (np.arange(1, n + 1) * m) // n + 8).astype(I64)
    F = np.zeros((m, n), order="F")
    for k in range(n):
        F[:St[k], k] = rng.standard_normal(St[k])
    return F, St


FRONTS = [(6, 4, 4), (5, 8, 3), (40, 30, 12), (64, 96, 32), (130, 70, 70), (266, 422, 124), (380, 380, 380),
          (97, 33, 0), (33, 97, 97), (1, 5, 2), (700, 64, 64),
          # tall panels: the sub-panel pipeline (register-resident column groups, one launch per 8 columns)
          (1500, 96, 64), (2600, 80, 80), (900, 200, 40), (1100, 72, 72), (4200, 40, 40), (800, 1000, 900),
          # more than 4096 rows: 2-column groups (16 rows per thread), 16 groups per panel; more than 8192: one workgroup
          (6000, 48, 48), (8192, 40, 33), (8300, 36, 36),
          # rows run out in the middle of a sub-panel of a pipelined panel
          (1003, 1100, 900), (781, 900, 300), (1290, 1400, 64)]


@pytest.mark.parametrize("m,n,npiv", FRONTS)
@pytest.mark.parametrize("bigcols,tall_min", [(128, 256), (8, 256), (8, 0)])
@pytest.mark.parametrize("stair", ["ramp", "full", "steps"])
def test_qr_front(pkg, oracle, m, n, npiv, bigcols, tall_min, stair):
    """tall_min = 0: every panel of the large-front path runs as a pipeline of 8-column groups, whatever its height."""
    F0, St0 = make_front(m, n, 1234 + m + n, stair)
    Fg, Sg = F0.copy(order="F"), St0.copy()
    Fo, So = F0.copy(order="F"), St0.copy()
    pkg.set_options(big_front_cols=bigcols, tall_min_rows=tall_min)
    try:
        rg, Tg, Dg, flg = pkg.qr_front(m, n, npiv, -1.0, n, Fg, Sg)
    finally:
        pkg.set_options(big_front_cols=64, tall_min_rows=0)
    ro, To, Do, flo = oracle.front(Fo, So, npiv, -1.0, n)
    assert rg == ro
    np.testing.assert_array_equal(Sg, So)
    np.testing.assert_array_equal(Dg, Do)
    assert flg == flo
    # dense random fronts have no rounding-noise pivots: everything is uniquely determined
    scale = np.linalg.norm(Fo)
    assert np.linalg.norm(Tg - To) <= 1e-11 * max(np.linalg.norm(To), 1)
    assert np.linalg.norm(Fg - Fo) <= 1e-11 * scale


@pytest.mark.parametrize("m,n,npiv", FRONTS + [(12000, 64, 64), (20000, 40, 33)])
@pytest.mark.parametrize("stair", ["ramp", "full", "steps"])
@pytest.mark.parametrize("late", [None, 2])
def test_qr_front_gram_panel(pkg, oracle, monkeypatch, m, n, npiv, stair, late):
    """panel_algo = 2: the Gram-based panel (k_panel_ca) for every panel of the front, any height (row slabs of 480 rows, one
    workgroup each; no 8192-row limit).  late = 2: slab workgroup 2 starts ~1 ms late and therefore owns the chain."""
    F0, St0 = make_front(m, n, 1234 + m + n, stair)
    Fg, Sg = F0.copy(order="F"), St0.copy()
    Fo, So = F0.copy(order="F"), St0.copy()
    pkg.set_options(big_front_cols=8, panel_algo=2)
    if late is not None:
        monkeypatch.setenv("STMMQR_DBG", str(2048 + (late << 20)))
    try:
        rg, Tg, Dg, flg = pkg.qr_front(m, n, npiv, -1.0, n, Fg, Sg)
    finally:
        if late is not None:
            monkeypatch.delenv("STMMQR_DBG")
        pkg.set_options(big_front_cols=64, panel_algo=0)
    ro, To, Do, flo = oracle.front(Fo, So, npiv, -1.0, n)
    assert rg == ro and flg == flo
    np.testing.assert_array_equal(Sg, So)
    np.testing.assert_array_equal(Dg, Do)
    assert np.linalg.norm(Tg - To) <= 1e-11 * max(np.linalg.norm(To), 1)
    assert np.linalg.norm(Fg - Fo) <= 1e-11 * np.linalg.norm(Fo)


@pytest.mark.parametrize("late", [0, 1, 2, 3])
@pytest.mark.parametrize("m,n,npiv", [(266, 422, 124), (781, 900, 300), (1290, 1400, 64), (1500, 96, 64), (33, 97, 97),
                                     (5000, 70, 64)])
def test_pipeline_with_a_late_column_group(pkg, oracle, monkeypatch, late, m, n, npiv):
    """The column groups of a pipelined panel may start in any order and arbitrarily late (a launch with more
    workgroups than the GPU holds at once): STMMQR_DBG bit 11 delays group `late` by ~1 ms.  Fronts whose rows run out
    in the middle of a panel are the interesting ones (the finalising group is then not the last one)."""
    F0, St0 = make_front(m, n, 4321 + m + n, "steps")
    Fg, Sg = F0.copy(order="F"), St0.copy()
    Fo, So = F0.copy(order="F"), St0.copy()
    pkg.set_options(big_front_cols=8, tall_min_rows=0)
    monkeypatch.setenv("STMMQR_DBG", str(2048 + (late << 20)))
    try:
        rg, Tg, Dg, flg = pkg.qr_front(m, n, npiv, -1.0, n, Fg, Sg)
    finally:
        monkeypatch.delenv("STMMQR_DBG")
        pkg.set_options(big_front_cols=64, tall_min_rows=0)
    ro, To, Do, flo = oracle.front(Fo, So, npiv, -1.0, n)
    assert rg == ro and flg == flo
    np.testing.assert_array_equal(Sg, So)
    assert np.linalg.norm(Tg - To) <= 1e-11 * max(np.linalg.norm(To), 1)
    assert np.linalg.norm(Fg - Fo) <= 1e-11 * np.linalg.norm(Fo)


@pytest.mark.parametrize("m", [90, 1300])
def test_qr_front_dead_columns(pkg, oracle, m):
    n, npiv = 60, 40
    F0, St0 = make_front(m, n, 77, "full")
    F0[:, 7] = F0[:, 3]; F0[:, 21] = 2 * F0[:, 20] - F0[:, 19]; F0[:, 33] = 0
    tol = 1e-9
    for bigcols, algo in ((128, 0), (8, 1), (8, 2)):
        Fg, Sg = F0.copy(order="F"), St0.copy()
        Fo, So = F0.copy(order="F"), St0.copy()
        pkg.set_options(big_front_cols=bigcols, panel_algo=algo)
        try:
            rg, Tg, Dg, _ = pkg.qr_front(m, n, npiv, tol, npiv, Fg, Sg)
        finally:
            pkg.set_options(big_front_cols=64, panel_algo=0)
        ro, To, Do, _ = oracle.front(Fo, So, npiv, tol, npiv)
        assert rg == ro == npiv - 3
        np.testing.assert_array_equal(Dg, Do)
        np.testing.assert_array_equal(Sg, So)
        assert Dg[7] == 1 and Dg[21] == 1 and Dg[33] == 1
        assert np.linalg.norm(Tg - To) <= 1e-10 * np.linalg.norm(To)
        assert np.linalg.norm(Fg - Fo) <= 1e-10 * np.linalg.norm(Fo)


@pytest.mark.parametrize("m,n,k", [(50, 20, 7), (200, 70, 32), (300, 33, 80), (64, 64, 64), (10, 3, 10)])
def test_qr_larftb(pkg, oracle, m, n, k):
    import ctypes as C
    from stmmqr_testlib import _dp
    rng = np.random.default_rng(m * 7 + n)
    kk = min(k, m)
    V = np.asfortranarray(rng.standard_normal((m, k)))
    Tau = rng.uniform(1.0, 2.0, k)
    if k > 3:
        Tau[2] = 0.0
    Cm = np.asfortranarray(rng.standard_normal((m, n)))
    Cg, Co = Cm.copy(order="F"), Cm.copy(order="F")
    pkg.qr_larftb(0, m, n, kk, m, m, V, Tau, Cg)
    # oracle: apply in chunks of <= 32 exactly like qr_front / qr_panel do
    W = np.zeros(32 * 32 + 32 * n + 64)
    for k1 in range(0, kk, 32):
        nb = min(32, kk - k1)
        dptr = lambda a, off: C.cast(a.ctypes.data + 8 * off, C.POINTER(C.c_double))
        oracle.lib.orc_larftb(0, m - k1, n, nb, m, m, dptr(V, k1 + k1 * m), dptr(Tau, k1), dptr(Co, k1), _dp(W))
    assert np.linalg.norm(Cg - Co) <= 1e-12 * np.linalg.norm(Co)


def _dense_q(V, Tau):
    """Q = H_1 ... H_k with H_j = I - tau_j v_j v_j', v_j = column j of V with a unit diagonal and zeros above it."""
    m, k = V.shape
    Q = np.eye(m)
    for j in range(min(k, m)):
        v = V[:, j].copy()
        v[:j] = 0.0
        v[j] = 1.0
        Q = Q @ (np.eye(m) - Tau[j] * np.outer(v, v))
    return Q


@pytest.mark.parametrize("method", [0, 1, 2, 3])
@pytest.mark.parametrize("m,n,k", [(50, 20, 7), (200, 70, 32), (130, 33, 80), (64, 64, 64), (10, 3, 10), (33, 1, 5)])
def test_qr_larftb_all_methods(pkg, oracle, method, m, n, k):
    """All four methods of the exported seam (the reference's qr_panel calls it with any of them, SparseQR.c:1659,1663)
    against the dense Q built reflector by reflector; the left-side ones also against the oracle's dlarfb restatement.
    QR_QTX/QR_QX: C is m x n, V m x k.  QR_XQT/QR_XQ: C is n x m here (so that V stays m x k)."""
    import ctypes as C
    from stmmqr_testlib import _dp
    rng = np.random.default_rng(1000 * method + m * 7 + n)
    kk = min(k, m)
    V = np.asfortranarray(rng.standard_normal((m, k)))
    Tau = rng.uniform(1.0, 2.0, k)
    if k > 3:
        Tau[2] = 0.0
    Q = _dense_q(V[:, :kk], Tau[:kk])
    if method < 2:
        C0 = np.asfortranarray(rng.standard_normal((m, n)))
        want = (Q.T @ C0) if method == 0 else (Q @ C0)
        Cg = C0.copy(order="F")
        pkg.qr_larftb(method, m, n, kk, m, m, V, Tau, Cg)
        if kk <= 32:
            Co = C0.copy(order="F")
            W = np.zeros(32 * 32 + 32 * n + 64)
            oracle.lib.orc_larftb(method, m, n, kk, m, m, _dp(V), _dp(Tau), _dp(Co), _dp(W))
            assert np.linalg.norm(Cg - Co) <= 1e-12 * np.linalg.norm(Co)
    else:
        C0 = np.asfortranarray(rng.standard_normal((n, m)))
        want = (C0 @ Q.T) if method == 2 else (C0 @ Q)
        Cg = C0.copy(order="F")
        pkg.qr_larftb(method, n, m, kk, n, m, V, Tau, Cg)
    assert np.linalg.norm(Cg - want) <= 1e-12 * np.linalg.norm(want)


def test_qr_larftb_export_reports_failures(pkg):
    """The reference-named export never returns silently: an unknown method leaves cc->status < 0 and a message."""
    import ctypes as C
    from stmmqr_testlib import _dp
    lay = pkg.get_common_layout() if hasattr(pkg, "get_common_layout") else None
    cc = (C.c_char * 2048)()
    V = np.asfortranarray(np.ones((4, 2)))
    Tau = np.ones(2)
    Cm = np.asfortranarray(np.ones((4, 3)))
    W = np.zeros(64)
    pkg.lib.qr_larftb(7, 4, 3, 2, 4, 4, _dp(V), _dp(Tau), _dp(Cm), _dp(W), C.cast(cc, C.c_void_p))
    off = 1004 if lay is None else lay["status"]
    status = C.c_int.from_buffer(cc, off).value
    assert status == -4
    assert b"method" in pkg.lib.stmmqr_last_error()
    np.testing.assert_array_equal(Cm, 1.0)
    # and a valid call through the same export works and leaves status alone
    cc2 = (C.c_char * 2048)()
    pkg.lib.qr_larftb(1, 4, 3, 2, 4, 4, _dp(V), _dp(Tau), _dp(Cm), _dp(W), C.cast(cc2, C.c_void_p))
    assert C.c_int.from_buffer(cc2, off).value == 0
    assert not np.allclose(Cm, 1.0)


@pytest.mark.parametrize("m,n,npiv,g", [(10, 8, 3, 3), (40, 50, 20, 17), (300, 200, 64, 64), (5, 9, 2, 2), (6, 6, 6, 6)])
def test_qr_cpack(pkg, oracle, m, n, npiv, g):
    from stmmqr_testlib import _dp
    rng = np.random.default_rng(9)
    F = np.asfortranarray(rng.standard_normal((m, n)))
    cm, Cg = pkg.qr_cpack(m, n, npiv, g, F)
    Co = np.zeros(max(pkg.qr_fcsize(m, n, npiv, g), 1))
    cmo = oracle.lib.orc_cpack(m, n, npiv, g, _dp(F), _dp(Co))
    assert cm == cmo
    np.testing.assert_array_equal(Cg, Co[:Cg.size])


@pytest.mark.parametrize("m,n,npiv", [(12, 9, 4), (64, 96, 32), (266, 422, 124), (50, 20, 20), (7, 30, 10)])
def test_qr_rhpack(pkg, oracle, m, n, npiv):
    import ctypes as C
    from stmmqr_testlib import _dp, _ip
    F0, St0 = make_front(m, n, 5)
    Fo, So = F0.copy(order="F"), St0.copy()
    oracle.front(Fo, So, npiv, -1.0, n)            # a realistic post-factorization staircase
    So[1] = 0 if npiv > 1 else So[1]                # plus one dead pivot
    rs, rm, Rg = pkg.qr_rhpack(m, n, npiv, So, Fo)
    Ro = np.zeros(m * n); rmo = C.c_long(0)
    rso = oracle.lib.orc_rhpack(m, n, npiv, _ip(So), _dp(Fo), _dp(Ro), C.byref(rmo))
    assert (rs, rm) == (rso, rmo.value)
    np.testing.assert_array_equal(Rg, Ro[:rs])


def test_micro_assembly_column_sums(pkg):
    """The synthetic assembly of bench.py --workload micro at a small size: qr_assemble is an extend-COPY,
    so every column of F must sum to the S entries plus the children's packed C entries that map to it; the seam reports
    the device time of its kernels."""
    import bench
    P, CN = 4, 40
    a, nbytes, (fm, fn) = bench.micro_assembly_input(seed=5, P=P, FN=96, FP=32, CN=CN, NS=12)
    F, _ = pkg.qr_assemble(**a)
    assert F.shape == (fm, fn)
    expect = np.zeros(fn)
    Sp, Sj, Sx, Fmap, Rp, Rj = a["Sp"], a["Sj"], a["Sx"], a["Fmap"], a["Rp"], a["Rj"]
    for r in range(len(Sp) - 1):
        for q in range(Sp[r], Sp[r + 1]):
            expect[Fmap[Sj[q]]] += Sx[q]
    for c in (0, 1):
        cols = Fmap[Rj[Rp[c] + P:Rp[c + 1]]]
        pos = 0
        for k in range(CN):                                   # packed upper triangle: column k holds rows 0..k
            expect[cols[k]] += a["Cblocks"][c][pos:pos + k + 1].sum()
            pos += k + 1
    assert np.allclose(F.sum(axis=0), expect, rtol=0, atol=1e-12 * max(1.0, np.abs(expect).max()))
    assert np.count_nonzero(F) == len(Sx) + 2 * (CN * (CN + 1) // 2)
    st = a["Stair"]
    assert np.all(np.diff(st) >= 0) and st[-1] == fm
    assert pkg.last_seam_ms() > 0.0


def test_bench_micro_contract(pkg):
    """bench.py --workload micro: one JSON line, every front factorized with the rank it must have, device
    times reported by the seams, the synthetic assembly conserves its entries."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--workload", "micro", "--no-cpu", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "dtype", "config"):
        assert k in d
    assert len(d["fronts"]) == 10
    for f in d["fronts"]:
        assert f["rank"] == f["fp"] and f["device_ms"] > 0 and f["gflops"] > 0
    assert d["assembly"]["entries_conserved"] and d["assembly"]["device_ms"] > 0
    dense8k = [f for f in d["fronts"] if f["fm"] == 8192 and f["staircase"] == "dense"][0]
    assert dense8k["gflops"] > 3000.0          # (the 2-column pipeline of the 4096..8192-row panels, not the one-workgroup fallback)


# ---------------------------------------------------------------------------------------------------------------------
# qr_stranspose2 / qr_fsize / qr_assemble / qr_hpinv, seam by seam against the oracle's functions on the fronts of real
# fixtures: the oracle factorization is walked front by front (tests/oracle_plan.py) and every front's inputs -- the
# children's packed C blocks, Cm, Hr, Hii as the ORACLE left them -- go through the HIP seam: everything must be identical
# (integers, and F entry by entry: the assembly is an extend-COPY).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["syn_grid3d", "syn_rankdef_grid", "syn_star", "bcsstk14", "epb1", "lns_3937"])
def test_front_seams_on_real_fronts(pkg, oracle, name):
    import ctypes as C
    from oracle_plan import OraclePlan
    from stmmqr_testlib import Symbolic, _dp, _ip, c_double_p, golden_names, load_golden, scalar
    if name not in golden_names():
        pytest.skip("fixture not present")
    g = load_golden(name)
    S = Symbolic(g)
    # qr_stranspose2: exact gather
    Sx_g = pkg.qr_stranspose2(S.m, S.n, g["in_Ap"], g["in_Ai"], g["in_Ax"], S.arr.get("Qfill"), S.Sp, S.PLinv)
    P = OraclePlan(S, oracle)
    P.begin(g["in_Ax"], scalar(g, "in_tol"), int(scalar(g, "in_ntol")), g["in_Ap"], g["in_Ai"])
    np.testing.assert_array_equal(Sx_g, P.Sx[:S.anz])
    L = oracle.lib
    checked = 0
    orig = L.orc_assemble

    for f in S.Post[:S.nf]:
        f = int(f)
        fn = int(S.Rp[f + 1] - S.Rp[f])
        # ---- qr_fsize on the oracle's state before this front ----
        Fmap_o, Fmap_g = P.Fmap.copy(), P.Fmap.copy()
        St_o = np.zeros(max(fn, 1), I64); St_g = np.zeros(max(fn, 1), I64)
        fm_o = L.orc_fsize(f, _ip(S.Super), _ip(S.Rp), _ip(S.Rj), _ip(S.Sleft), _ip(S.Child), _ip(S.Childp), _ip(P.Cm), _ip(Fmap_o), _ip(St_o))
        fm_g = pkg.qr_fsize(f, S.Super, S.Rp, S.Rj, S.Sleft, S.Child, S.Childp, P.Cm, Fmap_g, St_g)
        assert fm_g == fm_o
        np.testing.assert_array_equal(St_g[:fn], St_o[:fn])
        np.testing.assert_array_equal(Fmap_g[S.Rj[S.Rp[f]:S.Rp[f + 1]]], Fmap_o[S.Rj[S.Rp[f]:S.Rp[f + 1]]])
        # ---- qr_assemble: same inputs to both ----
        kids = [int(S.Child[q]) for q in range(S.Childp[f], S.Childp[f + 1])]
        if fm_o * fn > 0 and (checked < 40 or fn >= 64):
            Fo = np.zeros(max(fm_o * fn, 1)); Cmap_o = np.zeros(max(S.maxfn, 1), I64)
            Hii_o, Hii_g = P.Hii.copy(), P.Hii.copy()
            So, Sg = St_o.copy(), St_o.copy()
            ptrs = (c_double_p * (S.nf + 1))()
            for c in kids:
                ptrs[c] = _dp(P.Cblk[c])
            orig(f, fm_o, _ip(S.Super), _ip(S.Rp), _ip(S.Rj), _ip(S.Sp), _ip(S.Sj), _ip(S.Sleft), _ip(S.Child), _ip(S.Childp),
                 _dp(P.Sx), _ip(Fmap_o), _ip(P.Cm), ptrs, _ip(P.Hr), _ip(So), _ip(Hii_o), _ip(S.Hip), _dp(Fo), _ip(Cmap_o))
            Fg, Cmap_g = pkg.qr_assemble(f, fm_o, S.Super, S.Rp, S.Rj, S.Sp, S.Sj, S.Sleft, S.Child, S.Childp, P.Sx, Fmap_o,
                                         P.Cm, {c: P.Cblk[c] for c in kids}, P.Hr, Sg, Hii_g, S.Hip)
            np.testing.assert_array_equal(Fg, Fo[:fm_o * fn].reshape((fn, fm_o)).T)
            np.testing.assert_array_equal(Sg[:fn], So[:fn])
            a = int(S.Hip[f])
            np.testing.assert_array_equal(Hii_g[a:a + fm_o], Hii_o[a:a + fm_o])
            # Cmap is scratch that each child overwrites: what is left is the LAST child's row map
            if kids:
                cm = int(P.Cm[kids[-1]])
                np.testing.assert_array_equal(Cmap_g[:cm], Cmap_o[:cm])
            checked += 1
        P.group[:] = -1
        P.group[f] = 0
        P.run_group(0)                          # the oracle factorizes the front: state for the parents
    assert checked >= 1
    # ---- qr_hpinv on the oracle's final Hii / Hm / Hr ----
    sym = {**S.sc, **{k: v for k, v in S.arr.items() if v is not None}}
    No = oracle.factorize(S, g["in_Ap"], g["in_Ai"], g["in_Ax"], scalar(g, "in_tol"), int(scalar(g, "in_ntol")))
    Hii_g = P.Hii.copy()
    HPinv_g, maxfm_g = pkg.qr_hpinv(sym, P.Hm, P.Hr, Hii_g)
    np.testing.assert_array_equal(HPinv_g, No.HPinv[:S.m])
    assert maxfm_g == No.c.maxfm
    for f in range(S.nf):
        a = int(S.Hip[f])
        np.testing.assert_array_equal(Hii_g[a:a + No.Hm[f]], No.Hii[a:a + No.Hm[f]])


@pytest.mark.parametrize("scale", [1e160, 1e-160, 3e-200])
@pytest.mark.parametrize("m,n,npiv,bigcols,algo", [(40, 30, 12, 128, 0), (380, 380, 380, 128, 0), (700, 64, 64, 8, 1),
                                                  (1500, 96, 64, 8, 1), (1500, 96, 64, 8, 2), (900, 200, 40, 8, 2)])
def test_qr_front_extreme_magnitudes(pkg, oracle, m, n, npiv, bigcols, algo, scale):
    """Entries around 1e+-160: the sums of squares of an unguarded dlarfg overflow / underflow there (LAPACK's dlarfg
    rescales, the oracle restates that: orc_larfg, SparseQR_factorize.c:1309-1326).  The panel kernels carry ONE power-of-two
    factor for the whole factorization in their sums (stm_larfg_guarded): V and Tau must match the oracle as at scale 1,
    R up to the scale.  All panel paths: one-workgroup LDS panel, column pipeline, Gram-based."""
    F0, St0 = make_front(m, n, 991 + m + n, "steps")
    F0 *= scale
    Fg, Sg = F0.copy(order="F"), St0.copy()
    Fo, So = F0.copy(order="F"), St0.copy()
    pkg.set_options(big_front_cols=bigcols, panel_algo=algo)
    try:
        rg, Tg, Dg, flg = pkg.qr_front(m, n, npiv, -1.0, n, Fg, Sg)
    finally:
        pkg.set_options(big_front_cols=64, panel_algo=0)
    ro, To, Do, flo = oracle.front(Fo, So, npiv, -1.0, n)
    assert rg == ro and flg == flo
    np.testing.assert_array_equal(Sg, So)
    assert np.all(np.isfinite(Fg)) and np.all(np.isfinite(Tg))
    assert np.linalg.norm(Tg - To) <= 1e-11 * max(np.linalg.norm(To), 1)
    # V is scale free, R carries the scale: compare entry by entry relative to the column's largest entry
    colmax = np.maximum(np.abs(Fo).max(axis=0), np.finfo(float).tiny)
    assert np.max(np.abs(Fg / colmax - Fo / colmax)) <= 1e-10
