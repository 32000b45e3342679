"""R / H export (qr_rcount, qr_rconvert, qr_trapezoidal of csrc/stmmqr_export.cpp under the reference's names,
STMMQR/src/qr/SparseLQ.c:102-689) against the outputs of the REFERENCE's functions of the same names.
The input is the reference's own numeric factorization (the packed R+H stack, HStair, HTau, Hii, Hm of the golden fixture:
oracle/refdump.c), the expected output what the reference's qr_rcount / qr_rconvert / qr_trapezoidal made of that same
factorization (tests/golden/api/api_reference.npz: oracle/refapi.c).  The functions move and count data: bit-exact.
Host-only, no GPU."""
import ctypes as C
import importlib
from pathlib import Path

import numpy as np
import pytest

from stmmqr_testlib import load_golden, scalar

ROOT = Path(__file__).resolve().parent.parent
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"
GOLD = ROOT / "tests" / "golden" / "api" / "api_reference.npz"
CASES = ["syn_dupcol", "syn_rankdef_grid", "syn_wide5x8", "syn_star", "syn_chain", "syn_rand60x40"]
I64 = np.int64


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG)


def reference_objects(pkg, g):
    """qr_symbolic / qr_numeric of the fixture as the C structs (keeps the arrays alive in the returned dict)"""
    capi = importlib.import_module(PKG + ".capi")
    keep = {}
    S = capi.QrSymbolicC()
    for k in ("m", "n", "anz", "nf", "maxfn", "rjsize", "do_rank_detection", "maxstack", "hisize", "keepH"):
        setattr(S, k, int(scalar(g, "sym_" + k)))
    for k in ("Sp", "Sj", "PLinv", "Sleft", "Parent", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip", "Fm", "Cm"):
        keep["s" + k] = np.ascontiguousarray(g["sym_" + k], I64)
        setattr(S, k, ip(keep["s" + k]))
    nf = S.nf
    stack = np.ascontiguousarray(g["num_Stack"], np.float64)
    off = np.asarray(g["num_Rblock_off"], I64)
    keep["stack"] = stack
    rblock = (C.c_void_p * max(nf, 1))(*[stack.ctypes.data + 8 * int(o) for o in off[:nf]])
    keep["rblock"] = rblock
    N = capi.QrNumericC()
    N.Rblock = C.cast(rblock, C.c_void_p)
    N.keepH, N.nf, N.n, N.m, N.rjsize, N.hisize = 1, nf, S.n, S.m, S.rjsize, S.hisize
    for k, key in (("HStair", "num_HStair"), ("Hii", "num_Hii"), ("Hm", "num_Hm"), ("Hr", "num_Hr"), ("HPinv", "num_HPinv")):
        keep[k] = np.ascontiguousarray(g[key], I64)
        setattr(N, k, ip(keep[k]))
    keep["HTau"] = np.ascontiguousarray(g["num_HTau"], np.float64)
    N.HTau = dp(keep["HTau"])
    keep["Rdead"] = np.ascontiguousarray(g["num_Rdead"], np.int8)
    N.Rdead = C.cast(keep["Rdead"].ctypes.data, C.c_char_p)
    keep["S"], keep["N"] = S, N
    return keep


@pytest.mark.parametrize("name", CASES)
def test_rcount_rconvert_trapezoidal_match_reference(pkg, name):
    gold = np.load(GOLD)
    key = f"{name}@-1"
    ref = {k.split(":", 1)[1]: gold[k] for k in gold.files if k.startswith(key + ":")}
    assert "rc_Rp" in ref
    g = load_golden(name)
    assert "num_Stack" in g and len(g["num_Stack"]) >= int(np.sum(g["num_rh_size"]))       # (fixtures that store the full stack)
    K = reference_objects(pkg, g)
    S, N = K["S"], K["N"]
    lib = pkg.lib
    lib.qr_rcount.restype = None
    lib.qr_rconvert.restype = None
    lib.qr_trapezoidal.restype = C.c_long
    n, econ = S.n, S.m
    Ra = np.zeros(n + 1, I64)
    H2p = np.zeros(S.rjsize + 2, I64)
    nh = C.c_long(0)
    lib.qr_rcount(C.byref(S), C.byref(N), C.c_long(0), C.c_long(econ), C.c_long(n), 0, ip(Ra), None, ip(H2p), C.byref(nh))
    Rp = np.concatenate([[0], np.cumsum(Ra[:n])]).astype(I64)
    np.testing.assert_array_equal(Rp, ref["rc_Rp"])
    assert nh.value == int(ref["rc_nh"][0])
    np.testing.assert_array_equal(H2p[:nh.value + 1], ref["rc_Hp"])
    tot, hnz = int(Rp[-1]), int(H2p[nh.value])
    fill = Rp.copy()
    Ri, Rx = np.zeros(max(tot, 1), I64), np.zeros(max(tot, 1))
    Hi, Hx, Ht = np.zeros(max(hnz, 1), I64), np.zeros(max(hnz, 1)), np.zeros(max(nh.value, 1))
    lib.qr_rconvert(C.byref(S), C.byref(N), C.c_long(0), C.c_long(econ), C.c_long(n), 0, ip(fill), ip(Ri), dp(Rx), None, None, None,
                    ip(H2p), ip(Hi), dp(Hx), dp(Ht))
    np.testing.assert_array_equal(Ri[:tot], ref["rc_Ri"])
    np.testing.assert_array_equal(Rx[:tot], ref["rc_Rx"])
    np.testing.assert_array_equal(Hi[:hnz], ref["rc_Hi"])
    np.testing.assert_array_equal(Hx[:hnz], ref["rc_Hx"])
    np.testing.assert_array_equal(Ht[:nh.value], ref["rc_HTau"])
    # qr_trapezoidal on that R (allocations through the cc accounting: a zeroed sparse_common stands in)
    cc = (C.c_char * 2048)()
    Tp, Ti, Qt = C.POINTER(C.c_long)(), C.POINTER(C.c_long)(), C.POINTER(C.c_long)()
    Tx = C.POINTER(C.c_double)()
    Qfill = np.ascontiguousarray(g["sym_Qfill"], I64) if len(g["sym_Qfill"]) else None
    rank = lib.qr_trapezoidal(C.c_long(n), ip(Rp), ip(Ri), dp(Rx), C.c_long(0), None if Qfill is None else ip(Qfill), 0, C.byref(Tp),
                              C.byref(Ti), C.byref(Tx), C.byref(Qt), C.cast(cc, C.c_void_p))
    assert rank == int(ref["rc_trap_rank"][0])
    if "rc_Tp" in ref:
        np.testing.assert_array_equal(np.ctypeslib.as_array(Tp, shape=(n + 1,)), ref["rc_Tp"])
        np.testing.assert_array_equal(np.ctypeslib.as_array(Ti, shape=(max(tot, 1),))[:tot], ref["rc_Ti"])
        np.testing.assert_array_equal(np.ctypeslib.as_array(Tx, shape=(max(tot, 1),))[:tot], ref["rc_Tx"])
        np.testing.assert_array_equal(np.ctypeslib.as_array(Qt, shape=(n,)), ref["rc_Qtrap"])
        assert C.c_size_t.from_buffer(cc, 1032).value == 4              # four counted allocations (cc->malloc_count)
