"""The host half of stmmqr_sparseqr (csrc/stmmqr_sparseqr.cpp + stmmqr_colamd.cpp + stmmqr_symbolic.cpp: column singletons,
COLAMD + column-etree post-order, R1 / Y split, symbolic analysis) against what the REFERENCE's SparseQR() did with the same
matrix (golden fixtures dumped by oracle/refdump.c from the compiled reference, driver defaults = COLAMD):
  * n1rows / n1cols,
  * the matrix handed to qr_factorize -- A itself with the permutation Qfill, or Y after singleton removal -- bit for bit
    (this pins Q1fill, P1inv and the R1 / Y split: Y's columns are A's columns in Q1fill order with rows renumbered by P1inv),
  * every array of the qr_symbolic.
Integer work (values are copied, not computed): exact equality.  Host-only, no GPU."""
import importlib

import numpy as np
import pytest

from stmmqr_testlib import golden_names, load_golden, scalar

PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"
ARRAYS = ["Sp", "Sj", "PLinv", "Sleft", "Parent", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip", "Fm", "Cm"]
QR_ORDERING_DEFAULT, QR_ORDERING_COLAMD = 7, 2


def colamd_fixtures():
    out = []
    for name in golden_names(True):
        g = load_golden(name)
        # (the large stand-ins store no copy of A: without singletons the matrix handed to qr_factorize IS A)
        if ("A_p" in g or int(scalar(g, "n1cols")) == 0) and int(scalar(g, "ordering")) in (QR_ORDERING_DEFAULT, QR_ORDERING_COLAMD):
            out.append(name)
    return out


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG)


@pytest.mark.parametrize("name", colamd_fixtures())
def test_sparseqr_symbolic_matches_reference(pkg, name):
    g = load_golden(name)
    if "A_p" in g:
        m, n = int(scalar(g, "A_m")), int(scalar(g, "A_n"))
        Ap, Ai, Ax = g["A_p"], g["A_i"], g["A_x"]
    else:
        m, n = int(scalar(g, "in_m")), int(scalar(g, "in_n"))
        Ap, Ai, Ax = g["in_Ap"], g["in_Ai"], g["in_Ax"]
    relax = pkg.relax_for_qr(n, int(Ap[-1]))
    tol = scalar(g, "QR_tol") if "QR_tol" in g else scalar(g, "in_tol")
    Q = pkg.SparseQR(m, n, Ap, Ai, Ax, ordering=QR_ORDERING_DEFAULT, tol=tol, relax=relax, symbolic_only=True)
    info = Q.info
    assert int(info["n1rows"]) == int(scalar(g, "n1rows"))
    assert int(info["n1cols"]) == int(scalar(g, "n1cols"))
    Y = Q.Y()
    if int(info["n1cols"]) > 0:
        assert Y is not None
        np.testing.assert_array_equal(Y[0], g["in_Ap"])
        np.testing.assert_array_equal(Y[1], g["in_Ai"])
        np.testing.assert_array_equal(Y[2], g["in_Ax"])
        # the singleton columns come first in Q1fill, the rest follow in Y's column order
        q1 = Q.Q1fill
        assert sorted(q1.tolist()) == list(range(n))
    else:
        assert Y is None
        np.testing.assert_array_equal(Q.Q1fill, np.asarray(g["sym_Qfill"], np.int64))
    S = Q.symbolic()
    for k in ("m", "n", "anz", "nf", "maxfn", "rjsize", "hisize", "maxstack", "do_rank_detection"):
        assert S[k] == int(scalar(g, "sym_" + k)), k
    for k in ARRAYS:
        want, got = np.asarray(g["sym_" + k], np.int64), S[k]
        if k == "Rj":
            want = want[:len(got)]
        if k in ("Fm", "Cm"):
            got, want = got[:S["nf"]], want[:S["nf"]]
        np.testing.assert_array_equal(got, want, err_msg=k)
    assert info["flop_bound"] == scalar(g, "flopcount_bound")
    Q.close()


def test_orderings_not_built_are_refused_loudly(pkg):
    Ap = np.array([0, 1, 2], np.int64); Ai = np.array([0, 1], np.int64); Ax = np.ones(2)
    for ordering in (5, 6, 10, 11, 4, 8):
        with pytest.raises(pkg.StmmqrError, match="not built"):
            pkg.SparseQR(2, 2, Ap, Ai, Ax, ordering=ordering, tol=0.0, symbolic_only=True)
