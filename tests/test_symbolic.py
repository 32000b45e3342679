"""Own symbolic phase (stmmqr_analyze, csrc/stmmqr_symbolic.cpp) against the reference's qr_analyze: given
the matrix handed to qr_factorize and the column permutation (Qfill) of a golden fixture, EVERY array and scalar of the
reference's qr_symbolic (sym_* of the fixture, dumped from the compiled reference by oracle/refdump.c) must come out bit
for bit -- supernodes, row structures, frontal tree, weighted post-order, S in row form, front sizes, bounds.
Integer work: exact equality.  Host-only, no GPU."""
import importlib

import numpy as np
import pytest

from stmmqr_testlib import golden_names, load_golden, scalar

PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"
NAMES = golden_names(True)
ARRAYS = ["Sp", "Sj", "PLinv", "Sleft", "Parent", "Child", "Childp", "Super", "Rp", "Rj", "Post", "Hip", "Fm", "Cm"]


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG)


def original_size(g):
    """(n, nnz) of the matrix the DRIVER read (Relaxfactor_setting is called with those, qrtest.c:153), which differs
    from the matrix handed to qr_factorize when singletons were removed"""
    n = int(scalar(g, "A_n")) if "A_n" in g else int(scalar(g, "in_n"))
    if "A_p" in g:
        return n, int(g["A_p"][-1])
    if "A_x" in g:
        return n, len(g["A_x"])
    return n, len(g["in_Ax"]) if "in_Ax" in g else int(g["in_Ap"][-1])


@pytest.mark.parametrize("name", NAMES)
def test_analysis_matches_reference(pkg, name):
    g = load_golden(name)
    m, n = int(scalar(g, "in_m")), int(scalar(g, "in_n"))
    Q = g["sym_Qfill"] if "sym_Qfill" in g and len(g["sym_Qfill"]) else None
    n0, nnz0 = original_size(g)
    relax = pkg.relax_for_qr(n0, nnz0)
    A = pkg.analyze(m, n, g["in_Ap"], g["in_Ai"], Q, bool(scalar(g, "sym_do_rank_detection")), relax)
    for k in ("m", "n", "anz", "nf", "maxfn", "rjsize", "hisize", "maxstack", "do_rank_detection", "keepH", "ntasks", "ns"):
        assert A[k] == int(scalar(g, "sym_" + k)), k
    for k in ARRAYS:
        want = np.asarray(g["sym_" + k], np.int64)
        got = A[k]
        if k == "Rj":
            want = want[:len(got)]
        if k in ("Fm", "Cm"):
            got, want = got[:A["nf"]], want[:A["nf"]]       # (entry nf is allocated but never written by the reference)
        np.testing.assert_array_equal(got, want, err_msg=k)
    if Q is not None:
        np.testing.assert_array_equal(A["Qfill"], np.asarray(Q, np.int64))
    assert A["info"][0] == scalar(g, "flopcount_bound")
    # FCHUNK as qr_analyze left it (80 when the Cholesky analysis says fl / lnz >= 1000, SparseQR_analyze.c:666-670)
    assert (80 if A["info"][3] else 32) == int(scalar(g, "FCHUNK"))


def test_analysis_rejects_bad_input(pkg):
    Ap = np.array([0, 1, 2], np.int64)
    Ai = np.array([0, 5], np.int64)
    with pytest.raises(pkg.StmmqrError):
        pkg.analyze(2, 2, Ap, Ai)                       # row index out of range
    with pytest.raises(pkg.StmmqrError):
        pkg.analyze(2, 2, Ap, np.array([0, 1], np.int64), np.array([0, 0], np.int64))   # not a permutation
    with pytest.raises(pkg.StmmqrError):
        pkg.analyze(2, 2, np.array([0, 2, 1], np.int64), np.array([0, 1], np.int64))    # decreasing pointers


def test_analysis_of_empty_and_trivial_matrices(pkg):
    A = pkg.analyze(3, 0, np.zeros(1, np.int64), np.zeros(0, np.int64))
    assert A["nf"] == 0 and A["hisize"] == 0 and A["maxfn"] == 0
    np.testing.assert_array_equal(A["PLinv"], [0, 1, 2])
    A = pkg.analyze(2, 2, np.array([0, 0, 0], np.int64), np.zeros(0, np.int64))      # two empty columns
    assert A["nf"] >= 1 and A["Fm"][:A["nf"]].sum() == 0
