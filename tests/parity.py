"""Parity rules shared by the GPU tests and __graft_entry__.smoke() (TEST INFRASTRUCTURE).

What "identical to the reference" means for this path (see tests/test_oracle_golden.py for the evidence):
  * integer outputs -- Hm, Hr, HStair, Hii, HPinv, Rdead, rank, rank1, maxfrank, maxfm, block offsets: bit exact;
  * R: unique up to the sign of each row -> per-row (|diag|, norm, |<row,w>|) signatures to `ftol`;
  * Tau / Householder vectors: element-wise only on inputs without rounding-noise pivots (ELEMENTWISE set);
    everywhere: backward error ||A E - Q R|| / ||A|| <= 1e-13 through the packed factors.
"""
import numpy as np

from stmmqr_testlib import (aqr_probe_error, cond_probe, front_R, numeric_from_gpu, rrow_excess, rrow_signature, rrow_signature_of_block, scalar)

ELEMENTWISE = ("syn_dense6x4", "syn_wide5x8", "syn_dupcol", "syn_emptycol", "syn_chain", "syn_star",
               "syn_rand60x40")


def rrow_sig_all(S, N):
    blocks = N.rh_blocks(S)
    out = []
    for f in range(S.nf):
        fn, fp = S.Rp[f + 1] - S.Rp[f], S.Super[f + 1] - S.Super[f]
        out.append(rrow_signature_of_block(blocks[f], N.HStair[S.Rp[f]:S.Rp[f + 1]], fp, fn, N.Hm[f]))
    return np.concatenate(out) if out else np.zeros((0, 3))


def compare_integers(S, N, ref: dict):
    """N: oracle-style Numeric of the implementation under test; ref: dict with num_* arrays."""
    nf, n, m = S.nf, S.n, S.m
    assert N.c.rank == scalar(ref, "num_rank")
    assert N.c.rank1 == scalar(ref, "num_rank1")
    assert N.c.maxfrank == scalar(ref, "num_maxfrank")
    assert N.c.maxfm == scalar(ref, "num_maxfm")
    np.testing.assert_array_equal(N.Rdead[:n], ref["num_Rdead"][:n])
    np.testing.assert_array_equal(N.Hm[:nf], ref["num_Hm"][:nf])
    np.testing.assert_array_equal(N.Hr[:nf], ref["num_Hr"][:nf])
    np.testing.assert_array_equal(N.HStair[:S.rjsize], ref["num_HStair"][:S.rjsize])
    np.testing.assert_array_equal(N.HPinv[:m], ref["num_HPinv"][:m])
    for f in range(nf):
        a = S.Hip[f]
        np.testing.assert_array_equal(N.Hii[a:a + N.Hm[f]], ref["num_Hii"][a:a + N.Hm[f]])
    np.testing.assert_array_equal(N.Rblock_off[:nf], ref["num_Rblock_off"][:nf])


def numeric_as_ref(S, No) -> dict:
    """oracle Numeric -> the num_* dict layout of the golden fixtures."""
    return {"num_rank": np.array([No.c.rank]), "num_rank1": np.array([No.c.rank1]),
            "num_maxfrank": np.array([No.c.maxfrank]), "num_maxfm": np.array([No.c.maxfm]),
            "num_Rdead": No.Rdead, "num_Hm": No.Hm, "num_Hr": No.Hr, "num_HStair": No.HStair, "num_HPinv": No.HPinv,
            "num_Hii": No.Hii, "num_Rblock_off": No.Rblock_off}


def compare_numeric(orc, S, G, No, g, ftol=1e-10, name=None, backward_tol=1e-13):
    """G: package QRNumeric from the GPU; No: oracle Numeric on the same input; g: fixture dict (inputs)."""
    N = numeric_from_gpu(S, G)
    compare_integers(S, N, numeric_as_ref(S, No))
    got, ref = rrow_sig_all(S, N), rrow_sig_all(S, No)
    assert got.shape == ref.shape
    # (per row: ftol of the row's own norm, or the derived bound TOL_C * eps * cond(R) of the largest row norm -- stmmqr_testlib)
    kappa = cond_probe(orc, S, No)
    ex = rrow_excess(got, ref, kappa, rel=ftol)
    assert ex <= 1.0, (name, ex, kappa)
    if name in ELEMENTWISE:
        assert np.linalg.norm(N.HTau[:S.rjsize] - No.HTau[:S.rjsize]) <= ftol * max(np.linalg.norm(No.HTau), 1.0)
        a, b = N.Stack[:N.c.rh_total], No.Stack[:No.c.rh_total]
        assert np.linalg.norm(a - b) <= ftol * max(np.linalg.norm(b), 1e-300)
    # backward error through the packed factors; rank-deficient factorizations on the live columns (dead pivot columns
    # are dropped by the method itself, to within tol)
    err = aqr_probe_error(orc, S, N, g["in_Ap"], g["in_Ai"], g["in_Ax"], live_only=(N.c.rank != S.n))
    assert err < backward_tol, err
    return N
