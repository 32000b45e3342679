"""Host-side pieces of bench.py that need no GPU: the --matrix workload builder (reader + own symbolic phase -> the same
qr_symbolic and the same matrix at the seam as the compiled reference recorded in the fixture) and the CPU-baseline legs."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"


@pytest.mark.parametrize("name", ["epb1", "bcsstk14", "dwt_992", "cvxqp3"])
def test_matrix_workload_equals_the_fixture(tmp_path, name):
    """bench.py --matrix F.mtx: what it hands to the timed numeric factorization is what the reference's driver handed to
    qr_factorize on the same file (fixture = refdump's record): the matrix after singleton removal, ntol, the whole qr_symbolic;
    tol to the last bits of a 2-norm"""
    import bench
    from stmmqr_testlib import Symbolic, load_golden, scalar
    pkg = importlib.import_module(PKG)
    g0 = load_golden(name)
    p = tmp_path / f"{name}.mtx"
    bench._write_mtx(p, int(g0["A_m"][0]), int(g0["A_n"][0]), g0["A_p"], g0["A_i"], g0["A_x"])
    g, meta = bench.matrix_workload(pkg, p, "default")
    S, S0 = Symbolic(g), Symbolic(g0)
    assert S.sc == S0.sc
    for k, a in S0.arr.items():
        if a is None or k in ("Fm", "Cm"):
            continue
        np.testing.assert_array_equal(S.arr[k], a, err_msg=k)
    for k in ("Fm", "Cm"):
        np.testing.assert_array_equal(S.arr[k][:S.nf], S0.arr[k][:S.nf])
    for k in ("in_Ap", "in_Ai", "in_Ax"):
        np.testing.assert_array_equal(g[k], g0[k])
    assert scalar(g, "in_ntol") == scalar(g0, "in_ntol")
    assert abs(scalar(g, "in_tol") - scalar(g0, "in_tol")) <= 4e-16 * scalar(g0, "in_tol")
    assert meta["n1cols"] == scalar(g0, "n1cols")


def test_other_orderings_are_refused():
    import bench
    pkg = importlib.import_module(PKG)
    with pytest.raises(SystemExit, match="third-party"):
        bench.matrix_workload(pkg, "nowhere.mtx", "metis")


@pytest.mark.skipif(not (ROOT / "oracle" / "_ref" / "refdump").exists(), reason="oracle/_ref not built")
def test_cpu_baseline_legs():
    """every leg is either measured on THIS host (best of 3 below 20 s, sized from the host's own first run) or reported with the
    reason it failed -- the TPSM leg of the reference as compiled in place dies in TPSM_Numa_SearchNodeSequence on a host without
    the 4 NUMA nodes of its checked-in Numainfo.h and must say so rather than vanish"""
    import bench
    from stmmqr_testlib import load_golden
    g = load_golden("syn_grid3d")
    cb = bench.cpu_baseline("syn_grid3d", g)
    assert cb["kind"] == "reference" and cb["value"] > 0
    legs = cb["legs"]
    assert legs[0]["cores"] == 1 and legs[0]["best_of"] == 3 and legs[0]["seconds"] > 0
    if bench.host_cores() > 1:
        assert len(legs) == 3 and "TPSM" in legs[2]["mode"]
        assert ("value" in legs[2]) != ("error" in legs[2])
        for l in legs:
            assert ("FAILED" in cb["sample"]) == any("error" in x for x in legs)
