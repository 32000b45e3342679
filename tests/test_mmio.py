"""The driver's Matrix Market reader (stmmqr_read_matrix_market, csrc/stmmqr_mmio.cpp) against what the
REFERENCE's reader (SparseCore_read_matrix, prefer = 1, as test/qrtest.c:112 calls it) makes of the same files.
The expected arrays in tests/golden/mm/reference_reader.npz were dumped from the compiled reference (oracle/_ref/refdump);
the two matrices of the reference's Data directory are compared with the A arrays of their golden fixtures when the
checkout is present.  No GPU needed."""
import importlib
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
MM = ROOT / "tests" / "golden" / "mm"
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"
NAMES = sorted(p.stem for p in MM.glob("*.mtx"))


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module(PKG)


@pytest.mark.parametrize("name", NAMES)
def test_reader_matches_reference_reader(pkg, name):
    ref = np.load(MM / "reference_reader.npz")
    if f"{name}_refused" in ref:                 # the reference's reader refuses the file: so must this one
        with pytest.raises(pkg.StmmqrError):
            pkg.read_matrix_market(MM / f"{name}.mtx")
        return
    m, n, Ap, Ai, Ax = pkg.read_matrix_market(MM / f"{name}.mtx")
    assert (m, n) == (int(ref[f"{name}_m"][0]), int(ref[f"{name}_n"][0]))
    np.testing.assert_array_equal(Ap, ref[f"{name}_Ap"])
    np.testing.assert_array_equal(Ai, ref[f"{name}_Ai"])
    np.testing.assert_array_equal(Ax, ref[f"{name}_Ax"])            # bit exact: copies and one sum of two file values


def test_reader_refuses_what_the_driver_refuses(pkg, tmp_path):
    dense = tmp_path / "dense.mtx"
    dense.write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(pkg.StmmqrError, match="sparse"):
        pkg.read_matrix_market(dense)
    cplx = tmp_path / "c.mtx"
    cplx.write_text("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 2.0\n")
    with pytest.raises(pkg.StmmqrError, match="complex"):
        pkg.read_matrix_market(cplx)
    oob = tmp_path / "o.mtx"
    oob.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n")
    with pytest.raises(pkg.StmmqrError, match="range"):
        pkg.read_matrix_market(oob)
    with pytest.raises(pkg.StmmqrError):
        pkg.read_matrix_market(tmp_path / "missing.mtx")


def test_reader_survives_hostile_headers(pkg, tmp_path):
    """Nothing is thrown across the C ABI and nothing is written out of bounds: absurd entry counts, a symmetric banner
    on a rectangular file whose mirror entries would fall outside, truncated files."""
    cases = {
        "huge.mtx": "2 2 99999999999999\n1 1 1.0\n",
        "neg.mtx": "%%MatrixMarket matrix coordinate real general\n2 2 -5\n1 1 1.0\n",
        "trunc.mtx": "%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 1.0\n",
        "nan.mtx": "nan nan nan\n",
    }
    for fn, text in cases.items():
        (tmp_path / fn).write_text(text)
        with pytest.raises(pkg.StmmqrError):
            pkg.read_matrix_market(tmp_path / fn)
    ok = tmp_path / "tall.mtx"
    ok.write_text("%%MatrixMarket matrix coordinate real symmetric\n6 2 3\n1 1 1\n5 1 2\n6 2 3\n")
    m, n, Ap, Ai, Ax = pkg.read_matrix_market(ok)
    assert (m, n) == (6, 2)
    np.testing.assert_array_equal(Ap, [0, 2, 3])
    np.testing.assert_array_equal(Ai, [0, 4, 5])


@pytest.mark.parametrize("name,fname", [("bcsstk14", "bcsstk14.mtx"), ("epb1", "epb1.mtx")])
def test_reference_data_files(pkg, name, fname):
    """bcsstk14.mtx is stored symmetric (lower triangle): the expansion must give the A the reference factorizes."""
    path = Path("/root/reference/Data") / fname
    if not path.exists():
        pytest.skip("reference checkout not present")
    from stmmqr_testlib import load_golden
    g = load_golden(name)
    m, n, Ap, Ai, Ax = pkg.read_matrix_market(path)
    assert (m, n) == (int(g["A_m"][0]), int(g["A_n"][0]))
    np.testing.assert_array_equal(Ap, g["A_p"])
    np.testing.assert_array_equal(Ai, g["A_i"])
    np.testing.assert_array_equal(Ax, g["A_x"])
