"""The link recipe of INTEGRATION.md 1, tested as documented: oracle/_ref/refapi_relinked is the REFERENCE compiled without
SparseQR_factorize.o and SparseQR_multithreads.o and linked against libstmmqr_hip.so (plus the binding stub of
INTEGRATION.md 2).  In it the reference's SparseQR() calls this repository's qr_factorize (SparseQR.c:349,371), its
qr_panel calls this repository's qr_larftb with ALL FOUR methods (SparseQR.c:1659,1663) and its qr_freenum releases what
the library allocated.  The program (oracle/refapi.c) runs QR_qmult x {QTX, QX, XQT, XQ} and QR_solve x 4 systems; its
outputs are compared with those of the same program linked against the pure reference (tests/golden/api/api_reference.npz,
generator tests/golden/make_api_golden.py).  Needs a GPU and the prebuilt oracle/_ref (travels with the snapshot)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "golden"))
RELINKED = ROOT / "oracle" / "_ref" / "refapi_relinked"
GOLD = ROOT / "tests" / "golden" / "api" / "api_reference.npz"
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not RELINKED.exists(), reason="oracle/_ref/refapi_relinked not built")]

# fixtures whose factorization has no rounding-noise pivots (tests/parity.py ELEMENTWISE): every output is
# determined and compared entry by entry; elsewhere Householder vectors of noise rows are arbitrary rotations (MKL and a
# restated LAPACK already differ there), so only what IS determined is compared
EXACT = {"syn_star", "syn_chain", "syn_rand60x40", "syn_wide5x8", "syn_dupcol"}
CASES = [("bcsstk14", -1), ("epb1", -1), ("epb1", 0), ("syn_grid3d", -1), ("syn_dupcol", -1), ("syn_rankdef_grid", -1),
         ("syn_wide5x8", -1), ("syn_star", -1), ("syn_chain", -1), ("syn_rand60x40", -1), ("lns_3937", -1)]


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("name,ordering", CASES)
def test_relinked_reference_api(tmp_path, name, ordering):
    from make_api_golden import run_refapi, write_mtx
    from stmmqr_testlib import load_golden
    gold = np.load(GOLD)
    key = f"{name}@{ordering}"
    ref = {k.split(":", 1)[1]: gold[k] for k in gold.files if k.startswith(key + ":")}
    assert ref, "no golden outputs for " + key
    g = load_golden(name)
    mtx = tmp_path / "a.mtx"
    write_mtx(mtx, g)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    got, text = run_refapi(RELINKED, mtx, ordering, env, timeout=300)
    m, n, nr = int(ref["m"][0]), int(ref["n"][0]), 3
    # (malloc_count / memory_inuse at exit: qr_freenum released everything the library allocated -- what is left is the
    #  common workspace, the same count and the same bytes as in the pure reference's run)
    for k in ("m", "n", "rank", "n1rows", "n1cols", "status_after_factorize", "status_end", "malloc_count_exit", "memory_inuse_exit"):
        assert int(got[k][0]) == int(ref[k][0]), k
    for meth in range(4):
        assert int(got[f"qmult_{meth}_status"][0]) == 0             # (a failing qr_larftb seam leaves cc->status < 0)
    X = got["qmult_x"].reshape(nr, m).T
    Y = [got[f"qmult_{k}"] for k in range(4)]
    Y0, Y1 = Y[0].reshape(nr, m).T, Y[1].reshape(nr, m).T           # m x nr (column-major in the file)
    Y2, Y3 = Y[2].reshape(m, nr), Y[3].reshape(m, nr)                # nr x m column-major = m x nr row-major view
    # -- properties that hold whatever the signs of the reflectors: Q orthogonal, the four methods consistent
    assert rel(got["qmult_10"], got["qmult_x"]) < 1e-13             # Q (Q'X) = X        (methods 0 and 1)
    for Yk in (Y0, Y1):
        assert abs(np.linalg.norm(Yk) - np.linalg.norm(X)) <= 1e-13 * np.linalg.norm(X)
    assert rel(Y3, Y0) < 1e-13                                       # X'Q  = (Q'X)'      (method 3 against 0)
    assert rel(Y2, Y1) < 1e-13                                       # X'Q' = (Q X)'      (method 2 against 1)
    # -- the driver's acceptance flow (x = E R \\ Q'b with b = A [0..n-1]) against the pure reference's solution
    full = int(ref["rank"][0]) == min(m, n) and m >= n
    if full:
        assert rel(got["driver_x"], ref["driver_x"]) < 1e-7
    # -- determined factorizations: every output of the pure reference, entry by entry
    if name in EXACT:
        for k in ("qmult_0", "qmult_1", "qmult_2", "qmult_3", "solve_0", "solve_1", "solve_2", "solve_3"):
            assert rel(got[k], ref[k]) < 1e-8, (k, rel(got[k], ref[k]))
