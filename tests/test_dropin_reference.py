"""Drop-in check with the REAL reference around the seam: oracle/_ref/refdump runs the reference's own driver flow
(SparseQR -> QR_qmult -> QR_solve -> SparseQR_free, STMMQR/test/qrtest.c:11-53,180-204) while the interposed
qr_factorize is forwarded to libstmmqr_hip.so.  The solve residual must match the reference's own run.
Needs a GPU and the prebuilt oracle/_ref (it travels with the gpurun snapshot)."""
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
REFDUMP = ROOT / "oracle" / "_ref" / "refdump"
HIPLIB = ROOT / "stm-multifrontal-qr-factorization-empowered-by-gcn_amd" / "libstmmqr_hip.so"
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not REFDUMP.exists(), reason="oracle/_ref not built")]


def run_child(args, env, timeout, cwd=None):
    """Run a child with a time limit; when the limit passes, say WHERE it sat before killing it.  (Round 3: this child -- normally
    0.25 s -- once did not finish in 120 s, in the THIRD parametrization of a session whose first two had just run the same binary,
    so "the box was still paging MKL in" does not explain it; the limit was raised and nothing was learnt.  The limit is back at
    120 s and a timeout now reports the child's partial output -- refdump prints "seam routed to", the analysis / factorization
    times and the residual as it goes -- and every thread's kernel wait channel and state, which tell MKL initialisation, dlopen,
    hipInit, the factorization and the exit handlers apart.)"""
    import time
    p = subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=cwd)
    try:
        so, se = p.communicate(timeout=timeout)
        return subprocess.CompletedProcess(args, p.returncode, so, se)
    except subprocess.TimeoutExpired:
        where = []
        for t in sorted(Path(f"/proc/{p.pid}/task").glob("*")):
            rec = [t.name]
            for f in ("comm", "wchan", "stat", "syscall"):
                try:
                    txt = (t / f).read_text().strip()
                    rec.append(f + "=" + (" ".join(txt.split()[:3]) if f == "stat" else txt))
                except OSError:
                    pass
            where.append(" ".join(rec))
        try:
            maps = [ln.split()[-1] for ln in Path(f"/proc/{p.pid}/maps").read_text().splitlines() if ".so" in ln]
            libs = sorted({Path(m).name for m in maps})
        except OSError:
            libs = []
        p.kill()
        so, se = p.communicate()
        raise AssertionError(f"child {args[0]} did not finish in {timeout} s\n--- stdout so far ---\n{so}\n--- stderr so far ---\n{se}"
                             f"\n--- threads (tid comm wchan stat syscall) ---\n" + "\n".join(where)
                             + "\n--- shared objects mapped ---\n" + " ".join(libs))


def write_mtx(path, g):
    Ap, Ai, Ax = g["A_p"], g["A_i"], g["A_x"]
    m, n = int(g["A_m"][0]), int(g["A_n"][0])
    cols = np.repeat(np.arange(n), np.diff(Ap))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, len(Ax)))
        np.savetxt(f, np.c_[Ai + 1, cols + 1, Ax], fmt="%d %d %.17g")


@pytest.mark.parametrize("name", ["bcsstk14", "epb1", "syn_grid3d", "syn_dupcol", "syn_star"])
def test_reference_driver_on_hip_factorization(tmp_path, name):
    from stmmqr_testlib import load_golden, scalar
    g = load_golden(name)
    mtx = tmp_path / "a.mtx"
    write_mtx(mtx, g)
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", REFDUMP_HIPLIB=str(HIPLIB))
    out = run_child([str(REFDUMP), str(mtx), "-1", "1", "d", "-", "1"], env=env, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "seam routed to" in out.stdout
    res = float(re.search(r"res =\s*([0-9.eE+-]+)", out.stdout).group(1))
    bwd = float(re.search(r"backward =\s*([0-9.eE+-]+)", out.stdout).group(1))
    rank = int(re.search(r"rank = (\d+)", out.stdout).group(1))
    assert rank == scalar(g, "QR_rank")
    ref_res = scalar(g, "res")
    if scalar(g, "QR_rank") == scalar(g, "A_n"):
        assert bwd < 1e-13
        assert res <= max(10 * ref_res, 1e-9)


DRIVER = ROOT / "stm-multifrontal-qr-factorization-empowered-by-gcn_amd" / "stmmqr_qrtest"
REFLIB = ROOT / "oracle" / "_ref" / "libstmmqr_ref.so"


@pytest.mark.parametrize("name,ordering", [("epb1", "0"), ("syn_grid3d", "0"), ("syn_grid3d", "2")])
def test_qrtest_driver_with_reference_orderings(tmp_path, name, ordering):
    """stmmqr_qrtest with an ordering that is a third-party package of the reference (0 AMD, 2 METIS): the reference library
    named with --reflib provides SparseQR() (ordering + analysis) around this library's qr_factorize / qr_larftb, loaded as
    INTEGRATION.md describes.  (Default / COLAMD runs need no reference at all: tests/test_gpu_sparseqr.py.)"""
    if not DRIVER.exists() or not REFLIB.exists():
        pytest.skip("driver or reference library not built")
    from stmmqr_testlib import load_golden, scalar
    g = load_golden(name)
    mtx = tmp_path / "a.mtx"
    write_mtx(mtx, g)
    (tmp_path / "Results").mkdir()
    args = [str(DRIVER), str(mtx), "42", ordering, f"--reflib={REFLIB}"]
    env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL")
    out = run_child(args, env=env, timeout=180, cwd=tmp_path)
    assert out.returncode == 0, out.stdout + out.stderr
    m, n, nnz = int(g["A_m"][0]), int(g["A_n"][0]), len(g["A_x"])
    assert "Matrix %6d-by-%-6d nnz: %6d" % (m, n, nnz) in out.stdout
    assert re.search(r"SparseQR TOTAL time: [0-9.]+", out.stdout)
    res = float(re.search(r"res =\s*([0-9.eE+-]+)", out.stdout).group(1))
    rec = (tmp_path / "Results" / "QR_Time.txt").read_text().split()
    assert rec[0] == "42" and len(rec) == 5 and float(rec[4]) == res
    assert float(rec[2]) > 0                                        # Fac_time of the interposed factorization
    assert res <= 1e-9
