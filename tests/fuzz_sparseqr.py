"""Randomized end-to-end check of the whole product path (own singletons / COLAMD / analysis, device factorization, Q'b and
R-solve on the resident factors): least-squares solutions and ranks against dense LAPACK on matrices the fixtures do not
contain -- random sparse + diagonal, 2-D grid differences, matrices with dense rows and columns, rank-deficient ones
(duplicated and zero columns, default tolerance).  TEST INFRASTRUCTURE: `python tests/fuzz_sparseqr.py [seed [iterations]]` on
a GPU box, and a short fixed-seed run in tests/test_gpu_sparseqr.py."""
import importlib, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
pkg = importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
rng = np.random.default_rng(1)
nfail = 0

def run(tag, A, check_dense=True, rankdef=False):
    global nfail
    A = sp.csc_matrix(A); A.sum_duplicates(); A.sort_indices()
    m, n = A.shape
    Ap, Ai, Ax = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
    t0 = time.perf_counter()
    Q = pkg.SparseQR(m, n, Ap, Ai, Ax, ordering=7, tol=-2.0, relax=pkg.relax_for_qr(n, int(Ap[-1])))
    info = Q.info
    b = rng.standard_normal(m)
    x = Q.solve(1, Q.qmult(0, b))[:, 0]
    dt = time.perf_counter() - t0
    r = A @ x - b
    nrm = np.linalg.norm(A.T @ r) / max(np.linalg.norm(A.data) * np.linalg.norm(b), 1e-300)
    msg = f"{tag:28s} m={m:6d} n={n:6d} nnz={A.nnz:8d} rank={int(info['rank']):6d} retries={int(info['retries'])} |A'r|={nrm:.1e} {dt*1e3:7.1f} ms"
    ok = int(info["retries"]) == 0
    if not rankdef:
        ok = ok and int(info["rank"]) == n and nrm < 1e-10
    if check_dense and m * n <= 4_000_000:
        Ad = A.toarray()
        xr, *_ = np.linalg.lstsq(Ad, b, rcond=None)
        rr = np.linalg.norm(Ad @ xr - b)
        rel = abs(np.linalg.norm(r) - rr) / max(rr, 1e-300)
        msg += f" |r|-|r_lapack| rel={rel:.1e}"
        if rankdef:
            msg += f" rank_np={np.linalg.matrix_rank(Ad)}"
            ok = ok and rel < 1e-8 and int(info["rank"]) == np.linalg.matrix_rank(Ad)
        else:
            ok = ok and rel < 1e-9 and np.linalg.norm(x - xr) <= 1e-7 * max(np.linalg.norm(xr), 1)
    if not ok:
        nfail += 1
        msg += "   <<<<<< FAIL"
    print(msg, flush=True)
    Q.close()

def grid2d(k, extra):
    n = k * k
    idx = np.arange(n).reshape(k, k)
    rows, cols, vals = [], [], []
    e = 0
    for di, dj in ((0, 1), (1, 0)):
        a = idx[: k - di, : k - dj].ravel(); b_ = idx[di:, dj:].ravel()
        for (p, q) in ((a, b_),):
            rows += list(range(e, e + len(p))) * 2; cols += list(p) + list(q); vals += [1.0] * len(p) + [-1.0] * len(p); e += len(p)
    A = sp.coo_matrix((vals, (rows, cols)), shape=(e, n))
    return sp.vstack([A, sp.eye(n) * 0.1, sp.random(extra, n, density=3.0 / n, random_state=int(rng.integers(1 << 30)))]).tocsc()



def main(seed=1, iters=12, big=True):
    """-> number of failed cases"""
    global rng, nfail
    rng = np.random.default_rng(seed)
    nfail = 0
    for it in range(iters):
        n = int(rng.integers(50, 1500)); m = n + int(rng.integers(0, n))
        d = float(rng.uniform(1.5, 8.0)) / n
        A = sp.random(m, n, density=d, random_state=int(rng.integers(1 << 30)), data_rvs=rng.standard_normal) + sp.eye(m, n) * (1.0 + rng.random())
        run("random+diag", A)
        k = int(rng.integers(12, 60))
        run(f"grid2d k={k}", grid2d(k, int(rng.integers(0, 50))))
        # a few dense rows and columns
        n = int(rng.integers(200, 1200)); m = n + int(rng.integers(10, 300))
        A = sp.random(m, n, density=3.0 / n, random_state=int(rng.integers(1 << 30)), data_rvs=rng.standard_normal).tolil()
        for r_ in rng.integers(0, m, 3): A[r_, :] = rng.standard_normal(n)
        for c_ in rng.integers(0, n, 2): A[:, c_] = rng.standard_normal((m, 1))
        run("dense rows/cols", sp.csc_matrix(A) + sp.eye(m, n) * 2.0)
        # rank deficient: duplicated and zero columns
        n = int(rng.integers(60, 600)); m = n + int(rng.integers(0, 200))
        A = (sp.random(m, n, density=4.0 / n, random_state=int(rng.integers(1 << 30)), data_rvs=rng.standard_normal) + sp.eye(m, n) * 2.0).tolil()
        for _ in range(3):
            a, b_ = rng.integers(0, n, 2)
            A[:, b_] = A[:, a]
        A[:, int(rng.integers(0, n))] = 0
        run("rank deficient", sp.csc_matrix(A), rankdef=True)
    if big:                                               # larger ones: residual only
        run("grid2d k=140", grid2d(140, 200), check_dense=False)
        n = 30000
        A = sp.random(45000, n, density=4.0 / n, random_state=7, data_rvs=rng.standard_normal) + sp.eye(45000, n) * 3.0
        run("random 45000x30000", A, check_dense=False)
    print("FAILURES:", nfail)
    return nfail



if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 12) else 0)
