"""Golden case for the DEFAULT tolerance (QR_DEFAULT_TOL = -2 -> qr_tol, SparseQR.c:126-130): a 160 x 120 matrix with three
duplicated columns and one zero column, and the rank the compiled reference's driver finds for it (oracle/_ref/refdump, whose
driver computes the same tolerance, qrtest.c:135-142).  Run in the build container:  python tests/golden/make_rankdef_golden.py"""
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent.parent
rng = np.random.default_rng(5)
n, m = 120, 160
A = (sp.random(m, n, density=4.0 / n, random_state=3, data_rvs=rng.standard_normal) + sp.eye(m, n) * 2.0).tolil()
for a, b in [(10, 50), (20, 90), (33, 34)]:
    A[:, b] = A[:, a]
A[:, 70] = 0
A = sp.csc_matrix(A); A.sum_duplicates(); A.sort_indices(); A.eliminate_zeros()
Ap, Ai, Ax = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
mtx = "/tmp/rankdef_default_tol.mtx"
cols = np.repeat(np.arange(n), np.diff(Ap))
with open(mtx, "w") as f:
    f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, len(Ax)))
    np.savetxt(f, np.c_[Ai + 1, cols + 1, Ax], fmt="%d %d %.17g")
out = subprocess.run([str(ROOT / "oracle" / "_ref" / "refdump"), mtx, "-1", "1", "d", "-", "1"], capture_output=True, text=True,
                     env=dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL"), timeout=120)
rank = int(re.search(r"rank = (\d+)", out.stdout).group(1))
np.savez_compressed(ROOT / "tests" / "golden" / "api" / "rankdef_default_tol.npz", Ap=Ap, Ai=Ai, Ax=Ax, m=m, n=n, ref_rank=rank,
                    lapack_rank=np.linalg.matrix_rank(A.toarray()))
print("reference rank", rank, "LAPACK rank", np.linalg.matrix_rank(A.toarray()))
