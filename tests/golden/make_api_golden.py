#!/usr/bin/env python3
"""Generate tests/golden/api/api_reference.npz: the outputs of the REFERENCE's public API (SparseQR -> QR_qmult x 4 methods ->
QR_solve x 4 systems, oracle/refapi.c) on the small fixtures, from the compiled reference with nothing interposed.
Build container only:   make -C oracle ref && python tests/golden/make_api_golden.py
tests/test_relinked_reference.py runs the same program linked as INTEGRATION.md 1 prescribes (reference minus
SparseQR_factorize.o / SparseQR_multithreads.o, plus libstmmqr_hip.so) against these."""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
from make_golden import parse_dump  # noqa: E402

REFAPI = HERE.parent.parent / "oracle" / "_ref" / "refapi"
CASES = [("bcsstk14", -1), ("epb1", -1), ("epb1", 0), ("syn_grid3d", -1), ("syn_dupcol", -1), ("syn_rankdef_grid", -1),
         ("syn_wide5x8", -1), ("syn_star", -1), ("syn_chain", -1), ("syn_rand60x40", -1), ("lns_3937", -1)]


def write_mtx(path, g):
    Ap, Ai, Ax = g["A_p"], g["A_i"], g["A_x"]
    m, n = int(g["A_m"][0]), int(g["A_n"][0])
    cols = np.repeat(np.arange(n), np.diff(Ap))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, len(Ax)))
        np.savetxt(f, np.c_[Ai + 1, cols + 1, Ax], fmt="%d %d %.17g")


def run_refapi(exe, mtx, ordering, env=None, timeout=600):
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "api.bin"
        r = subprocess.run([str(exe), str(mtx), str(ordering), str(out)], capture_output=True, text=True, env=env, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError(f"{exe.name} failed ({r.returncode})\n{r.stdout}\n{r.stderr}")
        return parse_dump(out), r.stdout


def main():
    if not REFAPI.exists():
        sys.exit("build the reference first: make -C oracle ref")
    from stmmqr_testlib import load_golden
    out = {}
    env = {"MKL_THREADING_LAYER": "SEQUENTIAL", "PATH": "/usr/bin:/bin"}
    with tempfile.TemporaryDirectory() as td:
        for name, ordering in CASES:
            g = load_golden(name)
            if "A_x" not in g:
                print(f"{name}: no A in the fixture, skipped")
                continue
            mtx = Path(td) / f"{name}.mtx"
            write_mtx(mtx, g)
            d, text = run_refapi(REFAPI, mtx, ordering, env)
            key = f"{name}@{ordering}"
            for k, v in d.items():
                out[f"{key}:{k}"] = v
            print(f"{key:24s} m={d['m'][0]} n={d['n'][0]} rank={d['rank'][0]} {text.strip().splitlines()[0]}")
    np.savez_compressed(HERE / "api" / "api_reference.npz", **out)
    print("wrote", HERE / "api" / "api_reference.npz", (HERE / "api" / "api_reference.npz").stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
