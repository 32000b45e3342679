#!/usr/bin/env python3
"""Generate tests/golden/mm/reference_reader.npz: what the REFERENCE's reader (SparseCore_read_matrix with prefer = 1, as
test/qrtest.c:112 calls it) makes of every tests/golden/mm/*.mtx.  Build container only (needs oracle/_ref/refdump, which
records the matrix as read under the names A_m, A_n, A_p, A_i, A_x):
    make -C oracle ref && python tests/golden/make_mm_golden.py
Files the reference refuses are recorded as <name>_refused = 1 (the product's reader must refuse them too)."""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_golden import REFDUMP, parse_dump  # noqa: E402


def main():
    if not REFDUMP.exists():
        sys.exit("build the reference first: make -C oracle ref")
    out = {}
    env = {"MKL_THREADING_LAYER": "SEQUENTIAL", "PATH": "/usr/bin:/bin"}
    for mtx in sorted((HERE / "mm").glob("*.mtx")):
        name = mtx.stem
        with tempfile.TemporaryDirectory() as td:
            binp = Path(td) / "dump.bin"
            r = subprocess.run([str(REFDUMP), str(mtx), "-1", "1", "d", str(binp), "1"], capture_output=True, text=True, env=env)
            d = parse_dump(binp) if binp.exists() else {}
        if "A_p" not in d:
            out[f"{name}_refused"] = np.array([1])
            print(f"{name:28s} refused by the reference (rc {r.returncode})")
            continue
        for k in ("m", "n", "p", "i", "x"):
            key = {"m": "m", "n": "n", "p": "Ap", "i": "Ai", "x": "Ax"}[k]
            out[f"{name}_{key}"] = d[f"A_{k}"]
        print(f"{name:28s} {int(d['A_m'][0])} x {int(d['A_n'][0])}  nnz {len(d['A_x'])}")
    np.savez_compressed(HERE / "mm" / "reference_reader.npz", **out)


if __name__ == "__main__":
    main()
