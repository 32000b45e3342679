"""The host-only pieces of the library (Matrix-Market reader, symbolic analysis, COLAMD, the host half of SparseQR()) under
AddressSanitizer + UndefinedBehaviorSanitizer.  CPU build only -- GPU sanitizers are not available on the pool --: the four
sources are plain C++ and are compiled with g++ together with tests/sanitize_host.cpp (which stubs the device entry points that
are never called).  Inputs: every file of tests/golden/mm (hostile ones included), hostile files made here, and the committed
fixtures written as Matrix-Market files (bcsstk14 with 40 column singletons, the rank-deficient lns_3937, epb1, ...)."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from stmmqr_testlib import golden_names, load_golden

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "stm-multifrontal-qr-factorization-empowered-by-gcn_amd" / "csrc"
SOURCES = ["stmmqr_mmio.cpp", "stmmqr_symbolic.cpp", "stmmqr_colamd.cpp", "stmmqr_sparseqr.cpp"]

HOSTILE = {
    "empty.mtx": "",
    "banner_only.mtx": "%%MatrixMarket matrix coordinate real general\n",
    "truncated.mtx": "%%MatrixMarket matrix coordinate real general\n4 4 6\n1 1 1.0\n2 2 2.0\n",
    "index_out_of_range.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n7 9 2.0\n",
    "negative_index.mtx": "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1.0\n-2 1 2.0\n",
    "huge_dims.mtx": "%%MatrixMarket matrix coordinate real general\n99999999999 99999999999 1\n1 1 1.0\n",
    "zero_by_zero.mtx": "%%MatrixMarket matrix coordinate real general\n0 0 0\n",
    "empty_columns.mtx": "%%MatrixMarket matrix coordinate real general\n5 6 2\n1 1 1.0\n5 6 2.0\n",
    "wide_dense_row.mtx": "%%MatrixMarket matrix coordinate pattern general\n2 40 40\n" + "".join(f"1 {j}\n" for j in range(1, 41)),
    "garbage.mtx": "%%MatrixMarket matrix coordinate real general\nfoo bar baz\n1 x 2\n",
    "array_format.mtx": "%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n",
}


def _write_mtx(path, g):
    Ap, Ai, Ax = (g["A_p"], g["A_i"], g["A_x"]) if "A_p" in g else (g["in_Ap"], g["in_Ai"], g["in_Ax"])
    m, n = (int(g["A_m"][0]), int(g["A_n"][0])) if "A_m" in g else (int(g["in_m"][0]), int(g["in_n"][0]))
    cols = np.repeat(np.arange(n), np.diff(Ap))
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, len(Ax)))
        np.savetxt(f, np.c_[Ai + 1, cols + 1, Ax], fmt="%d %d %.17g")


@pytest.mark.timeout(900)
def test_host_code_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "sanitize_host"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-o", str(exe)] + [str(CSRC / s) for s in SOURCES] + [str(ROOT / "tests" / "sanitize_host.cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "asan" in b.stderr.lower() and "cannot find" in b.stderr.lower():
        pytest.skip("the sanitizer runtimes are not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    files = sorted(str(p) for p in (ROOT / "tests" / "golden" / "mm").glob("*.mtx"))
    for name, text in HOSTILE.items():
        (tmp_path / name).write_text(text)
        files.append(str(tmp_path / name))
    # fixtures small enough for a sanitized run of seconds (the reader parses text: 60 000 entries for bcsstk14)
    for name in ("bcsstk14", "lns_3937", "epb1", "syn_rankdef_grid", "syn_emptycol", "syn_dupcol", "syn_wide5x8", "syn_star", "grid20_standin"):
        if name in golden_names(True):
            _write_mtx(tmp_path / f"{name}.mtx", load_golden(name))
            files.append(str(tmp_path / f"{name}.mtx"))
    r = subprocess.run([str(exe)] + files, capture_output=True, text=True, timeout=800,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    nread = int(last.split()[1].rstrip(","))
    assert nread >= 12 + 5                        # the well-formed golden files and the fixtures were really analysed
