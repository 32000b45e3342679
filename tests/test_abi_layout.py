"""The byte offsets compiled into the library for the reference's sparse_common equal offsetof() in the real header
(needs /root/reference: build container only)."""
import ctypes as C
import importlib
import subprocess
import tempfile
from pathlib import Path

import pytest

REF = Path("/root/reference/STMMQR")
pytestmark = pytest.mark.skipif(not REF.is_dir(), reason="reference checkout not present")

FIELDS = ["status", "malloc_count", "memory_usage", "memory_inuse", "blas_ok", "SPQR_grain", "SPQR_small",
          "SPQR_shrink", "SPQR_flopcount", "SPQR_flopcount_bound"]


def test_common_layout_matches_reference_header():
    pkg = importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"SparseQR.h\"\nint main(){\n" + \
          "".join(f'printf("%zu\\n", offsetof(sparse_common,{f}));\n' for f in FIELDS) + \
          'printf("%zu %zu %zu\\n", sizeof(sparse_csc), sizeof(qr_symbolic), sizeof(qr_numeric));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "p.c").write_text(src)
        subprocess.check_call(["gcc", "-std=gnu99", "-fcommon", "-w", "-DDLONG", f"-I{REF}/include", f"-I{REF}/include/tpsm",
                               str(Path(td) / "p.c"), "-o", str(Path(td) / "p")])
        out = subprocess.check_output([str(Path(td) / "p")]).decode().split()

    class L(C.Structure):
        _fields_ = [(f, C.c_size_t) for f in FIELDS]
    lay = L()
    pkg.lib.stmmqr_get_common_layout(C.byref(lay))
    assert [getattr(lay, f) for f in FIELDS] == [int(x) for x in out[:len(FIELDS)]]
    assert [int(x) for x in out[len(FIELDS):]] == [88, 272, 192]
