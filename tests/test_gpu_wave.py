"""The DPP / v_permlane reduction primitives of the panel kernels (csrc/stmmqr_wave.h), compiled as a stand-alone HIP
program on the GPU box and checked bitwise on integer-valued data (wave_sum, the 8-value halving butterfly, the
lane broadcast)."""
import pathlib
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"


def test_wave_primitives(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "wave_selftest"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-I", str(PKG / "csrc"), str(ROOT / "tests" / "wave_selftest.hip"),
                    "-o", str(exe)], check=True, capture_output=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad=0" in r.stdout
