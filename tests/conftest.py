import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference; build container only)")


@pytest.fixture(scope="session")
def oracle():
    from stmmqr_testlib import Oracle
    return Oracle()


def _install_option_policy():
    """Tests that move big_front_cols away from its default do so to drive small fixtures through the large-front kernels (panel
    pipeline, Gram panel, row-parallel update): the mid-front kernel (options.mid_front_cols, one workgroup per front) would take
    those fronts away from them, so a test that does not name mid_front_cols itself always runs with it off (the library's default) (tests/test_gpu_factorize.py::test_mid_fronts_everywhere does)."""
    import importlib
    try:
        pkg = importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
    except Exception:
        return
    if getattr(pkg, "_option_policy_installed", False):
        return
    inner = pkg.set_options

    def set_options(**kw):
        if "big_front_cols" in kw and "mid_front_cols" not in kw:
            kw["mid_front_cols"] = 0
        return inner(**kw)

    pkg.set_options = set_options
    pkg._option_policy_installed = True


_install_option_policy()
