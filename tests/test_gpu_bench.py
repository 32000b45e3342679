"""bench.py on a Matrix Market file (round-3 verdict item 2): reader -> own symbolic phase -> timed numeric factorization; the flop
count of the line must be the compiled reference's for the same matrix (fixture), and the CPU leg runs the reference on that file."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(args, env=None, timeout=600):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                         env=env or dict(os.environ))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("name,cpu", [("epb1", True), ("cvxqp3", False)])
def test_bench_on_a_matrix_market_file(tmp_path, name, cpu):
    sys.path.insert(0, str(ROOT))
    import bench
    from stmmqr_testlib import load_golden, scalar
    g0 = load_golden(name)
    p = tmp_path / f"{name}.mtx"
    bench._write_mtx(p, int(g0["A_m"][0]), int(g0["A_n"][0]), g0["A_p"], g0["A_i"], g0["A_x"])
    d = _bench(["--matrix", str(p), "--steps", "2", "--warmup", "1"] + ([] if cpu else ["--no-cpu"]))
    assert d["config"]["flops_per_step"] == scalar(g0, "flopcount")
    assert d["config"]["retries"] == 0 and d["value"] > 0 and "Matrix Market file" in d["config"]["workload"]
    assert d["config"]["matrix_file"]["n1cols"] == scalar(g0, "n1cols")
    if cpu and (ROOT / "oracle" / "_ref" / "refdump").exists():
        cb = d["cpu_baseline"]
        assert cb["kind"] == "reference" and cb["flops_match_device"] is True
        assert cb["legs"][0]["best_of"] == 3


def test_data_dir_replaces_the_standin(tmp_path):
    """xenon1 / sme3Dc / 3D_51448_3D are read from $STMMQR_DATA_DIR when present (stand-in otherwise).  A small file
    under the real name stands for the absent matrix here: the line must say it ran the file, not the stand-in."""
    sys.path.insert(0, str(ROOT))
    import bench
    from stmmqr_testlib import load_golden, scalar
    g0 = load_golden("t2d_q9")
    bench._write_mtx(tmp_path / "xenon1.mtx", int(g0["A_m"][0]), int(g0["A_n"][0]), g0["A_p"], g0["A_i"], g0["A_x"])
    d = _bench(["--steps", "1", "--warmup", "1", "--no-cpu"], env=dict(os.environ, STMMQR_DATA_DIR=str(tmp_path)))
    assert d["config"]["flops_per_step"] == scalar(g0, "flopcount")
    assert "xenon1.mtx" in d["config"]["workload"] and "stand-in" not in d["config"]["workload"]
