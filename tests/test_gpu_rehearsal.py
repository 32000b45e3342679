"""bench.py --gpus 2 as REAL processes on a one-GPU box (STMMQR_BENCH_REHEARSAL=1: both ranks on device 0, gloo, contribution
blocks through the host): spawn before any GPU call, rendezvous, tree-of-joins partition, phase loop with the exchange,
max-over-ranks timing, ONE JSON line from rank 0.  What it cannot cover is RCCL itself (tests/test_gpu_multi.py, two GPUs)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("workload,nranks", [("grid20_standin", 2), ("epb1", 4)])
def test_bench_spawns_its_ranks_and_shards(workload, nranks):
    env = dict(os.environ, STMMQR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(nranks), "--workload", workload, "--steps", "2",
                          "--warmup", "1", "--no-cpu"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == nranks and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "REHEARSAL" in d["config"]["parallelism"] and f"x{nranks}" in d["config"]["parallelism"]
    assert 0 < d["config"]["critical_path_flop_share"] <= 1.0


def test_a_failed_rank_gives_the_line_not_a_hang():
    """first contact with a multi-GPU node must fail loudly: rank 1 dies after the rendezvous (STMMQR_BENCH_FAIL_RANK), the
    launcher terminates the others, and rank 0 still prints the contract's ONE JSON line -- value null, the reason in `error` --
    with a non-zero exit code, in seconds"""
    import time
    env = dict(os.environ, STMMQR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", STMMQR_BENCH_FAIL_RANK="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "grid20_standin", "--steps", "1",
                          "--warmup", "1", "--no-cpu"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert out.returncode != 0
    assert time.monotonic() - t0 < 200
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "error" in d
