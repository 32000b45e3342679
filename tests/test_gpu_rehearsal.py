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
