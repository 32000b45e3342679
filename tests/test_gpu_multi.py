"""Two (or four) processes, each with a REAL HIP plan on its own GPU, run one sharded factorization (tree of joins, contribution
blocks device to device over RCCL) and the merged result must be bit-identical to the unsharded factorization on one GPU.
Needs >= 2 GPUs: skipped on the one-GPU test box (the orchestration itself is covered on CPU by tests/test_sharded_cpu.py
with gloo, and the device-to-device export / import by tests/test_gpu_sharded.py on one GPU)."""
import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"


def _worker(rank, world, port, name, spread, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from stmmqr_testlib import Symbolic, finish_ranks, load_golden, scalar
    pkg = importlib.import_module(PKG)
    sh = importlib.import_module(PKG + ".sharded")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    done = False
    try:
        g = load_golden(name)
        S = Symbolic(g)
        sym = {**S.sc, **{k: v for k, v in S.arr.items() if v is not None}}
        tol, ntol = scalar(g, "in_tol"), int(scalar(g, "in_ntol"))
        plan = pkg.HipQR(sym, device=rank)
        comm = sh.Comm(dist, dev)
        sp = None
        if spread:
            # heavy top fronts shared by the ranks of their group: panel messages device to device over RCCL
            pkg.set_options(pair_update=0, big_front_cols=16)
            owner, phase, span = sh.spread_partition(sym, world, min_step_flops=0, min_share=0.01, min_cols=32, min_panels_per_rank=1)
            sp = sh.ShardPlan(plan, sym, owner, phase, comm, span)
        st, owner, phase = sh.factorize_sharded(plan, sym, g["in_Ax"], tol, ntol, comm, Ap=g["in_Ap"], Ai=g["in_Ai"], shard_plan=sp)
        G = sh.gather_numeric(plan, sym, comm, owner, shard_plan=sp)
        plan.close()
        if rank == 0:
            ref = pkg.qr_factorize(sym, g["in_Ap"], g["in_Ai"], g["in_Ax"], tol, ntol)
            ok = (G.rank, G.maxfrank, G.maxfm, G.rh_total) == (ref.rank, ref.maxfrank, ref.maxfm, ref.rh_total)
            for k in ("Hm", "Hr", "HStair", "HPinv", "Rdead", "Rblock_off", "Hii", "HTau"):
                ok = ok and np.array_equal(getattr(G, k), getattr(ref, k))
            ok = ok and np.array_equal(G.Stack[:G.rh_total], ref.Stack[:ref.rh_total])
            q.put((bool(ok), len(sh.cross_edges(sym, owner, phase)), 0 if sp is None else int((sp.span > 1).sum())))
        done = True
    finally:
        finish_ranks(dist, done)


@pytest.mark.parametrize("name,world,spread", [("epb1", 2, False), ("grid20_standin", 2, False), ("grid20_standin", 4, False),
                                               ("grid20_standin", 2, True), ("grid20_standin", 4, True), ("lns_3937", 4, True)])
def test_real_hip_plans_over_rccl(name, world, spread):
    # The skip decision comes from the library this pytest process has ALREADY loaded (stmmqr_device_count = hipGetDeviceCount on
    # the runtime that is live here), not from torch.  Round 3's form was `import torch; torch.cuda.device_count()`: this module
    # sorts first among the files that import torch, so that line was the session's FIRST `import torch` -- several GB of shared
    # objects paged in on a fresh box (1-2 minutes at the best of times) inside a process with a live HIP context -- followed by
    # torch's own device enumeration (amdsmi / a second look at the KFD topology).  The one full-suite run that went silent for 7
    # minutes stopped exactly there; faulthandler (pytest.ini) would have named either the dlopen under `import torch` or
    # `_device_count_amdsmi`.  The parent now touches neither; only the spawned ranks import torch, each in a fresh process, and
    # run_ranks ends them whatever happens.
    pkg = importlib.import_module(PKG)
    if pkg.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    from stmmqr_testlib import run_ranks
    ok, ncross, nshared = run_ranks(_worker, world, (name, spread), timeout=600)
    assert ok and ncross >= 1
    assert nshared >= 1 or not spread
