"""The subtree-sharding orchestration (stm-...-gcn_amd/sharded.py) on CPU: partition properties, and a world_size-2
gloo run whose compute object is the oracle-backed stand-in -- the assembled result must equal the serial oracle
bit for bit (same per-front arithmetic, only the placement and the contribution-block transport differ)."""
import importlib
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"


def shard_mod():
    return importlib.import_module(PKG + ".sharded")


def sym_dict(S):
    return {**S.sc, **{k: v for k, v in S.arr.items() if v is not None}}


@pytest.mark.parametrize("name", ["epb1", "syn_grid3d", "grid20_standin", "lns_3937", "syn_star"])
@pytest.mark.parametrize("nranks", [1, 2, 4, 8])
def test_partition_properties(name, nranks):
    from stmmqr_testlib import Symbolic, load_golden
    sh = shard_mod()
    S = Symbolic(load_golden(name))
    sym = sym_dict(S)
    owner, phase = sh.partition(sym, nranks)
    parent, Child, Childp = sh.tree_arrays(sym)
    nf = S.nf
    assert owner.shape == (nf,) and set(np.unique(owner)) <= set(range(nranks))
    assert np.all(owner[phase == 1] == 0)
    for f in range(nf):
        if parent[f] >= 0:
            if phase[f] == 1:
                assert phase[parent[f]] == 1                  # the top set is closed upwards
            elif phase[parent[f]] == 0:
                assert owner[f] == owner[parent[f]]           # subtrees are never split
    if nranks == 1:
        assert not phase.any()
    elif nf > 50:
        fl = sh.front_flop_bounds(sym)
        load = np.array([fl[(owner == r) & (phase == 0)].sum() for r in range(nranks)])
        sub_piece = max((fl[(owner == r) & (phase == 0)].sum() for r in range(nranks)), default=0)
        # LPT bound: no rank carries more than the mean plus one piece; pieces are at most the heaviest subtree
        assert load.max() <= load.mean() + sub_piece + 1


def _worker(rank, world, port, name, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from oracle_plan import OraclePlan
    from stmmqr_testlib import Oracle, Symbolic, load_golden, scalar
    sh = shard_mod()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = load_golden(name)
        S = Symbolic(g)
        sym = sym_dict(S)
        orc = Oracle()
        comm = sh.Comm(dist)
        plan = OraclePlan(S, orc)
        st, owner, phase = sh.factorize_sharded(plan, sym, g["in_Ax"], scalar(g, "in_tol"), int(scalar(g, "in_ntol")), comm,
                                                Ap=g["in_Ap"], Ai=g["in_Ai"])
        G = sh.gather_numeric(plan, sym, comm, owner)
        if rank == 0:
            No = orc.factorize(S, g["in_Ap"], g["in_Ai"], g["in_Ax"], scalar(g, "in_tol"), int(scalar(g, "in_ntol")))
            nf = S.nf
            ok = (G.rank == No.c.rank and np.array_equal(G.Hm[:nf], No.Hm[:nf]) and np.array_equal(G.Hr[:nf], No.Hr[:nf])
                  and np.array_equal(G.HStair[:S.rjsize], No.HStair[:S.rjsize])
                  and np.array_equal(G.HPinv[:S.m], No.HPinv[:S.m]) and np.array_equal(G.Rdead[:S.n], No.Rdead[:S.n])
                  and np.array_equal(G.Rblock_off[:nf], No.Rblock_off[:nf]) and G.rh_total == No.c.rh_total
                  and np.array_equal(G.Stack[:G.rh_total], No.Stack[:No.c.rh_total])
                  and np.array_equal(G.HTau[:S.rjsize], No.HTau[:S.rjsize]) and G.maxfm == No.c.maxfm)
            for f in range(nf):
                a = S.Hip[f]
                ok = ok and np.array_equal(G.Hii[a:a + G.Hm[f]], No.Hii[a:a + No.Hm[f]])
            ncross = len([e for e in sh.cross_edges(sym, owner, phase) if owner[e[0]] != 0])
            q.put((bool(ok), int(phase.sum()), ncross))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["syn_grid3d", "epb1", "syn_rankdef_grid"])
def test_two_rank_gloo_matches_serial_oracle(name):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok, ntop, ncross = q.get(timeout=5)
    assert ok
    assert ntop >= 1 and ncross >= 1          # the run really exchanged contribution blocks
