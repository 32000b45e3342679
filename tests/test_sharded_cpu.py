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
    nphase = int(phase.max()) + 1
    assert nphase <= int(np.log2(nranks)) + 1
    for f in range(nf):
        p = parent[f]
        if p >= 0:
            assert phase[f] <= phase[p]                       # a child never runs in a later phase than its parent
            if owner[f] != owner[p]:
                assert phase[f] < phase[p]                    # a contribution block that moves is ready one phase earlier
            if phase[p] == 0:
                assert owner[f] == owner[p] and phase[f] == 0  # subtrees are never split
    # tree of joins: the top set of phase k lives on the first rank of an aligned group of 2^k' >= 2^k ranks, and every
    # front below it lives inside that group
    for f in range(nf):
        if phase[f] > 0:
            kids = [int(Child[q]) for q in range(Childp[f], Childp[f + 1])]
            span = max([abs(int(owner[c]) - int(owner[f])) for c in kids], default=0)
            assert all(owner[c] >= owner[f] for c in kids) and span < nranks
    if nranks == 1:
        assert not phase.any()
    elif nf > 50:
        fl = sh.front_flop_bounds(sym)
        load = np.array([fl[(owner == r) & (phase == 0)].sum() for r in range(nranks)])
        assert load.max() <= 0.75 * load.sum() + 1            # the subtrees are spread over the ranks
        crit, tot = sh.critical_path_flops(sym, owner, phase, nranks)
        assert 0 < crit <= tot


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
            ncross = len(sh.cross_edges(sym, owner, phase))
            q.put((bool(ok), int((phase > 0).sum()), ncross))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("syn_grid3d", 2), ("epb1", 2), ("syn_rankdef_grid", 2),
                                        ("epb1", 4), ("syn_grid3d", 4)])      # 4 ranks: three phases, ranks pair up 4 -> 2 -> 1
def test_gloo_ranks_match_serial_oracle(name, world):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok, ntop, ncross = q.get(timeout=5)
    assert ok
    assert ntop >= 1 and ncross >= 1          # the run really exchanged contribution blocks
