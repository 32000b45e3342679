"""The subtree-sharding orchestration (stm-...-gcn_amd/sharded.py) on CPU: partition properties, and a world_size-2
gloo run whose compute object is the oracle-backed stand-in -- the assembled result must equal the serial oracle
bit for bit (same per-front arithmetic, only the placement and the contribution-block transport differ)."""
import importlib
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


SPREAD_SMALL = dict(min_flops=0, min_step_flops=0, min_share=0.01, min_cols=32, min_panels_per_rank=1)   # shared fronts on the small fixtures


@pytest.mark.parametrize("name", ["epb1", "syn_grid3d", "grid20_standin", "lns_3937", "c5mini_standin"])
@pytest.mark.parametrize("nranks", [2, 4, 8])
def test_spread_partition_properties(name, nranks):
    """shared fronts: alone in their phase, shared by an aligned group that contains every rank below them; children never
    later than parents; the model bound of the critical path does not get worse and passes 3x where the tree is one front"""
    from stmmqr_testlib import Symbolic, load_golden
    sh = shard_mod()
    S = Symbolic(load_golden(name))
    sym = sym_dict(S)
    small = dict(min_step_flops=0) if name == "c5mini_standin" else SPREAD_SMALL
    owner, phase, span = sh.spread_partition(sym, nranks, **small)
    parent, Child, Childp = sh.tree_arrays(sym)
    nf = S.nf
    assert np.all((span == 1) | (phase > 0))
    for f in range(nf):
        R = int(span[f])
        assert R >= 1 and (R & (R - 1)) == 0 and owner[f] % R == 0 and owner[f] + R <= nranks
        p = parent[f]
        if p >= 0:
            assert phase[f] <= phase[p]
            if span[f] > 1 or span[p] > 1 or owner[f] != owner[p]:
                assert phase[f] < phase[p]
            # everything below a front lives inside the ranks of its group
            assert owner[p] <= owner[f] and owner[f] + span[f] <= max(owner[p] + span[p], owner[f] + span[f])
        if R > 1:
            same = np.nonzero((phase == phase[f]) & (np.arange(nf) != f))[0]
            for o in same:                                    # another front of the same phase: another group of ranks
                assert owner[o] + span[o] <= owner[f] or owner[f] + R <= owner[o]
    o0, p0 = sh.partition(sym, nranks)
    c0, t0 = sh.critical_path_flops(sym, o0, p0, nranks)
    c1, t1 = sh.critical_path_flops(sym, owner, phase, nranks, span)
    assert t0 == t1 and c1 <= c0 * 1.0001
    if name == "c5mini_standin":
        assert int((span > 1).sum()) == 1 and t1 / c1 >= {2: 1.9, 4: 3.5, 8: 6.0}[nranks]


def _worker(rank, world, port, name, spread, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    from oracle_plan import OraclePlan
    from stmmqr_testlib import Oracle, Symbolic, finish_ranks, load_golden, scalar
    sh = shard_mod()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    done = False
    try:
        g = load_golden(name)
        S = Symbolic(g)
        sym = sym_dict(S)
        orc = Oracle()
        comm = sh.Comm(dist)
        plan = OraclePlan(S, orc)
        sp = None
        if spread:
            owner, phase, span = sh.spread_partition(sym, world, **SPREAD_SMALL)
            sp = sh.ShardPlan(plan, sym, owner, phase, comm, span)
        st, owner, phase = sh.factorize_sharded(plan, sym, g["in_Ax"], scalar(g, "in_tol"), int(scalar(g, "in_ntol")), comm,
                                                Ap=g["in_Ap"], Ai=g["in_Ai"], shard_plan=sp)
        G = sh.gather_numeric(plan, sym, comm, owner, shard_plan=sp)
        import torch
        fl = torch.tensor([st["flops"]], dtype=torch.float64)
        dist.all_reduce(fl)
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
            ok = ok and float(fl[0]) == No.c.flopcount            # a shared front is counted once
            ncross = len(sh.cross_edges(sym, owner, phase))
            q.put((bool(ok), int((phase > 0).sum()), ncross, 0 if sp is None else int((sp.span > 1).sum())))
        done = True
    finally:
        finish_ranks(dist, done)


def _run_world(name, world, spread):
    from stmmqr_testlib import run_ranks
    return run_ranks(_worker, world, (name, spread), timeout=300)


@pytest.mark.parametrize("name,world", [("syn_grid3d", 2), ("epb1", 2), ("syn_rankdef_grid", 2),
                                        ("epb1", 4), ("syn_grid3d", 4)])      # 4 ranks: three phases, ranks pair up 4 -> 2 -> 1
def test_gloo_ranks_match_serial_oracle(name, world):
    ok, ntop, ncross, _ = _run_world(name, world, False)
    assert ok
    assert ntop >= 1 and ncross >= 1          # the run really exchanged contribution blocks


@pytest.mark.parametrize("name,world", [("epb1", 2), ("grid20_standin", 2), ("syn_rankdef_grid", 2), ("lns_3937", 4), ("syn_grid3d", 4)])
def test_gloo_shared_fronts_match_serial_oracle(name, world):
    """the same with the heavy top fronts SHARED by the ranks of their group (sharded.run_shared_front): panel messages
    between the ranks, the packed contribution block gathered on the group's first rank, the packed R+H block merged by
    columns -- the stand-in plan poisons every column a rank does not own, so a wrong merge cannot pass"""
    ok, ntop, ncross, nshared = _run_world(name, world, True)
    assert ok
    assert nshared >= 1 and ncross >= 1


def _failing_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import torch
    import torch.distributed as dist
    from stmmqr_testlib import finish_ranks
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    done = False
    try:
        if rank == 1:
            raise RuntimeError("rank 1 fails before it sends anything")
        t = torch.zeros(4)
        dist.recv(t, src=1)                      # would wait for ever
        done = True
    finally:
        finish_ranks(dist, done)


def test_failed_rank_ends_the_run_instead_of_hanging_it():
    """test hygiene (round-3 verdict): one rank's exception must neither hang the others (no barrier in `finally`) nor leave
    them running (run_ranks terminates every child); the failure is reported within seconds, not at the timeout"""
    import time
    from stmmqr_testlib import run_ranks
    t0 = time.monotonic()
    with pytest.raises(AssertionError, match="rank exit codes"):
        run_ranks(_failing_worker, 2, (), timeout=120)
    assert time.monotonic() - t0 < 60
