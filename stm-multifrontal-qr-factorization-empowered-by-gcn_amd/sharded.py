"""Subtree sharding of ONE factorization over the GPUs of a node.

The reference splits the frontal tree into tasks and stacks for its NUMA thread pool
(STMMQR/src/qr/SparseQR_analyze.c:701-1034: `big_flops = total/grain`, tasks = subtrees below the big fronts,
TaskStack chains; dispatch in SparseQR_multithreads.c:14-115).  Here the same decomposition is made for devices:

* `partition(sym, nranks)`: the fronts are cut into a TOP set (the chain of large fronts near the root, all on
  rank 0) and independent subtrees below it, which are bin-packed onto the ranks by their flop bound (the
  reference's own flop formula on the symbolic staircase, SparseQR_analyze.c:557-571).
* phase 0: every rank factorizes its subtrees (no communication);
* exchange: for every tree edge that crosses into the top set the owner sends the child's packed contribution
  block, its row ids and (fm, rank, cm) to rank 0 -- point-to-point only (torch.distributed send/recv: RCCL over
  xGMI with backend "nccl", gloo in the CPU tests), one message group per cross edge, no collective;
* phase 1: rank 0 factorizes the top set.
The R+H blocks stay on the rank that produced them; `gather_numeric` assembles the reference-format result on rank 0.

The per-rank compute object only has to offer the small "plan" interface of capi.HipQR
(set_groups / begin / run_group / finish / export_front / import_front / download), so the CPU tests drive the
very same orchestration with a CPU stand-in plan.
"""
from __future__ import annotations

import numpy as np

I64 = np.int64


# --------------------------------------------------------------------------------------------------
# symbolic helpers
# --------------------------------------------------------------------------------------------------
def front_flop_bounds(sym: dict) -> np.ndarray:
    """Upper bound of the reference's flop count per front from the symbolic row bound Fm (same formula as
    SparseQR_analyze.c:557-571 with Stair[j] = min(Fm, ...) unknown -> full staircase)."""
    nf = int(sym["nf"])
    Rp, Super, Fm = (np.asarray(sym[k], I64) for k in ("Rp", "Super", "Fm"))
    fl = np.zeros(nf)
    for f in range(nf):
        fn, fm = int(Rp[f + 1] - Rp[f]), int(Fm[f])
        j = np.arange(min(fn, fm), dtype=np.float64)
        h = fm - j
        fl[f] = float(np.sum(3 * h + 4 * h * (fn - j - 1)))
    return fl


def tree_arrays(sym: dict):
    nf = int(sym["nf"])
    Child, Childp = np.asarray(sym["Child"], I64), np.asarray(sym["Childp"], I64)
    parent = np.full(nf, -1, I64)
    for f in range(nf):
        for q in range(Childp[f], Childp[f + 1]):
            parent[Child[q]] = f
    return parent, Child, Childp


def partition(sym: dict, nranks: int, oversub: int = 4):
    """-> (owner[nf], phase[nf]).  phase 1 = top set (rank 0), phase 0 = subtrees owned by `owner`."""
    nf = int(sym["nf"])
    parent, Child, Childp = tree_arrays(sym)
    Post = np.asarray(sym["Post"], I64)[:nf]
    fl = front_flop_bounds(sym)
    sub = fl.copy()
    for f in Post:                       # children before parents
        if parent[f] >= 0:
            sub[parent[f]] += sub[f]
    roots = [int(f) for f in range(nf) if parent[f] < 0]
    top = np.zeros(nf, bool)
    if nranks > 1:
        import heapq
        heap = [(-sub[r], r) for r in roots]
        heapq.heapify(heap)
        final = []                        # pieces that cannot be split (a single front)
        # split the heaviest subtree until there are enough pieces to balance
        while heap and len(heap) + len(final) < oversub * nranks:
            w, f = heapq.heappop(heap)
            kids = [int(Child[q]) for q in range(Childp[f], Childp[f + 1])]
            if not kids:
                final.append((w, f))
                continue
            top[f] = True
            for c in kids:
                heapq.heappush(heap, (-sub[c], c))
        pieces = sorted(((-w, f) for w, f in heap + final), reverse=True)
    else:
        pieces = [(sub[r], r) for r in roots]
    owner = np.zeros(nf, I64)
    load = np.zeros(nranks)
    # the top set runs on rank 0 after everything else: start rank 0 with that much load
    load[0] = float(fl[top].sum())
    root_owner = {}
    for w, f in pieces:                  # LPT
        r = int(np.argmin(load))
        root_owner[f] = r
        load[r] += w
    # propagate subtree ownership downwards (parents before children = reverse postorder)
    sub_root = np.full(nf, -1, I64)
    for f in Post[::-1]:
        f = int(f)
        if top[f]:
            continue
        if f in root_owner:
            sub_root[f] = f
        else:
            sub_root[f] = sub_root[parent[f]]
        owner[f] = root_owner[int(sub_root[f])]
    owner[top] = 0
    phase = top.astype(I64)
    return owner, phase


def cross_edges(sym: dict, owner, phase):
    """tree edges child -> parent whose child is a phase-0 front and whose parent is in the top set"""
    parent, _, _ = tree_arrays(sym)
    return [(int(c), int(parent[c])) for c in range(int(sym["nf"])) if parent[c] >= 0 and phase[c] == 0 and phase[parent[c]] == 1]


# --------------------------------------------------------------------------------------------------
# communication (torch.distributed point-to-point; a None comm = single process)
# --------------------------------------------------------------------------------------------------
class Comm:
    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank() if dist else 0
        self.size = dist.get_world_size() if dist else 1

    def _t(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device is not None else t

    def send(self, a, dst):
        self.dist.send(self._t(a), dst)

    def recv(self, shape, dtype, src):
        import torch
        t = torch.empty(shape, dtype={np.float64: torch.float64, np.int64: torch.int64}[dtype],
                        device=self.device if self.device is not None else "cpu")
        self.dist.recv(t, src)
        return t.cpu().numpy()


def factorize_sharded(plan, sym: dict, Ax, tol, ntol, comm: Comm, Ap=None, Ai=None, owner=None, phase=None,
                      device_ptr=None):
    """Run one sharded factorization.  `plan` is this rank's compute object.  Returns (stats, owner, phase)."""
    nf = int(sym["nf"])
    if owner is None:
        owner, phase = partition(sym, comm.size)
    r = comm.rank
    group = np.full(nf, -1, np.int32)
    group[(owner == r) & (phase == 0)] = 0
    if r == 0:
        group[phase == 1] = 1
    plan.set_groups(group)
    plan.begin(Ax, tol, ntol, Ap, Ai, device_ptr=device_ptr)
    plan.run_group(0)
    edges = cross_edges(sym, owner, phase)
    # contribution blocks move up the tree only where subtrees join the top set
    for c, _p in edges:
        src = int(owner[c])
        if src == 0:
            continue                                  # already on rank 0
        if r == src:
            info, Cb, rows = plan.export_front(c)
            comm.send(np.array([info["fm"], info["rank"], info["cm"], info["csize"]], I64), 0)
            if info["csize"] > 0:
                comm.send(Cb, 0)
                comm.send(rows, 0)
        elif r == 0:
            meta = comm.recv((4,), np.int64, src)
            fm, rk, cm, csize = (int(x) for x in meta)
            Cb = comm.recv((csize,), np.float64, src) if csize > 0 else np.zeros(0)
            rows = comm.recv((cm,), np.int64, src) if csize > 0 else np.zeros(0, I64)
            plan.import_front(c, fm, rk, cm, Cb, rows)
    if r == 0 and np.any(phase == 1):
        plan.run_group(1)
    stats = plan.finish()
    return stats, owner, phase


def shard_of(N, sym: dict, owned):
    """The pieces of one rank's download that the merge needs: (arrays..., owned fronts, block sizes)."""
    nf = int(sym["nf"])
    Post = np.asarray(sym["Post"], I64)[:nf]
    owned = np.asarray(owned, bool)
    post_own = [int(f) for f in Post if owned[f]]
    size = {}
    for i, f in enumerate(post_own):
        end = N.Rblock_off[post_own[i + 1]] if i + 1 < len(post_own) else N.rh_total
        size[f] = int(end - N.Rblock_off[f])
    return {"Stack": N.Stack[:N.rh_total], "Rblock_off": N.Rblock_off, "Rdead": N.Rdead, "HStair": N.HStair,
            "HTau": N.HTau, "Hii": N.Hii, "Hm": N.Hm, "Hr": N.Hr, "own": post_own, "size": size,
            "maxfrank": int(N.maxfrank)}


def merge_shards(sym: dict, shards):
    """Reference-format result from per-rank shards: packed R+H blocks in Post order (the single shrunk stack of
    the reference's serial run), H arrays merged, HPinv / Hii by qr_hpinv."""
    from .capi import QRNumeric
    nf, n, m = int(sym["nf"]), int(sym["n"]), int(sym["m"])
    Rp, Hip, Post, Super = (np.asarray(sym[k], I64) for k in ("Rp", "Hip", "Post", "Super"))
    rs = np.zeros(nf, I64)
    for sh in shards:
        for f in sh["own"]:
            rs[f] = sh["size"][f]
    off = np.zeros(nf, I64)
    run = 0
    for f in Post[:nf]:
        off[f] = run
        run += rs[f]
    G = QRNumeric(nf, n, m, int(sym["rjsize"]), int(sym["hisize"]), int(run))
    G.Rblock_off[:nf] = off
    mf = 1
    for sh in shards:
        mf = max(mf, sh["maxfrank"])
        for f in sh["own"]:
            a = sh["Rblock_off"][f]
            G.Stack[off[f]:off[f] + rs[f]] = sh["Stack"][a:a + rs[f]]
            G.HStair[Rp[f]:Rp[f + 1]] = sh["HStair"][Rp[f]:Rp[f + 1]]
            G.HTau[Rp[f]:Rp[f + 1]] = sh["HTau"][Rp[f]:Rp[f + 1]]
            hm = int(sh["Hm"][f])
            G.Hii[Hip[f]:Hip[f] + hm] = sh["Hii"][Hip[f]:Hip[f] + hm]
            G.Hm[f], G.Hr[f] = hm, sh["Hr"][f]
            G.Rdead[Super[f]:Super[f + 1]] = sh["Rdead"][Super[f]:Super[f + 1]]
    G.rank = int(G.Hr[:nf].sum())
    G.rank1 = G.rank
    G.maxfrank = mf
    hpinv(sym, G)
    return G


def gather_numeric(plan, sym: dict, comm: Comm, owner):
    """Assemble the reference-format result on rank 0 (other ranks return None): every rank downloads the fronts it
    factorized and ships them to rank 0 (point-to-point)."""
    import pickle
    N = plan.download()
    shard = shard_of(N, sym, owner == comm.rank)
    if comm.size == 1:
        return merge_shards(sym, [shard])
    if comm.rank != 0:
        blob = np.frombuffer(pickle.dumps(shard), np.uint8)
        blob = np.pad(blob, (0, (-blob.size) % 8))
        comm.send(np.array([blob.size], I64), 0)
        comm.send(blob.view(np.int64).copy(), 0)
        return None
    shards = [shard]
    for src in range(1, comm.size):
        nb = int(comm.recv((1,), np.int64, src)[0])
        words = comm.recv((nb // 8,), np.int64, src)
        shards.append(pickle.loads(words.view(np.uint8).tobytes()))
    return merge_shards(sym, shards)


def hpinv(sym: dict, G):
    """qr_hpinv (SparseQR_factorize.c:991-1060) on merged arrays: HPinv, Hii rewritten, maxfm."""
    nf, n, m = int(sym["nf"]), int(sym["n"]), int(sym["m"])
    Sleft, Hip, Rp, Super, PLinv = (np.asarray(sym[k], I64) for k in ("Sleft", "Hip", "Rp", "Super", "PLinv"))
    W = np.zeros(max(m, 1), I64)
    row1, row2 = 0, m
    for i in range(int(Sleft[n]), m):
        row2 -= 1
        W[i] = row2
    for f in range(nf):
        Hi = G.Hii[Hip[f]:Hip[f] + G.Hm[f]]
        rm, fm = int(G.Hr[f]), int(G.Hm[f])
        W[Hi[:rm]] = np.arange(row1, row1 + rm)
        row1 += rm
        cn = int((Rp[f + 1] - Rp[f]) - (Super[f + 1] - Super[f]))
        cm = min(fm - rm, cn)
        tail = Hi[rm + cm:fm][::-1]
        W[tail] = np.arange(row2 - 1, row2 - 1 - tail.size, -1)
        row2 -= tail.size
    G.maxfm = int(G.Hm[:nf].max(initial=0))
    G.HPinv[:m] = W[PLinv[:m]]
    for f in range(nf):
        sl = slice(Hip[f], Hip[f] + G.Hm[f])
        G.Hii[sl] = W[G.Hii[sl]]
