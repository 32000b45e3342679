"""Subtree sharding of ONE factorization over the GPUs of a node.

The reference splits the frontal tree into tasks and stacks for its NUMA thread pool
(STMMQR/src/qr/SparseQR_analyze.c:701-1034: `big_flops = total/grain`, tasks = subtrees below the big fronts,
TaskStack chains that follow a path to the root until they meet a task that already has a stack, :1001-1019;
dispatch in SparseQR_multithreads.c:14-115).  Here the same decomposition is made for devices, as a TREE OF JOINS:

* `partition(sym, nranks)`: the ranks are split recursively in halves.  A group of ranks receives a set of subtrees
  ("pieces"); the heaviest piece is opened (its root front joins the group's TOP set, its children become pieces) until
  the pieces can be balanced, then they are bin-packed (LPT on the reference's flop bound, SparseQR_analyze.c:557-571)
  onto the two halves, recursively.  A group of 2^k ranks factorizes its top set in PHASE k on its first rank, so with
  8 ranks: phase 0 = 8 independent sets of subtrees, phase 1 = 4 ranks join pairs, phase 2 = 2 ranks join quads,
  phase 3 = rank 0 finishes the root -- the device analogue of the reference's stack chains (the unit that moves is
  the contribution block of a front whose parent lives on another stack, SparseQR_factorize.c:1228).
* before phase k every tree edge that enters a phase-k front from another rank moves the child's packed contribution
  block, its row ids and (fm, rank, cm): point-to-point only (torch.distributed isend/irecv batched per phase: RCCL
  over xGMI with backend "nccl", gloo in the CPU tests), no collective.  With a HipQR plan and a device the block goes
  device to device: the C-arena slice is copied into a device tensor, sent, and copied into the receiver's C arena.
* the R+H blocks stay on the rank that produced them; `gather_numeric` assembles the reference-format result on rank 0.

The per-rank compute object only has to offer the small "plan" interface of capi.HipQR
(set_groups / begin / run_group / finish / front_info / export_front / import_front / download), so the CPU tests drive
the very same orchestration with a CPU stand-in plan.
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
    """-> (owner[nf], phase[nf]).  Tree of joins: a front of phase k > 0 belongs to the top set of a group of ranks
    [r0, r0 + 2^k') (k' >= k is the level at which the group was formed; empty levels are skipped so phases are dense)
    and is owned by r0; phase 0 = whole subtrees.  Children never have a later phase than their parent, and a child on
    another rank always has an earlier one."""
    nf = int(sym["nf"])
    parent, Child, Childp = tree_arrays(sym)
    Post = np.asarray(sym["Post"], I64)[:nf]
    fl = front_flop_bounds(sym)
    sub = fl.copy()
    for f in Post:                       # children before parents
        if parent[f] >= 0:
            sub[parent[f]] += sub[f]
    owner = np.zeros(nf, I64)
    level = np.zeros(nf, I64)             # 0 = inside a subtree, k = top set of a group of 2^k ranks
    roots = [int(f) for f in range(nf) if parent[f] < 0]

    def assign_subtree(f, r):
        stack = [f]
        while stack:
            x = stack.pop()
            owner[x] = r
            level[x] = 0
            stack.extend(int(Child[q]) for q in range(Childp[x], Childp[x + 1]))

    def split(pieces, r0, nr):
        """pieces: subtree roots for ranks [r0, r0 + nr)"""
        if nr == 1:
            for f in pieces:
                assign_subtree(f, r0)
            return
        import heapq
        k = int(np.log2(nr))
        heap = [(-sub[f], f) for f in pieces]
        heapq.heapify(heap)
        closed = []                       # single fronts without children: cannot be opened
        # open the heaviest piece until the two halves can be balanced: enough pieces, none heavier than a half's share
        while heap:
            total = -sum(w for w, _ in heap) + sum(w for w, _ in closed)
            w, f = heap[0]
            if len(heap) + len(closed) >= oversub * 2 and -w <= 0.5 * total / 1.0 * 0.5:
                break
            heapq.heappop(heap)
            kids = [int(Child[q]) for q in range(Childp[f], Childp[f + 1])]
            if not kids:
                closed.append((-w, f))
                continue
            owner[f] = r0
            level[f] = k                  # top set of this group
            for c in kids:
                heapq.heappush(heap, (-sub[c], c))
        allp = sorted([(-w, f) for w, f in heap] + closed, reverse=True)
        half = [[], []]
        load = [0.0, 0.0]
        for w, f in allp:                 # LPT onto the two halves
            h = 0 if load[0] <= load[1] else 1
            half[h].append(f)
            load[h] += w
        split(half[0], r0, nr // 2)
        split(half[1], r0 + nr // 2, nr // 2)

    if nranks & (nranks - 1):
        raise ValueError("the tree of joins needs a power-of-two number of ranks")
    split(roots, 0, nranks)
    # dense phases: levels that ended up empty are skipped
    used = sorted(set(int(x) for x in level))
    remap = {lv: i for i, lv in enumerate(used)}
    if 0 not in remap:
        remap = {lv: i + 1 for i, lv in enumerate(used)}
    phase = np.array([remap[int(x)] for x in level], I64)
    return owner, phase


def cross_edges(sym: dict, owner, phase, k=None):
    """tree edges child -> parent whose ends live on different ranks (the parent's phase is then later); k: only the
    edges entering phase k"""
    parent, _, _ = tree_arrays(sym)
    out = []
    for c in range(int(sym["nf"])):
        p = parent[c]
        if p >= 0 and owner[c] != owner[p] and (k is None or phase[p] == k):
            out.append((int(c), int(p)))
    return out


def critical_path_flops(sym: dict, owner, phase, nranks):
    """(critical, total): sum over the phases of the heaviest rank's flop bound vs the whole tree's -- the bound on the
    strong-scaling speed-up is total / critical."""
    fl = front_flop_bounds(sym)
    crit = 0.0
    for k in range(int(phase.max(initial=0)) + 1):
        crit += max((float(fl[(owner == r) & (phase == k)].sum()) for r in range(nranks)), default=0.0)
    return crit, float(fl.sum())


# --------------------------------------------------------------------------------------------------
# communication (torch.distributed point-to-point, batched per phase; a None comm = single process)
# --------------------------------------------------------------------------------------------------
class Comm:
    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.device = device            # torch device of this rank (None: host tensors, gloo)
        self.rank = dist.get_rank() if dist else 0
        self.size = dist.get_world_size() if dist else 1

    def tensor(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device is not None else t

    def empty(self, n, dtype):
        import torch
        return torch.empty(int(n), dtype={np.float64: torch.float64, np.int64: torch.int64}[dtype],
                           device=self.device if self.device is not None else "cpu")

    def exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor, src)] -- one batched group of point-to-point operations"""
        if not sends and not recvs:
            return
        d = self.dist
        ops = [d.P2POp(d.isend, t, dst) for t, dst in sends] + [d.P2POp(d.irecv, t, src) for t, src in recvs]
        for req in d.batch_isend_irecv(ops):
            req.wait()


def _export(plan, f, comm):
    """-> (info, C block, rows) with the block as a tensor on comm.device when the plan can hand it over there"""
    if comm.device is not None and hasattr(plan, "export_front_dev"):
        info = plan.front_info(f)
        t = comm.empty(max(info["csize"], 1), np.float64)
        rows = plan.export_front_dev(f, t.data_ptr(), info)
        return info, t[:info["csize"]], comm.tensor(rows)
    info, Cb, rows = plan.export_front(f)
    return info, comm.tensor(Cb), comm.tensor(rows)


def _import(plan, f, fm, rank, cm, tC, trows, comm):
    if comm.device is not None and hasattr(plan, "import_front_dev"):
        import torch
        torch.cuda.synchronize(comm.device)
        plan.import_front_dev(f, fm, rank, cm, tC.data_ptr(), trows.cpu().numpy())
    else:
        plan.import_front(f, fm, rank, cm, tC.cpu().numpy(), trows.cpu().numpy())


class ShardPlan:
    """What one rank needs to know about a partition, computed ONCE per (plan, partition) -- not per factorization: this
    rank's groups (installed in the plan: a schedule rebuild and a few device allocations) and, per phase, the cross-rank
    edges it sends or receives."""

    def __init__(self, plan, sym: dict, owner, phase, comm):
        self.owner, self.phase = np.asarray(owner), np.asarray(phase)
        r = comm.rank
        self.group = np.where(self.owner == r, self.phase, -1).astype(np.int32)
        plan.set_groups(self.group)
        self.nphase = int(self.phase.max(initial=0)) + 1
        parent, _, _ = tree_arrays(sym)
        cross = [(int(c), int(parent[c])) for c in range(int(sym["nf"])) if parent[c] >= 0 and self.owner[c] != self.owner[parent[c]]]
        self.out = [[(c, p) for c, p in cross if self.phase[p] == k and self.owner[c] == r] for k in range(self.nphase)]
        self.inn = [[(c, p) for c, p in cross if self.phase[p] == k and self.owner[p] == r] for k in range(self.nphase)]
        self.has = [bool(np.any(self.group == k)) for k in range(self.nphase)]


def factorize_sharded(plan, sym: dict, Ax, tol, ntol, comm: Comm, Ap=None, Ai=None, owner=None, phase=None,
                      device_ptr=None, shard_plan: ShardPlan | None = None):
    """Run one sharded factorization.  `plan` is this rank's compute object.  Returns (stats, owner, phase).
    shard_plan: the ShardPlan of (plan, owner, phase) when the caller factorizes repeatedly (bench.py): the groups are then
    installed once, outside any timed region."""
    if shard_plan is None:
        if owner is None:
            owner, phase = partition(sym, comm.size)
        shard_plan = ShardPlan(plan, sym, owner, phase, comm)
    owner, phase, group = shard_plan.owner, shard_plan.phase, shard_plan.group
    plan.begin(Ax, tol, ntol, Ap, Ai, device_ptr=device_ptr)
    nphase = shard_plan.nphase
    for k in range(nphase):
        if k > 0 and comm.size > 1:
            # contribution blocks move up the tree only where subtrees join: every edge entering phase k from another rank
            mine_out, mine_in = shard_plan.out[k], shard_plan.inn[k]
            out = [(c, p) + _export(plan, c, comm) for c, p in mine_out]
            metas_in = [comm.empty(4, np.int64) for _ in mine_in]
            comm.exchange([(comm.tensor(np.array([i["fm"], i["rank"], i["cm"], i["csize"]], I64)), int(owner[p]))
                           for c, p, i, _, _ in out],
                          [(t, int(owner[c])) for t, (c, p) in zip(metas_in, mine_in)])
            metas = [[int(x) for x in t.cpu().numpy()] for t in metas_in]
            bufs = [(comm.empty(max(m[3], 1), np.float64), comm.empty(max(m[2], 1), np.int64)) for m in metas]
            sends, recvs = [], []
            for c, p, i, tC, trows in out:
                if i["csize"] > 0:
                    sends += [(tC, int(owner[p])), (trows, int(owner[p]))]
            for (c, p), m, (bC, bR) in zip(mine_in, metas, bufs):
                if m[3] > 0:
                    recvs += [(bC[:m[3]], int(owner[c])), (bR[:m[2]], int(owner[c]))]
            comm.exchange(sends, recvs)
            for (c, p), m, (bC, bR) in zip(mine_in, metas, bufs):
                _import(plan, c, m[0], m[1], m[2], bC[:m[3]], bR[:m[2]], comm)
        if shard_plan.has[k]:
            plan.run_group(k)
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


def merge_shards(sym: dict, shards, ntol=None):
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
    # rank1 = live pivots among the first ntol columns (SparseQR_factorize.c:726-740)
    G.rank1 = G.rank if ntol is None or ntol >= n else int(np.count_nonzero(G.Rdead[:max(int(ntol), 0)] == 0))
    G.maxfrank = mf
    hpinv(sym, G)
    return G


def gather_numeric(plan, sym: dict, comm: Comm, owner, ntol=None):
    """Assemble the reference-format result on rank 0 (other ranks return None): every rank downloads the fronts it
    factorized and ships the arrays of its shard to rank 0 (point-to-point, one batched group; this is the API consumer's
    "give me the factors on the host" -- it is not part of a factorization step)."""
    N = plan.download()
    shard = shard_of(N, sym, owner == comm.rank)
    if comm.size == 1:
        return merge_shards(sym, [shard], ntol)
    keys_f = ["Stack", "HTau"]
    keys_i = ["Rblock_off", "HStair", "Hii", "Hm", "Hr"]
    host = Comm(comm.dist, None) if comm.dist.get_backend() != "nccl" else comm

    def pack(sh):
        own = np.array(sh["own"], I64)
        sizes = np.array([sh["size"][f] for f in sh["own"]], I64)
        ints = [own, sizes, np.array([sh["maxfrank"]], I64), sh["Rdead"].astype(I64)] + [np.asarray(sh[k], I64) for k in keys_i]
        flts = [np.asarray(sh[k], np.float64) for k in keys_f]
        return ints, flts

    ints, flts = pack(shard)
    if comm.rank != 0:
        head = np.array([a.size for a in ints] + [a.size for a in flts], I64)
        host.exchange([(host.tensor(head), 0)], [])
        host.exchange([(host.tensor(np.concatenate(ints)), 0), (host.tensor(np.concatenate(flts)), 0)], [])
        return None
    shards = [shard]
    nh = len(ints) + len(flts)
    for src in range(1, comm.size):
        head = host.empty(nh, np.int64)
        host.exchange([], [(head, src)])
        head = head.cpu().numpy()
        ti, tf = host.empty(int(head[:len(ints)].sum()), np.int64), host.empty(int(head[len(ints):].sum()), np.float64)
        host.exchange([], [(ti, src), (tf, src)])
        ai = np.split(ti.cpu().numpy(), np.cumsum(head[:len(ints)])[:-1])
        af = np.split(tf.cpu().numpy(), np.cumsum(head[len(ints):])[:-1])
        sh = {"own": [int(x) for x in ai[0]], "maxfrank": int(ai[2][0]), "Rdead": ai[3].astype(np.int8)}
        sh["size"] = {f: int(z) for f, z in zip(sh["own"], ai[1])}
        for k, a in zip(keys_i, ai[4:]):
            sh[k] = a
        for k, a in zip(keys_f, af):
            sh[k] = a
        shards.append(sh)
    return merge_shards(sym, shards, ntol)


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
