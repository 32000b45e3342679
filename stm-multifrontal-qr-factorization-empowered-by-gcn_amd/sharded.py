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
* `spread_partition`: the subtrees alone cannot scale a tree whose flops sit in its top fronts (c5: the root front is
  99.98 % of the work).  A heavy front of a group's top set is therefore SHARED by the group's ranks
  (include/stmmqr_hip.h, "A front SHARED between plans"): every rank assembles the whole front, panel q is factorized by
  rank q mod R of the group and sent point-to-point to the others, every rank updates the 32-column blocks of the panels
  it owns -- same arithmetic per column block, so the result is still bit-identical to one GPU (without the pair update).
  A shared front is alone in its phase; its packed contribution block is gathered on the group's first rank, its packed
  R+H block is merged by columns in `merge_shards`.

The per-rank compute object only has to offer the small "plan" interface of capi.HipQR
(set_groups / begin / run_group / finish / front_info / export_front / import_front / download, and for shared fronts
group_steps / run_step / panel_doubles / export_panel / import_panel / export_front_cols / import_front_cols / front_rhoff /
front_flops), so the CPU tests drive
the very same orchestration with a CPU stand-in plan.
"""
from __future__ import annotations

import os

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


def _partition_levels(sym: dict, nranks: int, oversub: int = 4):
    """-> (owner[nf], level[nf]).  Tree of joins: a front of phase k > 0 belongs to the top set of a group of ranks
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
    return owner, level


def partition(sym: dict, nranks: int, oversub: int = 4):
    """-> (owner[nf], phase[nf]): the tree of joins with dense phases (levels that ended up empty are skipped)"""
    owner, level = _partition_levels(sym, nranks, oversub)
    used = sorted(set(int(x) for x in level))
    remap = {lv: i for i, lv in enumerate(used)}
    if 0 not in remap:
        remap = {lv: i + 1 for i, lv in enumerate(used)}
    phase = np.array([remap[int(x)] for x in level], I64)
    return owner, phase


NB = 32                                   # panel width of the device path (STM_NB): column ownership inside a shared front


def front_flop_split(sym: dict):
    """(panel[nf], update[nf]): the flop bound of front_flop_bounds split into what the panel factorizations do (the chain
    that stays serial inside a shared front) and what the trailing updates do (what is spread over the ranks)."""
    nf = int(sym["nf"])
    Rp, Fm = (np.asarray(sym[k], I64) for k in ("Rp", "Fm"))
    pan, upd = np.zeros(nf), np.zeros(nf)
    for f in range(nf):
        fn, fm = int(Rp[f + 1] - Rp[f]), int(Fm[f])
        j = np.arange(min(fn, fm), dtype=np.float64)
        h = fm - j
        k2 = np.minimum((j // NB + 1) * NB, fn)
        pan[f] = float(np.sum(3 * h + 4 * h * (k2 - j - 1)))
        upd[f] = float(np.sum(4 * h * (fn - k2)))
    return pan, upd


def spread_partition(sym: dict, nranks: int, oversub: int = 4, min_share: float = 0.02, min_panels_per_rank: int = 4,
                     min_cols: int = 256, min_flops: float = 0.0, min_step_flops: float = 6e9):
    """-> (owner[nf], phase[nf], span[nf]).  partition() plus shared fronts: a front of the top set of a group of R = 2^k
    ranks whose flop bound is at least `min_share` of the whole tree's and `min_flops`, and at least `min_step_flops` per
    panel step -- round 4, measured with the native loop (stmmqr_factorize_shared_front, tools/shared_step_cost.py on one MI355X):
    a shared step costs its owner +38 us (5976-column front) to +60 us (7818 columns) over the unshared step (block 0 as a launch
    pair of its own, the panel message packed and unpacked), plus the message itself on a real link (2-13 MB: ~50-250 us over one
    xGMI link, hidden behind the update only on the ranks that do not wait for it); splitting the update R ways must save more
    than that: at the 10-15 TFLOP/s these updates run at, 6e9 flops are ~400-600 us of update per step.  The round-3 value, 2e10,
    was sized for the Python loop (hundreds of microseconds of host time per step).  The xenon1 / sme3Dc stand-ins' top fronts
    carry ~1e9 flops per step -- 40 us of update: sharing them cannot pay, whatever the loop costs --, with at
    least `min_cols` columns and `min_panels_per_rank` panels per rank, is shared by the ranks [owner, owner + R) (span = R, else 1).  A shared front is
    alone in its phase: inside a group's top set the fronts are numbered in postorder and every shared front closes the
    stage before it, so children still never run later than their parents."""
    owner, level = _partition_levels(sym, nranks, oversub)
    nf = int(sym["nf"])
    Rp, Fm = (np.asarray(sym[k], I64) for k in ("Rp", "Fm"))
    Post = np.asarray(sym["Post"], I64)[:nf]
    fl = front_flop_bounds(sym)
    total = float(fl.sum())
    span = np.ones(nf, I64)
    for f in range(nf):
        R = 1 << int(level[f])
        fn, fm = int(Rp[f + 1] - Rp[f]), int(Fm[f])
        npanels = (min(fn, fm) + NB - 1) // NB
        if R > 1 and fl[f] >= max(min_share * total, min_flops, min_step_flops * npanels) and fn >= min_cols and fm >= 64 and npanels >= min_panels_per_rank * R:
            span[f] = R
    stage = np.zeros(nf, I64)
    state = {}                            # (level, first rank of the group) -> [next stage, the current stage has fronts]
    for f in Post:
        if level[f] == 0:
            continue
        st = state.setdefault((int(level[f]), int(owner[f])), [0, False])
        if span[f] > 1:
            if st[1]:
                st[0] += 1
            stage[f] = st[0]
            st[0] += 1
            st[1] = False
        else:
            stage[f] = st[0]
            st[1] = True
    keys = sorted(set((int(level[f]), int(stage[f])) for f in range(nf)))
    if (0, 0) not in keys:
        keys = [(0, 0)] + keys
    remap = {k: i for i, k in enumerate(keys)}
    phase = np.array([remap[(int(level[f]), int(stage[f]))] for f in range(nf)], I64)
    return owner, phase, span


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


def critical_path_flops(sym: dict, owner, phase, nranks, span=None):
    """(critical, total): sum over the phases of the heaviest rank's flop bound vs the whole tree's -- the bound on the
    strong-scaling speed-up is total / critical.  A shared front (span R > 1) counts its panel chain in
    full and 1/R of its trailing updates on every rank of its group."""
    fl = front_flop_bounds(sym)
    owner, phase = np.asarray(owner), np.asarray(phase)
    if span is None:
        span = np.ones(len(fl), I64)
    span = np.asarray(span)
    pan, upd = front_flop_split(sym) if np.any(span > 1) else (fl, fl * 0)
    crit = 0.0
    for k in range(int(phase.max(initial=0)) + 1):
        load = np.zeros(nranks)
        for f in np.nonzero(phase == k)[0]:
            R = int(span[f])
            if R > 1:
                load[int(owner[f]):int(owner[f]) + R] += pan[f] + upd[f] / R
            else:
                load[int(owner[f])] += fl[f]
        crit += float(load.max(initial=0.0))
    return crit, float(fl.sum())


# --------------------------------------------------------------------------------------------------
# communication (torch.distributed point-to-point, batched per phase; a None comm = single process)
# --------------------------------------------------------------------------------------------------
class Comm:
    def __init__(self, dist=None, device=None, native=None):
        self.dist = dist
        self.device = device            # torch device of this rank (None: host tensors, gloo)
        self.native = native            # a stmmqr_transport (capi.RcclTransport): the panel loop of a shared front then runs as ONE
                                        #  native call (stmmqr_factorize_shared_front) instead of the step-by-step loop below
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


SHARED = 1 << 30                          # STMMQR_GROUP_SHARED
PREP, PANEL, UPDATE, GRAM, POST = 1, 2, 4, 8, 16     # STMMQR_STEP_*


class ShardPlan:
    """What one rank needs to know about a partition, computed ONCE per (plan, partition) -- not per factorization: this
    rank's groups (installed in the plan: a schedule rebuild and a few device allocations) and, per phase, the
    contribution blocks it sends or receives and the shared front it takes part in."""

    def __init__(self, plan, sym: dict, owner, phase, comm, span=None):
        nf = int(sym["nf"])
        self.owner, self.phase = np.asarray(owner), np.asarray(phase)
        self.span = np.ones(nf, I64) if span is None else np.asarray(span, I64)
        r = comm.rank
        lo, hi = self.owner, self.owner + self.span
        mine = (lo <= r) & (r < hi)                           # fronts factorized (or shared) here
        self.mine = mine
        self.group = np.where(mine, self.phase + np.where(self.span > 1, SHARED, 0), -1).astype(np.int32)
        # the cut schedule (a front gets the panels up to the one where it runs out of rows: include/stmmqr_hip.h,
        # stmmqr_plan_set_early_end): a front that outlives it is reported at finish and EVERY rank factorizes again on the full
        # schedule -- factorize_sharded agrees on that with one 8-byte exchange per factorization.  STMMQR_EARLY_END_SHARDED=0: off
        self.early = hasattr(plan, "set_early_end") and os.environ.get("STMMQR_EARLY_END_SHARDED", "1") != "0"
        plan.set_groups(self.group)
        if self.early:
            plan.set_early_end(1)
        self.nphase = int(self.phase.max(initial=0)) + 1
        parent, _, _ = tree_arrays(sym)
        self.parent = parent
        # a contribution block lives (complete) on owner[c]; it goes to every rank that assembles the parent
        self.out = [[] for _ in range(self.nphase)]           # [(c, dst)]
        self.inn = [[] for _ in range(self.nphase)]           # [(c, src)]
        for c in range(nf):
            p = int(parent[c])
            if p < 0:
                continue
            src = int(self.owner[c])
            for dst in range(int(lo[p]), int(hi[p])):
                if dst != src:
                    if src == r:
                        self.out[int(self.phase[p])].append((c, dst))
                    if dst == r:
                        self.inn[int(self.phase[p])].append((c, src))
        self.has = [bool(np.any(mine & (self.phase == k) & (self.span == 1))) for k in range(self.nphase)]
        self.shared_at = [None] * self.nphase                 # phase -> the shared front this rank takes part in
        self.ring = {}
        for f in np.nonzero(mine & (self.span > 1))[0]:
            self.shared_at[int(self.phase[f])] = int(f)
        self.shared = [int(f) for f in np.nonzero(mine & (self.span > 1))[0]]

    def c_phases(self):
        """the phase lists as the C structure of stmmqr_factorize_phases (built once; the arrays stay alive with it)"""
        if getattr(self, "_c_phases", None) is None:
            from .capi import ShardPhasesC
            import ctypes as C
            optr = np.cumsum([0] + [len(x) for x in self.out]).astype(I64)
            iptr = np.cumsum([0] + [len(x) for x in self.inn]).astype(I64)
            arrs = [optr, np.array([c for l in self.out for c, _ in l], I64), np.array([p for l in self.out for _, p in l], np.int32),
                    iptr, np.array([c for l in self.inn for c, _ in l], I64), np.array([p for l in self.inn for _, p in l], np.int32),
                    np.array([-1 if f is None else f for f in self.shared_at], I64),
                    np.array([0 if f is None else int(self.owner[f]) for f in self.shared_at], np.int32),
                    np.array([1 if f is None else int(self.span[f]) for f in self.shared_at], np.int32),
                    np.array([int(h) for h in self.has], np.int32)]
            ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_long if a.dtype == I64 else C.c_int))       # noqa: E731
            self._c_phases = (ShardPhasesC(self.nphase, *[ptr(a) for a in arrs]), arrs)
        return self._c_phases

    def panel_ring(self, plan, f, comm):
        """R message buffers of a shared front (slot q mod R: a rank receives R - 1 panels between two of its own exports,
        and an export waits for everything queued before it, so a slot is free again when its turn comes)"""
        if f not in self.ring:
            nd = plan.panel_doubles(f)
            self.ring[f] = [comm.empty(nd, np.float64) for _ in range(int(self.span[f]))]
        return self.ring[f]


def _use_dev(plan, comm, name):
    return comm.device is not None and hasattr(plan, name)


def _sync_recv(comm):
    if comm.device is not None:
        import torch
        torch.cuda.current_stream(comm.device).synchronize()


def run_shared_front(plan, sp: ShardPlan, f, comm: Comm):
    """The panel loop of one shared front on this rank (include/stmmqr_hip.h): i = my place in the group, panel q belongs to
    place q mod R.  Column block b of step q holds the columns of panel q + 1 + b, so my blocks of step q start at
    (i - q - 1) mod R with stride R.  The owner of panel q + 1 updates its block 0 first, factorizes and sends the panel,
    and only then updates the rest of step q -- the others meanwhile run their whole update of step q."""
    g, r0, R = int(sp.phase[f]), int(sp.owner[f]), int(sp.span[f])
    i = comm.rank - r0
    nsteps = plan.group_steps(g)
    native = getattr(comm, "native", None) is not None and hasattr(plan, "shared_front_native")
    if native:
        # the same loop in C++ (csrc/stmmqr_multi.cpp): panels, updates and messages enqueued without a host round trip per step
        plan.shared_front_native(g, f, r0, R, comm.native)
    ring = None if native else sp.panel_ring(plan, f, comm)
    dev = _use_dev(plan, comm, "export_panel_dev")
    if not native:
        plan.run_step(g, 0, PREP)
    for t in range(0 if native else nsteps):
        o = t % R
        first = (i - t) % R                                   # my first column block of step t - 1
        buf = ring[o]
        if i == o:
            if t > 0:
                plan.run_step(g, t - 1, UPDATE | GRAM, first, R, 1)
            plan.run_step(g, t, PANEL)
            if dev:
                plan.export_panel_dev(f, t, buf.data_ptr())
            else:
                buf.copy_(comm.tensor(plan.export_panel(f, t)))
            if t > 0:
                plan.run_step(g, t - 1, UPDATE, first + R, R, -1)
            comm.exchange([(buf, r0 + j) for j in range(R) if j != i], [])
        else:
            if t > 0:
                plan.run_step(g, t - 1, UPDATE | GRAM, first, R, -1)
            comm.exchange([], [(buf, r0 + o)])
            if dev:
                _sync_recv(comm)
                plan.import_panel_dev(f, t, buf.data_ptr())
            else:
                plan.import_panel(f, t, buf.cpu().numpy())
    if not native:
        plan.run_step(g, nsteps - 1, UPDATE | GRAM, (i - nsteps) % R, R, -1)
        plan.run_step(g, nsteps - 1, POST)
    # the packed contribution block is complete in my columns only: the group's first rank collects the others' columns
    if sp.parent[f] >= 0 and R > 1:
        devc = _use_dev(plan, comm, "export_front_cols_dev")
        if i == 0:
            sizes = [plan.front_cols_doubles(f, j, R) for j in range(R)]
            bufs = [comm.empty(max(sizes[j], 1), np.float64) for j in range(R)]
            comm.exchange([], [(bufs[j][:sizes[j]], r0 + j) for j in range(1, R) if sizes[j] > 0])
            for j in range(1, R):
                if sizes[j] > 0:
                    if devc:
                        _sync_recv(comm)
                        plan.import_front_cols_dev(f, j, R, bufs[j].data_ptr())
                    else:
                        plan.import_front_cols(f, j, R, bufs[j][:sizes[j]].cpu().numpy())
        else:
            n = plan.front_cols_doubles(f, i, R)
            if n > 0:
                if devc:
                    t = comm.empty(n, np.float64)
                    plan.export_front_cols_dev(f, i, R, t.data_ptr())
                else:
                    t = comm.tensor(plan.export_front_cols(f, i, R))
                comm.exchange([(t, r0)], [])


def factorize_sharded(plan, sym: dict, Ax, tol, ntol, comm: Comm, Ap=None, Ai=None, owner=None, phase=None,
                      device_ptr=None, shard_plan: ShardPlan | None = None, span=None):
    """Run one sharded factorization.  `plan` is this rank's compute object.  Returns (stats, owner, phase).
    shard_plan: the ShardPlan of (plan, owner, phase, span) when the caller factorizes repeatedly (bench.py): the groups
    are then installed once, outside any timed region.  stats["flops"] counts a shared front on the first rank of its
    group only, so the ranks' flops add up to the factorization's."""
    if shard_plan is None:
        if owner is None:
            owner, phase = partition(sym, comm.size)
        shard_plan = ShardPlan(plan, sym, owner, phase, comm, span)
    sp = shard_plan
    owner, phase = sp.owner, sp.phase
    if getattr(sp, "early", False):
        from .capi import StmmqrError, ERR_RESCHEDULE
        for attempt in (0, 1):
            bad, stats = 0, None
            try:
                stats = _factorize_sharded_once(plan, sp, Ax, tol, ntol, comm, Ap, Ai, device_ptr)
            except StmmqrError as e:
                if e.code != ERR_RESCHEDULE:
                    raise
                bad = 1
            if not _any_rank(comm, bad):
                stats["reschedules"] = attempt
                return stats, owner, phase
            # some rank's front outlived the cut schedule (rank-deficient fronts): every rank runs again on the full schedule, and
            # keeps it (the plan that failed has switched by itself and is told again: harmless)
            sp.early = False
            plan.set_early_end(0)
            Ap = Ai = None                                    # (the pattern is set)
    stats = _factorize_sharded_once(plan, sp, Ax, tol, ntol, comm, Ap, Ai, device_ptr)
    return stats, owner, phase


def _any_rank(comm, flag: int) -> int:
    """max of an integer flag over the ranks: one 8-byte message to and from every other rank (point-to-point, like everything here)"""
    if comm.size <= 1:
        return int(flag)
    mine = comm.tensor(np.array([int(flag)], I64))
    others = [comm.empty(1, np.int64) for _ in range(comm.size - 1)]
    peers = [r for r in range(comm.size) if r != comm.rank]
    comm.exchange([(mine, r) for r in peers], list(zip(others, peers)))
    return max([int(flag)] + [int(t.cpu().numpy()[0]) for t in others])


def _factorize_sharded_once(plan, sp, Ax, tol, ntol, comm, Ap, Ai, device_ptr):
    owner, phase = sp.owner, sp.phase
    plan.begin(Ax, tol, ntol, Ap, Ai, device_ptr=device_ptr)
    native = getattr(comm, "native", None)
    whole = native is not None and hasattr(plan, "phases_native") and os.environ.get("STMMQR_NATIVE_PHASES", "1") != "0"
    if whole:
        # every phase -- exchanges, shared fronts, this rank's groups -- enqueued by ONE native call (csrc/stmmqr_multi.cpp:
        # stmmqr_factorize_phases): no interpreter and no host wait between two phases
        plan.phases_native(sp.c_phases(), native)
    for k in range(0 if whole else sp.nphase):
        if k > 0 and comm.size > 1:
            # contribution blocks move up the tree only where subtrees join: every block that enters phase k from another rank
            mine_out, mine_in = sp.out[k], sp.inn[k]
            exported = {}
            for c, _ in mine_out:
                if c not in exported:
                    exported[c] = _export(plan, c, comm)
            metas_in = [comm.empty(4, np.int64) for _ in mine_in]
            comm.exchange([(comm.tensor(np.array([exported[c][0][q] for q in ("fm", "rank", "cm", "csize")], I64)), dst)
                           for c, dst in mine_out],
                          [(t, src) for t, (c, src) in zip(metas_in, mine_in)])
            metas = [[int(x) for x in t.cpu().numpy()] for t in metas_in]
            bufs = [(comm.empty(max(m[3], 1), np.float64), comm.empty(max(m[2], 1), np.int64)) for m in metas]
            sends, recvs = [], []
            for c, dst in mine_out:
                i, tC, trows = exported[c]
                if i["csize"] > 0:
                    sends += [(tC, dst), (trows, dst)]
            for (c, src), m, (bC, bR) in zip(mine_in, metas, bufs):
                if m[3] > 0:
                    recvs += [(bC[:m[3]], src), (bR[:m[2]], src)]
            comm.exchange(sends, recvs)
            for (c, src), m, (bC, bR) in zip(mine_in, metas, bufs):
                _import(plan, c, m[0], m[1], m[2], bC[:m[3]], bR[:m[2]], comm)
        if sp.shared_at[k] is not None:
            run_shared_front(plan, sp, sp.shared_at[k], comm)
        if sp.has[k]:
            plan.run_group(k)
    dup = [plan.front_flops(f) for f in sp.shared if comm.rank != int(owner[f])]
    stats = plan.finish()
    for fl, flu in dup:
        stats["flops"] = stats.get("flops", 0.0) - fl
        if "flops_update" in stats:
            stats["flops_update"] -= flu
    return stats


def shard_of(N, sym: dict, owned, plan=None, shard_plan: ShardPlan | None = None, rank=0):
    """The pieces of one rank's download that the merge needs: (arrays..., owned fronts, block sizes); for the shared
    fronts the column offsets of the packed R+H block and this rank's place in the group."""
    nf = int(sym["nf"])
    Post = np.asarray(sym["Post"], I64)[:nf]
    Rp = np.asarray(sym["Rp"], I64)
    owned = np.asarray(owned, bool)
    post_own = [int(f) for f in Post if owned[f]]
    size = {}
    for i, f in enumerate(post_own):
        end = N.Rblock_off[post_own[i + 1]] if i + 1 < len(post_own) else N.rh_total
        size[f] = int(end - N.Rblock_off[f])
    cols = {}
    if shard_plan is not None:
        for f in shard_plan.shared:
            cols[f] = (rank - int(shard_plan.owner[f]), int(shard_plan.span[f]), plan.front_rhoff(f, int(Rp[f + 1] - Rp[f])))
    return {"Stack": N.Stack[:N.rh_total], "Rblock_off": N.Rblock_off, "Rdead": N.Rdead, "HStair": N.HStair,
            "HTau": N.HTau, "Hii": N.Hii, "Hm": N.Hm, "Hr": N.Hr, "own": post_own, "size": size,
            "maxfrank": int(N.maxfrank), "cols": cols}


def merge_shards(sym: dict, shards, ntol=None):
    """Reference-format result from per-rank shards: packed R+H blocks in Post order (the single shrunk stack of
    the reference's serial run), H arrays merged, HPinv / Hii by qr_hpinv.  The block of a shared front is put together
    from the columns each rank of its group owns (panel q = columns [32 q, 32 q + 32) belongs to place q mod R)."""
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
            if f in sh.get("cols", {}):
                place, R, coff = sh["cols"][f]
                for q in range(place, (len(coff) - 1 + NB - 1) // NB, R):
                    c0, c1 = int(coff[q * NB]), int(coff[min((q + 1) * NB, len(coff) - 1)])
                    G.Stack[off[f] + c0:off[f] + c1] = sh["Stack"][a + c0:a + c1]
            else:
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


def gather_numeric(plan, sym: dict, comm: Comm, owner, ntol=None, shard_plan: ShardPlan | None = None):
    """Assemble the reference-format result on rank 0 (other ranks return None): every rank downloads the fronts it
    factorized and ships the arrays of its shard to rank 0 (point-to-point, one batched group; this is the API consumer's
    "give me the factors on the host" -- it is not part of a factorization step)."""
    N = plan.download()
    owned = shard_plan.mine if shard_plan is not None else (np.asarray(owner) == comm.rank)
    shard = shard_of(N, sym, owned, plan, shard_plan, comm.rank)
    if comm.size == 1:
        return merge_shards(sym, [shard], ntol)
    keys_f = ["Stack", "HTau"]
    keys_i = ["Rblock_off", "HStair", "Hii", "Hm", "Hr"]
    host = Comm(comm.dist, None) if comm.dist.get_backend() != "nccl" else comm

    def pack(sh):
        own = np.array(sh["own"], I64)
        sizes = np.array([sh["size"][f] for f in sh["own"]], I64)
        cf = sorted(sh["cols"])
        chead = np.array([x for f in cf for x in (f, sh["cols"][f][0], sh["cols"][f][1], len(sh["cols"][f][2]))], I64)
        coffs = np.concatenate([np.asarray(sh["cols"][f][2], I64) for f in cf]) if cf else np.zeros(0, I64)
        ints = [own, sizes, np.array([sh["maxfrank"]], I64), sh["Rdead"].astype(I64), chead, coffs] + \
               [np.asarray(sh[k], I64) for k in keys_i]
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
        sh = {"own": [int(x) for x in ai[0]], "maxfrank": int(ai[2][0]), "Rdead": ai[3].astype(np.int8), "cols": {}}
        sh["size"] = {f: int(z) for f, z in zip(sh["own"], ai[1])}
        pos = 0
        for q in range(0, ai[4].size, 4):
            f, place, R, ln = (int(x) for x in ai[4][q:q + 4])
            sh["cols"][f] = (place, R, ai[5][pos:pos + ln])
            pos += ln
        for k, a in zip(keys_i, ai[6:]):
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
