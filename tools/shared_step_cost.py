"""Per-step cost of the shared-front loops on ONE GPU: the top front of a fixture marked STMMQR_GROUP_SHARED with a group of one rank
(it owns every panel: no message is sent, everything else of the loop runs -- block 0 first, the panel, its export into the ring
buffer, the rest of the update) against the same factorization without sharing.  usage: shared_step_cost.py <fixture> [native|python]"""
import importlib, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from stmmqr_testlib import Symbolic, load_golden, scalar
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"
pkg = importlib.import_module(PKG); sh = importlib.import_module(PKG + ".sharded")
name = sys.argv[1] if len(sys.argv) > 1 else "c5mini_standin"
g = load_golden(name); S = Symbolic(g)
sym = {**S.sc, **{k: v for k, v in S.arr.items() if v is not None}}
tol, ntol = scalar(g, "in_tol"), int(scalar(g, "in_ntol"))
pkg.set_options(pair_update=0)
nf = S.nf
fn = np.diff(np.asarray(sym["Rp"]))[:nf]
root = int(np.argmax(fn))
parent, _, _ = sh.tree_arrays(sym)
owner = np.zeros(nf, np.int64); phase = np.zeros(nf, np.int64); span = np.ones(nf, np.int64)
# the root alone in phase 1 (and anything above it in phase 2: none, it is the root of its tree here)
phase[root] = 1
anc = parent[root]
while anc >= 0:
    phase[anc] = 2; anc = parent[anc]


class Comm1:
    rank, size, device, dist, native = 0, 1, None, None, None
    def exchange(self, s, r): assert not s and not r
    def tensor(self, a):
        import torch; return torch.from_numpy(np.ascontiguousarray(a))
    def empty(self, n, dt):
        import torch; return torch.empty(int(n), dtype=torch.float64)


def run(mode):
    comm = Comm1()
    plan = pkg.HipQR(sym)
    sp = sh.ShardPlan(plan, sym, owner, phase, comm, span)
    if mode != "plain":
        sp.span = span.copy(); sp.span[root] = 1
        grp = sp.group.copy(); grp[root] = 1 | sh.SHARED
        plan.set_groups(grp); sp.group = grp
        sp.shared_at[1] = root; sp.shared = [root]; sp.has[1] = False
        if mode == "native":
            comm.native = pkg.CallbackTransport(0, 1, lambda *a: 0, lambda *a: 0)
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        st, _, _ = sh.factorize_sharded(plan, sym, g["in_Ax"], tol, ntol, comm, Ap=g["in_Ap"] if it == 0 else None, Ai=g["in_Ai"] if it == 0 else None, shard_plan=sp)
        ts.append((time.perf_counter() - t0) * 1e3)
    steps = plan.group_steps(1)
    plan.close()
    return min(ts[1:]), steps


base, steps = run("plain")
for mode in sys.argv[2:] or ["native", "python"]:
    t, steps = run(mode)
    print(f"{name}: root front {fn[root]} columns, {steps} panel steps: unshared {base:.2f} ms, {mode} loop {t:.2f} ms -> +{(t - base) / steps * 1e3:.1f} us per step")
