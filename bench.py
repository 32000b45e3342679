#!/usr/bin/env python3
"""bench.py -- numerical-factorization GFLOP/s of the MI355X path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu]

A "step" is one numeric factorization (qr_factorize interval = the reference's "Factorize time",
STMMQR/src/qr/SparseQR.c:346-355) of one matrix whose symbolic analysis is a committed fixture and whose values
are already resident in HBM when the timed region starts; the factors stay in HBM (the D2H of the packed R+H is
reported separately in DESIGN.md, never in `value`).  Flops are the reference's own count
(SparseQR_factorize.c:1571), verified equal to the reference's on the same matrix.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank factorizes its own matrix
(independent objects, no data-path collective): "weak" scaling, value = all ranks' flops / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
PKG = "stm-multifrontal-qr-factorization-empowered-by-gcn_amd"

PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X dense fp64 matrix peak (= fp64 vector peak on gfx950)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy)


def cpu_baseline(name, g, budget_s=20.0):
    """The REAL reference (oracle/_ref/refdump, built by oracle/Makefile) timed on this host, 1 core (SPQR_grain = 1,
    STMMQR/README.md:71-72), MKL sequential; falls back to the CPU restatement (kind "port") when the reference
    build is not present.  Bounded sample: repetitions sized to ~budget_s seconds of CPU work."""
    refdump = ROOT / "oracle" / "_ref" / "refdump"
    flops = float(g["flopcount"][0])
    ref_s = float(g["fac_seconds"][0])                 # seconds in the build container: sizing hint only
    if refdump.exists():
        try:
            with tempfile.TemporaryDirectory() as td:
                mtx = Path(td) / "a.mtx"
                if "A_p" in g:
                    Ap, Ai, Ax = g["A_p"], g["A_i"], g["A_x"]
                    m, n = int(g["A_m"][0]), int(g["A_n"][0])
                else:
                    Ap, Ai, Ax = g["in_Ap"], g["in_Ai"], g["in_Ax"]
                    m, n = int(g["in_m"][0]), int(g["in_n"][0])
                cols = np.repeat(np.arange(n), np.diff(Ap))
                with open(mtx, "w") as f:
                    f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (m, n, len(Ax)))
                    np.savetxt(f, np.c_[Ai + 1, cols + 1, Ax], fmt="%d %d %.17g")
                reps = int(max(1, min(50, budget_s / max(ref_s * 1.5, 1e-3))))
                env = dict(os.environ, MKL_THREADING_LAYER="SEQUENTIAL", MKL_NUM_THREADS="1")
                ordering = str(int(g["ordering"][0])) if "ordering" in g else "-1"
                omap = {"5": "0", "2": "1", "11": "2", "6": "3"}      # QR_ORDERING_* -> refdump selector
                out = subprocess.run([str(refdump), str(mtx), omap.get(ordering, "-1"), "1", "d", "-", str(reps)],
                                     capture_output=True, text=True, env=env, timeout=600).stdout
                for line in out.splitlines():
                    if line.startswith("REF factorize seconds"):
                        sec = float(line.split(":")[1].split()[0])
                        rfl = [l for l in out.splitlines() if l.startswith("nf =")][0]
                        rflops = float(rfl.split("flops =")[1].split()[0])
                        cb = {"value": rflops / sec * 1e-9, "unit": "GFLOP/s", "cores": 1, "kind": "reference",
                              "sample": f"{name}: best of {reps} qr_factorize runs of the compiled reference "
                                        f"(SPQR_grain=1, MKL sequential), {sec * 1e3:.2f} ms each",
                              "seconds": sec}
                        for l2 in out.splitlines():
                            if l2.startswith("REF qmult(QTX) seconds"):
                                t = l2.replace(":", " ").split()
                                cb["qmult_qtx_seconds"] = float(t[3])
                                cb["solve_seconds"] = float(t[6])
                        return cb
        except Exception as e:  # fall through to the port
            print(f"[bench] reference baseline failed: {e}", file=sys.stderr)
    from stmmqr_testlib import Oracle, Symbolic, scalar
    orc = Oracle()
    S = Symbolic(g)
    t0 = time.perf_counter()
    N = orc.factorize(S, g["in_Ap"], g["in_Ai"], g["in_Ax"], scalar(g, "in_tol"), int(scalar(g, "in_ntol")))
    sec = time.perf_counter() - t0
    return {"value": N.c.flopcount / sec * 1e-9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"{name}: one run of oracle/stmmqr_oracle.c (scalar C restatement), {sec * 1e3:.1f} ms",
            "seconds": sec}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="xenon1_standin")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--big-front-cols", type=int, default=None)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N>1: replicas = every rank factorizes its own matrix (weak scaling, default); "
                         "sharded = ONE matrix, subtrees on the ranks, contribution blocks to rank 0 over RCCL (strong)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from stmmqr_testlib import Symbolic, load_golden, scalar
    pkg = importlib.import_module(PKG)
    if args.big_front_cols:
        pkg.set_options(big_front_cols=args.big_front_cols)
    name = args.workload
    g = load_golden(name)
    S = Symbolic(g)
    sym = {**S.sc, **{k: v for k, v in S.arr.items() if v is not None}}
    tol, ntol = scalar(g, "in_tol"), int(scalar(g, "in_ntol"))

    plan = pkg.HipQR(sym, device=local)
    plan.set_pattern(g["in_Ap"], g["in_Ai"])
    Ax = torch.from_numpy(np.ascontiguousarray(g["in_Ax"])).to(dev)      # values resident in HBM
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    st = None
    sharded = args.mode == "sharded" and world > 1
    if sharded:
        sh = importlib.import_module(PKG + ".sharded")
        comm = sh.Comm(dist, dev)
        owner, phase = sh.partition(sym, world)

        def step():
            return sh.factorize_sharded(plan, sym, None, tol, ntol, comm, owner=owner, phase=phase,
                                        device_ptr=Ax.data_ptr())[0]
    else:
        def step():
            return plan.factorize(None, tol, ntol, device_ptr=Ax.data_ptr())
    for _ in range(args.warmup):
        st = step()
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        st = step()
        dev_ms += st["ms_total"]
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall, st["flops"]], device=dev, dtype=torch.float64)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        wall = float(tmax[0].item())
        total_flops = float(tsum[1].item())          # replicas: world x flops; sharded: the shards add up to one matrix
    else:
        total_flops = st["flops"]
    flops = scalar(g, "flopcount")
    assert total_flops == (flops if (sharded or world == 1) else world * flops), (total_flops, flops)

    # one extra, un-timed step with per-category HIP events (forces a sync per level: not part of `value`)
    if sharded:
        plan.set_groups(np.zeros(S.nf, np.int32))
    det = plan.factorize(None, tol, ntol, device_ptr=Ax.data_ptr(), detail=True)
    if rank == 0:
        value = total_flops * args.steps / wall * 1e-9
        # Roofline of the dominant kernel.  Times are HIP-event sums on the plan's stream from the detail pass (one event
        # pair per launch category and level).  Algorithmic work: the reference's flop count (FLOP_COUNT, :1571) splits
        # into the dlarfb flops handed to the trailing update (4 * rows * cols * reflectors per panel) and the rest,
        # which the panel kernels do (dlarfg + in-panel dlarf + T).
        upd_tf = det["flops_update"] / max(det["ms_update"], 1e-9) * 1e-9 if det["ms_update"] > 0 else 0.0
        panel_flops = max(flops - det["flops_update"], 0.0)
        panel_tf = panel_flops / max(det["ms_front"], 1e-9) * 1e-9
        # trailing update: 2 reads + 1 write of the trailing block per panel = 24 B per (row, col); flops_update counts
        # 4 flops per (live column, row below its diagonal, trailing col), i.e. <= 4*32 per (row, col): a lower bound
        upd_bytes = det["flops_update"] * 24.0 / (4.0 * 32.0)
        upd_gbs = upd_bytes / max(det["ms_update"], 1e-9) * 1e-6
        pmc = {}
        for cand in sorted((ROOT / "profiles").glob("r*_pmc_fetch_write_per_kernel.json")):
            pmc_file = cand
            try:
                pmc = json.loads(cand.read_text())
            except Exception:
                pmc = {}

        def pmc_traffic(kernels):
            """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (KB units)."""
            tot, calls = 0.0, 0
            for k in kernels:
                e = pmc.get(k, {})
                if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                    tot += (e["FETCH_SIZE"]["sum_kb"] + e["WRITE_SIZE"]["sum_kb"]) * 1024.0
                    calls += e["FETCH_SIZE"]["calls"]
            return (tot / calls) if calls and "standin" in name else None

        if det["ms_front"] >= det["ms_update"]:
            roof = {"bound": "mfma", "kernel": "k_panel (+ k_front_wg): Householder panels, fp64 vector/MFMA rate",
                    "achieved": panel_tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": panel_tf / PEAK_FP64_MFMA_TFLOPS, "traffic": pmc_traffic(["k_panel"]),
                    "note": "latency-bound: one workgroup reduction per Householder column (DESIGN.md 4)"}
        else:
            roof = {"bound": "hbm", "kernel": "k_upd_w + k_upd_c (dlarfb on v_mfma_f64_16x16x4_f64)", "achieved": upd_gbs,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": upd_gbs / PEAK_HBM_GBS,
                    "traffic": pmc_traffic(["k_upd_w", "k_upd_c"])}
        roof["whole_factorization"] = {"bound": "mfma", "achieved": flops / max(det["ms_total"], 1e-9) * 1e-9,
                                       "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": flops / max(det["ms_total"], 1e-9) * 1e-9 / PEAK_FP64_MFMA_TFLOPS}
        roof["update_kernels"] = {"bound": "hbm", "kernel": "k_upd_w + k_upd_c", "achieved": upd_gbs, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": upd_gbs / PEAK_HBM_GBS, "tflops": upd_tf,
                                  "traffic": pmc_traffic(["k_upd_w", "k_upd_c"]),
                                  "traffic_source": str(pmc_file.relative_to(ROOT)) if pmc else None}
        roof["assembly"] = {"bound": "hbm", "achieved": det["bytes_assemble"] / max(det["ms_assemble"], 1e-9) * 1e-6,
                            "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": det["bytes_assemble"] / max(det["ms_assemble"], 1e-9) * 1e-6 / PEAK_HBM_GBS}
        roof["ms"] = {k: det[k] for k in ("ms_total", "ms_assemble", "ms_front", "ms_update", "ms_pack")}
        out = {
            "metric": "numerical-factorization GFLOP/s", "value": value, "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{name}: m={S.m} n={S.n} nnz={S.anz} fronts={S.nf} "
                                   f"flops/step={flops:.4g} (stand-in for xenon1.mtx, absent from the reference checkout)"
                       if "standin" in name else f"{name}: m={S.m} n={S.n} nnz={S.anz} fronts={S.nf} flops/step={flops:.4g}",
                       "inputs": "values resident in HBM, factors left in HBM", "parallelism": (f"subtree-sharded x{world}" if sharded else f"replica x{world}"),
                       "device_ms_per_step": dev_ms / args.steps, "launches_per_step": st["nlaunch"],
                       "levels": st["nlevels"]},
            "roofline": roof,
        }
        # SURVEY 8 (f1), outside the timed region: Q'b and the least-squares solve on the factors still resident in HBM
        # (wall time including the copies of the vectors), with the residual the reference's driver prints
        try:
            if not sharded:
                from stmmqr_testlib import csc_matvec
                Ap_, Ai_, Ax_ = g["in_Ap"], g["in_Ai"], g["in_Ax"]
                xt = np.arange(S.n, dtype=np.float64)
                b = csc_matvec(S.m, Ap_, Ai_, Ax_, xt)
                plan.qmult(0, b)
                t0 = time.perf_counter(); plan.qmult(0, b); t1 = time.perf_counter()
                xs = plan.solve(b); t2 = time.perf_counter()
                res = float(np.linalg.norm(csc_matvec(S.m, Ap_, Ai_, Ax_, xs) - b) /
                            (np.linalg.norm(Ax_) * np.linalg.norm(xs) + np.linalg.norm(b)))
                out["f1_resident_factors"] = {"qmult_qtx_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3, "residual": res}
        except Exception as e:  # rank-deficient inputs: the device solve refuses them
            out["f1_resident_factors"] = {"error": str(e)}
        if not args.no_cpu:
            cb = cpu_baseline(name, g)
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    plan.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
