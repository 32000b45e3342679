import sys, os, importlib, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stmmqr_testlib import Symbolic, load_golden, scalar
pkg=importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
name=sys.argv[1] if len(sys.argv)>1 else "xenon1_colamd_standin"
g=load_golden(name); S=Symbolic(g); sym={**S.sc, **{k:v for k,v in S.arr.items() if v is not None}}
tol,ntol=scalar(g,"in_tol"),int(scalar(g,"in_ntol"))
if os.environ.get("PAIR"): pkg.set_options(pair_update=int(os.environ["PAIR"]))
if os.environ.get("LA"): pkg.set_options(lookahead=int(os.environ["LA"]))
if os.environ.get("BFC"): pkg.set_options(big_front_cols=int(os.environ["BFC"]))
plan=pkg.HipQR(sym); plan.set_pattern(g["in_Ap"],g["in_Ai"])
for _ in range(3): st=plan.factorize(g["in_Ax"],tol,ntol)
ts=[]
for _ in range(5):
    t0=time.perf_counter(); st=plan.factorize(g["in_Ax"],tol,ntol); ts.append((time.perf_counter()-t0)*1e3)
print(name, "LA_MIN", os.environ.get("STMMQR_LA_MIN"), "lookahead", pkg.get_options()["lookahead"], "wall ms min %.2f med %.2f dev %.2f"%(min(ts), sorted(ts)[2], st["ms_total"]), "flops ok", st["flops"]==scalar(g,"flopcount"), "retries", st["retries"])
