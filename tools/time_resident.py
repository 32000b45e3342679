"""Q'b and least-squares solve on the resident factors of one fixture, timed (STMMQR_QT4=0: per-panel split Q-apply; STMMQR_QBIG_MIN: split
threshold; STMMQR_MEMDUMP=1: size of T4).  usage: python tools/time_resident.py [fixture]"""
import sys, os, importlib, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from stmmqr_testlib import Symbolic, load_golden, scalar, csc_matvec
pkg=importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
name=sys.argv[1] if len(sys.argv)>1 else "xenon1_colamd_standin"
g=load_golden(name); S=Symbolic(g); sym={**S.sc, **{k:v for k,v in S.arr.items() if v is not None}}
tol,ntol=scalar(g,"in_tol"),int(scalar(g,"in_ntol"))
plan=pkg.HipQR(sym); plan.set_pattern(g["in_Ap"],g["in_Ai"])
st=plan.factorize(g["in_Ax"],tol,ntol)
xt=np.arange(S.n,dtype=np.float64); b=csc_matvec(S.m,g["in_Ap"],g["in_Ai"],g["in_Ax"],xt)
plan.qmult(0,b)
ts=[]
for _ in range(5):
    t0=time.perf_counter(); y=plan.qmult(0,b); ts.append((time.perf_counter()-t0)*1e3)
t0=time.perf_counter(); xs=plan.solve(b); t1=time.perf_counter()
res=float(np.linalg.norm(csc_matvec(S.m,g["in_Ap"],g["in_Ai"],g["in_Ax"],xs)-b)/(np.linalg.norm(g["in_Ax"])*np.linalg.norm(xs)+np.linalg.norm(b)))
print(name, "QT4", os.environ.get("STMMQR_QT4"), "qmult ms", ["%.2f"%t for t in ts], "solve %.2f"%((t1-t0)*1e3), "res %.2e"%res, "|Q'b| %.6e"%np.linalg.norm(y))
