"""Longer randomized end-to-end run (tests/fuzz_sparseqr.py: rank, least-squares residual and solution against dense LAPACK) with the quad
update forced onto every large front (STMMQR_PAIR_MIN=1, big_front_cols 16 / 32): 24 seeds x 12 matrices.  usage: python tools/fuzz_quad.py"""
import sys, os, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ["STMMQR_PAIR_MIN"]="1"
pkg=importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
import fuzz_sparseqr
bad=0
for bfc in (16, 32):
    pkg.set_options(pair_update=4, big_front_cols=bfc)
    for seed in range(100, 112):
        r=fuzz_sparseqr.main(seed=seed, iters=3, big=(seed%4==0))
        print("bfc",bfc,"seed",seed,"->",r, flush=True)
        bad+= (r!=0)
print("BAD", bad)
