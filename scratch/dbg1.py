import sys, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from stmmqr_testlib import *
pkg=importlib.import_module("stm-multifrontal-qr-factorization-empowered-by-gcn_amd")
for name in ['syn_dupcol','lns_3937']:
    g=load_golden(name); S=Symbolic(g)
    sym={**S.sc, **{k:v for k,v in S.arr.items() if v is not None}}
    G=pkg.qr_factorize(sym,g['in_Ap'],g['in_Ai'],g['in_Ax'],scalar(g,'in_tol'),int(scalar(g,'in_ntol')))
    print(name,'tol',scalar(g,'in_tol'),'ntol',scalar(g,'in_ntol'),'rank',G.rank,scalar(g,'num_rank'))
    d=np.nonzero(G.Rdead[:S.n]!=g['num_Rdead'][:S.n])[0]
    print(' Rdead diff at',d[:20], 'gpu dead',np.nonzero(G.Rdead[:S.n])[0][:20],'ref dead',np.nonzero(g['num_Rdead'][:S.n])[0][:20])
    print(' Hm',G.Hm[:5],g['num_Hm'][:5],'Hr',G.Hr[:5],g['num_Hr'][:5])
    print(' stair gpu',G.HStair[:24]); print(' stair ref',g['num_HStair'][:24])
    print(' tau gpu',G.HTau[:10]); print(' tau ref',g['num_HTau'][:10])
