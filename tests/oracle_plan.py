"""Oracle-backed stand-in for capi.HipQR (TEST INFRASTRUCTURE): the same small "plan" interface, computed front by
front with the CPU restatement, so that the multi-rank orchestration (sharded.py) can be exercised with gloo on a
machine without GPUs."""
import ctypes as C

import numpy as np

from stmmqr_testlib import I64, Oracle, Symbolic, _dp, _ip, c_double_p, c_long_p


class Result:
    pass


class OraclePlan:
    def __init__(self, S: Symbolic, orc: Oracle):
        self.S, self.orc, self.L = S, orc, orc.lib
        self.nf = S.nf
        self.group = np.zeros(S.nf, np.int32)
        L = self.L
        L.orc_fsize.restype = C.c_long
        L.orc_fsize.argtypes = [C.c_long] + [c_long_p] * 9
        L.orc_assemble.restype = None
        L.orc_assemble.argtypes = [C.c_long, C.c_long] + [c_long_p] * 8 + [c_double_p, c_long_p, c_long_p,
                                                                          C.POINTER(c_double_p), c_long_p, c_long_p,
                                                                          c_long_p, c_long_p, c_double_p, c_long_p]

    def set_groups(self, group):
        self.group = np.ascontiguousarray(group, np.int32)

    def begin(self, Ax, tol, ntol, Ap=None, Ai=None, device_ptr=None):
        S = self.S
        if Ap is not None:
            self.Ap = np.ascontiguousarray(Ap, I64); self.Ai = np.ascontiguousarray(Ai, I64)
        Ax = np.ascontiguousarray(Ax, np.float64)
        self.tol = tol if S.do_rank_detection else -1.0
        self.ntol = ntol
        self.Sx = np.zeros(max(S.anz, 1)); W = np.zeros(max(S.m, S.nf) + 1, I64)
        self.L.orc_stranspose2(S.m, S.n, _ip(self.Ap), _ip(self.Ai), _dp(Ax), _ip(S.arr.get("Qfill")), _ip(S.Sp),
                               _ip(S.PLinv), _dp(self.Sx), _ip(W))
        self.Cm = np.zeros(max(S.nf, 1), I64); self.Hr = np.zeros(max(S.nf, 1), I64); self.Hm = np.zeros(max(S.nf, 1), I64)
        self.HStair = np.zeros(max(S.rjsize, 1), I64); self.HTau = np.zeros(max(S.rjsize, 1))
        self.Hii = np.zeros(max(S.hisize, 1), I64); self.Rdead = np.zeros(max(S.n, 1), np.int8)
        self.Cblk = {}; self.RH = {}; self.flops = 0.0; self.maxfrank = 1
        self.Fmap = np.zeros(max(S.n, 1), I64); self.Cmap = np.zeros(max(S.maxfn, 1), I64)

    def run_group(self, g, detail=False):
        S, L = self.S, self.L
        ch = self.orc.chunk()
        W = np.zeros(ch.fchunk * max(S.maxfn, 1) + 64)
        for f in S.Post[:S.nf]:
            f = int(f)
            if self.group[f] != g:
                continue
            Stair = self.HStair[S.Rp[f]:]
            fm = L.orc_fsize(f, _ip(S.Super), _ip(S.Rp), _ip(S.Rj), _ip(S.Sleft), _ip(S.Child), _ip(S.Childp),
                             _ip(self.Cm), _ip(self.Fmap), _ip(Stair))
            fn = int(S.Rp[f + 1] - S.Rp[f]); fp = int(S.Super[f + 1] - S.Super[f]); col1 = int(S.Super[f])
            self.Hm[f] = fm
            F = np.zeros(max(fm * fn, 1))
            ptrs = (c_double_p * (S.nf + 1))()
            for q in range(S.Childp[f], S.Childp[f + 1]):
                c = int(S.Child[q])
                ptrs[c] = _dp(self.Cblk[c])
            L.orc_assemble(f, fm, _ip(S.Super), _ip(S.Rp), _ip(S.Rj), _ip(S.Sp), _ip(S.Sj), _ip(S.Sleft), _ip(S.Child),
                           _ip(S.Childp), _dp(self.Sx), _ip(self.Fmap), _ip(self.Cm), ptrs, _ip(self.Hr), _ip(Stair),
                           _ip(self.Hii), _ip(S.Hip), _dp(F), _ip(self.Cmap))
            fl = C.c_double(0)
            Tau = self.HTau[S.Rp[f]:]
            frank = L.orc_front(fm, fn, fp, float(self.tol), int(self.ntol - col1), C.byref(ch), _dp(F), _ip(Stair),
                                C.cast(self.Rdead[col1:].ctypes.data, C.c_char_p), _dp(Tau), _dp(W), C.byref(fl))
            self.flops += fl.value
            self.maxfrank = max(self.maxfrank, int(frank))
            csize = L.orc_fcsize(fm, fn, fp, frank)
            Cb = np.zeros(max(csize, 1))
            self.Cm[f] = L.orc_cpack(fm, fn, fp, frank, _dp(F), _dp(Cb))
            self.Cblk[f] = Cb
            R = np.zeros(max(fm * fn, 1)); rm = C.c_long(0)
            rs = L.orc_rhpack(fm, fn, fp, _ip(Stair), _dp(F), _dp(R), C.byref(rm))
            self.Hr[f] = rm.value
            self.RH[f] = R[:rs].copy()

    def finish(self):
        return {"flops": self.flops}

    def front_info(self, f):
        S = self.S
        fn = int(S.Rp[f + 1] - S.Rp[f]); fp = int(S.Super[f + 1] - S.Super[f]); cm = int(self.Cm[f]); cn = fn - fp
        return {"fm": int(self.Hm[f]), "rank": int(self.Hr[f]), "cm": cm, "csize": cm * (cm + 1) // 2 + cm * (cn - cm),
                "fn": fn, "fp": fp}

    def export_front(self, f):
        info = self.front_info(f)
        a = self.S.Hip[f] + info["rank"]
        return info, self.Cblk[f][:info["csize"]].copy(), self.Hii[a:a + info["cm"]].copy()

    def import_front(self, f, fm, rank, cm, Cb, rows):
        self.Hm[f], self.Hr[f], self.Cm[f] = fm, rank, cm
        self.Cblk[f] = np.ascontiguousarray(Cb, np.float64) if len(Cb) else np.zeros(1)
        a = self.S.Hip[f] + rank
        self.Hii[a:a + cm] = rows

    def download(self):
        S = self.S
        N = Result()
        own = [int(f) for f in S.Post[:S.nf] if self.group[f] >= 0]
        N.Rblock_off = np.zeros(max(S.nf, 1), I64)
        run = 0
        for f in S.Post[:S.nf]:
            f = int(f)
            N.Rblock_off[f] = run
            if self.group[f] >= 0:
                run += self.RH[f].size
        N.rh_total = run
        N.Stack = np.zeros(max(run, 1))
        for f in own:
            N.Stack[N.Rblock_off[f]:N.Rblock_off[f] + self.RH[f].size] = self.RH[f]
        N.Rdead, N.HStair, N.HTau, N.Hii = self.Rdead, self.HStair, self.HTau, self.Hii
        N.Hm = np.where(self.group >= 0, self.Hm[:S.nf], 0).astype(I64)
        N.Hr = np.where(self.group >= 0, self.Hr[:S.nf], 0).astype(I64)
        N.HPinv = np.zeros(max(S.m, 1), I64)
        N.rank = int(N.Hr.sum()); N.rank1 = N.rank; N.maxfrank = self.maxfrank; N.maxfm = int(N.Hm.max(initial=0))
        N.stats = {"flops": self.flops}
        return N
