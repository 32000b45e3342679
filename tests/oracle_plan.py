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
        self.shared = np.zeros(S.nf, bool)
        L = self.L
        L.orc_fsize.restype = C.c_long
        L.orc_fsize.argtypes = [C.c_long] + [c_long_p] * 9
        L.orc_assemble.restype = None
        L.orc_assemble.argtypes = [C.c_long, C.c_long] + [c_long_p] * 8 + [c_double_p, c_long_p, c_long_p,
                                                                          C.POINTER(c_double_p), c_long_p, c_long_p,
                                                                          c_long_p, c_long_p, c_double_p, c_long_p]

    SHARED = 1 << 30
    PREP, PANEL, UPDATE, GRAM, POST = 1, 2, 4, 8, 16
    NB = 32

    def set_groups(self, group):
        g = np.ascontiguousarray(group, np.int32)
        self.shared = (g >= 0) & ((g & self.SHARED) != 0)
        self.group = np.where(g >= 0, g & ~self.SHARED, -1).astype(np.int32)

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
        self.open = {}; self.part = {}; self.fflops = {}

    def _dims(self, f):
        S = self.S
        return int(S.Rp[f + 1] - S.Rp[f]), int(S.Super[f + 1] - S.Super[f]), int(S.Super[f])

    def _assemble(self, f):
        S, L = self.S, self.L
        Stair = self.HStair[S.Rp[f]:]
        fm = L.orc_fsize(f, _ip(S.Super), _ip(S.Rp), _ip(S.Rj), _ip(S.Sleft), _ip(S.Child), _ip(S.Childp),
                         _ip(self.Cm), _ip(self.Fmap), _ip(Stair))
        fn, fp, col1 = self._dims(f)
        self.Hm[f] = fm
        F = np.zeros(max(fm * fn, 1))
        ptrs = (c_double_p * (S.nf + 1))()
        for q in range(S.Childp[f], S.Childp[f + 1]):
            c = int(S.Child[q])
            ptrs[c] = _dp(self.Cblk[c])
        L.orc_assemble(f, fm, _ip(S.Super), _ip(S.Rp), _ip(S.Rj), _ip(S.Sp), _ip(S.Sj), _ip(S.Sleft), _ip(S.Child),
                       _ip(S.Childp), _dp(self.Sx), _ip(self.Fmap), _ip(self.Cm), ptrs, _ip(self.Hr), _ip(Stair),
                       _ip(self.Hii), _ip(S.Hip), _dp(F), _ip(self.Cmap))
        return fm, F

    def _factor(self, f, fm, F):
        S, L = self.S, self.L
        ch = self.orc.chunk()
        W = np.zeros(ch.fchunk * max(S.maxfn, 1) + 64)
        fn, fp, col1 = self._dims(f)
        fl = C.c_double(0)
        frank = L.orc_front(fm, fn, fp, float(self.tol), int(self.ntol - col1), C.byref(ch), _dp(F), _ip(self.HStair[S.Rp[f]:]),
                            C.cast(self.Rdead[col1:].ctypes.data, C.c_char_p), _dp(self.HTau[S.Rp[f]:]), _dp(W), C.byref(fl))
        return int(frank), fl.value

    def _pack(self, f, fm, F, frank, fl):
        S, L = self.S, self.L
        fn, fp, col1 = self._dims(f)
        self.flops += fl
        self.fflops[f] = fl
        self.maxfrank = max(self.maxfrank, int(frank))
        csize = L.orc_fcsize(fm, fn, fp, frank)
        Cb = np.zeros(max(csize, 1))
        self.Cm[f] = L.orc_cpack(fm, fn, fp, frank, _dp(F), _dp(Cb))
        self.Cblk[f] = Cb
        R = np.zeros(max(fm * fn, 1)); rm = C.c_long(0)
        rs = L.orc_rhpack(fm, fn, fp, _ip(self.HStair[S.Rp[f]:]), _dp(F), _dp(R), C.byref(rm))
        self.Hr[f] = rm.value
        self.RH[f] = R[:rs].copy()

    def run_group(self, g, detail=False):
        S = self.S
        for f in S.Post[:S.nf]:
            f = int(f)
            if self.group[f] != g:
                continue
            assert not self.shared[f], "a shared front is driven by run_step"
            fm, F = self._assemble(f)
            frank, fl = self._factor(f, fm, F)
            self._pack(f, fm, F, frank, fl)

    # ---- a shared front, step by step (include/stmmqr_hip.h).  The stand-in factorizes the WHOLE front in the PANEL call of
    # step 0 (orc_front has no panel entry point) and ships the finished front in the message of panel 0; the later panel
    # messages are empty.  What each rank keeps at POST is poisoned (NaN) outside the columns it owns, so a merge that takes a
    # column from the wrong rank cannot go unnoticed. ----
    def _shared_front_of(self, g):
        fs = [int(f) for f in np.nonzero(self.group == g)[0]]
        assert len(fs) == 1 and self.shared[fs[0]]
        return fs[0]

    def group_steps(self, g):
        f = self._shared_front_of(g)
        fn, _, _ = self._dims(f)
        return max(1, (min(fn, int(self.S.Fm[f])) + self.NB - 1) // self.NB)

    def panel_doubles(self, f):
        fn, fp, _ = self._dims(f)
        return int(self.S.Fm[f]) * fn + 2 * fn + fp + 4

    def run_step(self, g, step, what, cb_first=0, cb_stride=1, cb_count=-1):
        f = self._shared_front_of(g)
        if what & self.PREP:
            fm, F = self._assemble(f)
            self.open[f] = [fm, F, None, 0.0]
        if (what & self.PANEL) and step == 0:
            st = self.open[f]
            st[2], st[3] = self._factor(f, st[0], st[1])
        if what & (self.UPDATE | self.GRAM):
            self.part[f] = ((cb_first + step + 1) % cb_stride, cb_stride)
        if what & self.POST:
            fm, F, frank, fl = self.open.pop(f)
            self._pack(f, fm, F, frank, fl)
            place, R = self.part[f]
            fn, fp, _ = self._dims(f)
            off = self.front_rhoff(f, fn)
            for k in range(fn):
                if (k // self.NB) % R != place:
                    self.RH[f][off[k]:off[k + 1]] = np.nan
            for j in range(fn - fp):
                if ((fp + j) // self.NB) % R != place:
                    self.Cblk[f][self._coff(f, j):self._coff(f, j + 1)] = np.nan

    def export_panel(self, f, p):
        buf = np.zeros(self.panel_doubles(f))
        if p == 0:
            S = self.S
            fm, F, frank, fl = self.open[f]
            fn, fp, col1 = self._dims(f)
            buf[:fm * fn] = F[:fm * fn]
            o = int(S.Fm[f]) * fn
            buf[o:o + fn] = self.HStair[S.Rp[f]:S.Rp[f] + fn]
            buf[o + fn:o + 2 * fn] = self.HTau[S.Rp[f]:S.Rp[f] + fn]
            buf[o + 2 * fn:o + 2 * fn + fp] = self.Rdead[col1:col1 + fp]
            buf[o + 2 * fn + fp:] = [fm, frank, fl, 1]
        return buf

    def import_panel(self, f, p, buf):
        if p != 0:
            return
        S = self.S
        fn, fp, col1 = self._dims(f)
        o = int(S.Fm[f]) * fn
        fm, frank, fl, tag = buf[o + 2 * fn + fp:o + 2 * fn + fp + 4]
        assert tag == 1 and int(fm) == self.open[f][0]
        fm = int(fm)
        self.open[f] = [fm, np.array(buf[:max(fm * fn, 1)]), int(frank), float(fl)]
        self.HStair[S.Rp[f]:S.Rp[f] + fn] = buf[o:o + fn].astype(I64)
        self.HTau[S.Rp[f]:S.Rp[f] + fn] = buf[o + fn:o + 2 * fn]
        self.Rdead[col1:col1 + fp] = buf[o + 2 * fn:o + 2 * fn + fp].astype(np.int8)

    def _coff(self, f, j):
        cm = int(self.Cm[f])
        return j * (j + 1) // 2 if j < cm else cm * (cm + 1) // 2 + (j - cm) * cm

    def _col_runs(self, f, part, nparts):
        fn, fp, _ = self._dims(f)
        runs = []
        for q in range(fp // self.NB, (fn + self.NB - 1) // self.NB):
            if q % nparts != part:
                continue
            j0, j1 = max(0, q * self.NB - fp), min(fn - fp, (q + 1) * self.NB - fp)
            if j1 > j0 and self.Cm[f] > 0:
                runs.append((self._coff(f, j0), self._coff(f, j1)))
        return runs

    def front_cols_doubles(self, f, part, nparts):
        return sum(b - a for a, b in self._col_runs(f, part, nparts))

    def export_front_cols(self, f, part, nparts):
        runs = self._col_runs(f, part, nparts)
        return np.concatenate([self.Cblk[f][a:b] for a, b in runs]) if runs else np.zeros(0)

    def import_front_cols(self, f, part, nparts, buf):
        pos = 0
        for a, b in self._col_runs(f, part, nparts):
            self.Cblk[f][a:b] = buf[pos:pos + b - a]
            pos += b - a

    def front_rhoff(self, f, fn):
        """column offsets of the packed R+H block (qr_rhpack, SparseQR_factorize.c: dead pivot columns keep the R rows so far)"""
        S = self.S
        _, fp, _ = self._dims(f)
        fm = int(self.Hm[f])
        St = self.HStair[S.Rp[f]:S.Rp[f] + fn]
        off = np.zeros(fn + 1, I64)
        rm = 0
        for k in range(min(fp, fn)):
            t = int(St[k])
            if t == 0:
                t = rm
            elif rm < fm:
                rm += 1
            off[k + 1] = off[k] + t
        h = rm
        for k in range(fp, fn):
            h = min(h + 1, fm)
            off[k + 1] = off[k] + rm + max(int(St[k]) - h, 0)
        return off

    def front_flops(self, f):
        return self.fflops.get(f, 0.0), 0.0

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
