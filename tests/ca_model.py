"""Numpy model of the Gram-based panel (csrc/stmmqr_capanel.hip): Householder QR of a panel where every reduction over the bottom rows is
replaced by ONE Gram matrix G = B'B (B = rows below the panel's pivot rows), downdated column by column; a column
whose downdated norm lost more than K of its magnitude triggers a refresh (real Gram of the remaining columns).
Evaluated on the real fronts of a fixture: trips, accuracy vs the oracle front."""
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, "tests")
from stmmqr_testlib import Oracle, Symbolic, load_golden, scalar, _ip, _dp, I64, c_double_p
from oracle_plan import OraclePlan

NB = 32
MODE = "ca"          # "cab": the blocked form (ca_panel_blocked)
STATS = {"panels": 0, "refresh": 0, "cols": 0, "maxerr": 0.0}
BUCK = {}


def ca_panel(F, Stair, Tau, Rdead, k1, nb, g, rank, npiv, ntol, tol, K):
    """column-owned formulation: G0 fixed since the last refresh; M (pending column ops) and Y = G0 M updated by the
    same column operation; Gram entries on demand: G_cur[j][x] = y_j' m_x"""
    m, n = F.shape
    g1 = g
    k2 = k1 + nb
    tmax = min(m, max(int(Stair[k2 - 1]), g1 + nb))
    r_top1 = min(tmax, g1 + nb)
    At = F[g1:r_top1, k1:k2].copy()
    B = F[r_top1:tmax, k1:k2]
    M = np.eye(nb)
    Y = B.T @ B                                  # Y = G0 M
    Gref = np.diag(Y).copy() + (At * At).sum(axis=0)
    jref = 0
    diag = [None] * nb
    done = False
    tlast = g1
    STATS["panels"] += 1
    rows_ = tmax - g1
    bk = 0 if rows_ <= 64 else 1 if rows_ <= 256 else 2 if rows_ <= 512 else 3 if rows_ <= 1024 else 4 if rows_ <= 2048 else 5
    BUCK.setdefault(bk, [0, 0]); BUCK[bk][0] += 1
    for j in range(nb):
        k = k1 + j
        if g >= m:
            for kk in range(k, n):
                if kk < npiv:
                    Rdead[kk] = 1; Stair[kk] = 0
                else:
                    Stair[kk] = m
                Tau[kk] = 0
            done = True
            break
        t = max(g + 1, int(Stair[k]))
        gi = g - g1
        alpha = At[gi, j]
        stop = float(At[gi + 1:, j] @ At[gi + 1:, j])
        gjj = float(Y[:, j] @ M[:, j])
        sabs = float(np.abs(Y[:, j] * M[:, j]).sum())
        ss = stop + gjj
        tot = alpha * alpha + max(ss, 0.0)
        if j > jref and B.shape[0] > 0 and (Gref[j] > K * tot or sabs > K * tot or (ss <= 0 and not (stop == 0 and gjj == 0))):
            B[:, :] = B @ M
            M = np.eye(nb)
            Y = B.T @ B
            Gref = np.diag(Y).copy() + (At[gi:, :] * At[gi:, :]).sum(axis=0)
            jref = j
            STATS["refresh"] += 1
            BUCK[bk][1] += 1
            gjj = float(Y[j, j])
            ss = stop + gjj
        ss = max(ss, 0.0)
        STATS["cols"] += 1
        if ss == 0.0:
            beta, tau, scal = alpha, 0.0, 0.0
        else:
            beta = -np.copysign(np.sqrt(alpha * alpha + ss), alpha)
            tau = (beta - alpha) / beta
            scal = 1.0 / (alpha - beta)
        dead = (k < ntol) and (abs(beta) <= tol)
        if dead:
            At[gi:, j] = 0.0
            M[:, j] = 0.0; Y[:, j] = 0.0
            Stair[k] = 0; Tau[k] = 0; Rdead[k] = 1
            diag[j] = None
            if k == npiv - 1:
                rank = g
            continue
        Stair[k] = t; Tau[k] = tau; diag[j] = g
        if tau != 0.0:
            vtop = At[gi + 1:, j] * scal
            gjx = Y[:, j] @ M[:, j + 1:]                   # G_cur[j][x] = y_j' m_x
            w = At[gi, j + 1:] + vtop @ At[gi + 1:, j + 1:] + scal * gjx
            cw = tau * w
            At[gi, j + 1:] -= cw
            At[gi + 1:, j + 1:] -= np.outer(vtop, cw)
            At[gi + 1:, j] = vtop
            c = cw * scal
            M[:, j + 1:] -= np.outer(M[:, j], c)
            Y[:, j + 1:] -= np.outer(Y[:, j], c)
            M[:, j] *= scal
            Y[:, j] *= scal
        else:
            M[:, j] = 0.0; Y[:, j] = 0.0
        At[gi, j] = beta
        tlast = t
        g += 1
        if k == npiv - 1:
            rank = g
    B[:, :] = B @ M
    F[g1:r_top1, k1:k2] = At
    return g, rank, done, diag, tlast


def ca_panel_blocked(F, Stair, Tau, Rdead, k1, nb, g, rank, npiv, ntol, tol, K, SB=8):
    """Blocked form of ca_panel (csrc/stmmqr_capanel.hip, round 3): the panel is factorized in sub-blocks of SB columns.
    Inside a sub-block only its SB candidate pivot rows P are explicit; every other row -- the rest of the top block (Lt)
    and the bottom rows B -- enters through one SB x (remaining columns) Gram block GL, so the column chain works on
    SB x SB matrices (T1 = At[P, sub-block], GL11, M11: one entry per lane of ONE wave, no workgroup barrier per column);
    the sub-block's reflectors are then applied to the columns behind it as ONE block reflector on the small matrices
    (At, M, G).  Refresh rule as in ca_panel, checked inside the chain (the chain stops, what it finished is applied, the
    Gram matrix is formed again from the real rows)."""
    m, n = F.shape
    g1 = g
    k2 = k1 + nb
    tmax = min(m, max(int(Stair[k2 - 1]), g1 + nb))
    r_top1 = min(tmax, g1 + nb)
    At = F[g1:r_top1, k1:k2].copy()
    nt = At.shape[0]
    B = F[r_top1:tmax, k1:k2]
    nB = B.shape[0]
    M = np.eye(nb)
    G = B.T @ B
    Gref = np.diag(G).copy() + (At * At).sum(axis=0)
    jref = -1
    diag = [None] * nb
    done = False
    tlast = g1
    STATS["panels"] += 1
    c0 = 0
    while c0 < nb and not done:
        if g >= m:
            for kk in range(k1 + c0, n):
                if kk < npiv:
                    Rdead[kk] = 1; Stair[kk] = 0
                else:
                    Stair[kk] = m
                Tau[kk] = 0
            done = True
            break
        c1 = min(c0 + SB, nb); w = c1 - c0
        gi0 = g - g1
        npr = min(SB, nt - gi0)
        P = slice(gi0, gi0 + npr); Lt = slice(gi0 + npr, nt)
        GB11 = G[c0:c1, c0:c1].copy(); GB12 = G[c0:c1, c1:nb].copy()
        AL = At[Lt, :]
        GL = G[c0:c1, c0:nb] + AL[:, c0:c1].T @ AL[:, c0:nb]
        GL11o = GL[:, :w].copy(); GL12 = GL[:, w:].copy()
        GL11 = GL11o.copy()
        T1 = At[P, c0:c1].copy()
        M11 = np.eye(w)
        tau_l = np.zeros(w); dl = [None] * w
        nd = 0; flag = False
        for jj in range(w):
            j = c0 + jj; k = k1 + j
            if g >= m:
                break
            gi = g - g1; pl = gi - gi0
            t = max(g + 1, int(Stair[k]))
            alpha = T1[pl, jj]
            stop = float(T1[pl + 1:, jj] @ T1[pl + 1:, jj])
            gjj = float(GL11[jj, jj])
            ss = stop + gjj
            tot = alpha * alpha + max(ss, 0.0)
            below = (nt - (gi0 + npr)) + nB
            if j != jref and below > 0 and (Gref[j] > K * tot or (ss <= 0 and not (stop == 0 and gjj == 0))):
                flag = True
                break
            ss = max(ss, 0.0)
            STATS["cols"] += 1
            if ss == 0.0:
                beta, tau, scal = alpha, 0.0, 0.0
            else:
                beta = -np.copysign(np.sqrt(alpha * alpha + ss), alpha)
                tau = (beta - alpha) / beta
                scal = 1.0 / (alpha - beta)
            dead = (k < ntol) and (abs(beta) <= tol)
            nd += 1
            if dead:
                T1[pl:, jj] = 0.0
                M11[:, jj] = 0.0
                Stair[k] = 0; Tau[k] = 0; Rdead[k] = 1
                if k == npiv - 1:
                    rank = g
                continue
            Stair[k] = t; Tau[k] = tau; diag[j] = g; dl[jj] = pl; tau_l[jj] = tau
            if tau != 0.0:
                vtop = T1[pl + 1:, jj] * scal
                gjx = GL11[jj, jj + 1:].copy()
                wv = T1[pl, jj + 1:] + vtop @ T1[pl + 1:, jj + 1:] + scal * gjx
                cw = tau * wv
                T1[pl, jj + 1:] -= cw
                T1[pl + 1:, jj + 1:] -= np.outer(vtop, cw)
                T1[pl + 1:, jj] = vtop
                c = cw * scal
                M11[:, jj + 1:] -= np.outer(M11[:, jj], c)
                M11[:, jj] *= scal
                GL11[jj + 1:, jj + 1:] -= np.outer(c, gjx) + np.outer(gjx, c) - gjj * np.outer(c, c)
            else:
                M11[:, jj] = 0.0
            T1[pl, jj] = beta
            tlast = t
            g += 1
            if k == npiv - 1:
                rank = g
        # ---- apply the nd reflectors of the sub-block to everything behind them ----
        At[P, c0:c1] = T1
        if nd > 0:
            Mn = M11[:, :nd]
            VP = np.zeros((npr, nd))
            for q in range(nd):
                if dl[q] is None or tau_l[q] == 0.0:
                    continue
                VP[dl[q], q] = 1.0
                VP[dl[q] + 1:, q] = T1[dl[q] + 1:, q]
            Kf = VP.T @ VP + Mn.T @ GL11o @ Mn
            Tf = np.zeros((nd, nd))
            for b_ in range(nd):
                Tf[b_, b_] = tau_l[b_]
                if b_ > 0 and tau_l[b_] != 0:
                    Tf[:b_, b_] = -tau_l[b_] * (Tf[:b_, :b_] @ Kf[:b_, b_])
            W2 = VP.T @ At[P, c1:nb] + Mn.T @ GL12
            Y2 = Tf.T @ W2
            At[P, c1:nb] -= VP @ Y2
            At[Lt, c0:c1] = At[Lt, c0:c1] @ M11
            At[Lt, c1:nb] -= At[Lt, c0:c0 + nd] @ Y2
            M[:, c0:c1] = M[:, c0:c1] @ M11
            M[:, c1:nb] -= M[:, c0:c0 + nd] @ Y2
            if not flag:
                ZB = Mn.T @ GB12
                KB = Mn.T @ GB11 @ Mn
                G[c1:nb, c1:nb] -= Y2.T @ ZB + ZB.T @ Y2 - Y2.T @ KB @ Y2
        if flag:
            if nB > 0:
                B[:, :] = B @ M
                M = np.eye(nb)
                G = B.T @ B
            gi = g - g1
            Gref = np.diag(G).copy() + (At[gi:, :] * At[gi:, :]).sum(axis=0)
            jref = c0 + nd
            STATS["refresh"] += 1
        c0 += nd
        if nd == 0 and not flag:
            break                                   # (g >= m at the first column of the sub-block: handled at the loop head)
    if g >= m and not done and c0 < nb:
        pass
    B[:, :] = B @ M
    F[g1:r_top1, k1:k2] = At
    if not done and g >= m and c0 < nb:
        for kk in range(k1 + c0, n):
            if kk < npiv:
                Rdead[kk] = 1; Stair[kk] = 0
            else:
                Stair[kk] = m
            Tau[kk] = 0
        done = True
    return g, rank, done, diag, tlast


def classic_panel(F, Stair, Tau, Rdead, k1, nb, g, rank, npiv, ntol, tol):
    m, n = F.shape
    g1 = g; k2 = k1 + nb
    diag = [None] * nb; done = False; tlast = g1
    for j in range(nb):
        k = k1 + j
        if g >= m:
            for kk in range(k, n):
                if kk < npiv:
                    Rdead[kk] = 1; Stair[kk] = 0
                else:
                    Stair[kk] = m
                Tau[kk] = 0
            done = True
            break
        t = max(g + 1, int(Stair[k]))
        alpha = F[g, k]
        x = F[g + 1:t, k]
        ss = float(x @ x)
        if ss == 0.0:
            beta, tau, scal = alpha, 0.0, 0.0
        else:
            beta = -np.copysign(np.sqrt(alpha * alpha + ss), alpha)
            tau = (beta - alpha) / beta; scal = 1.0 / (alpha - beta)
        dead = (k < ntol) and (abs(beta) <= tol)
        if dead:
            F[g:, k] = 0; Stair[k] = 0; Tau[k] = 0; Rdead[k] = 1
            if k == npiv - 1: rank = g
            continue
        Stair[k] = t; Tau[k] = tau; diag[j] = g
        if tau != 0.0:
            v = x * scal
            w = F[g, k + 1:k2] + v @ F[g + 1:t, k + 1:k2]
            F[g, k + 1:k2] -= tau * w
            F[g + 1:t, k + 1:k2] -= np.outer(v, tau * w)
            F[g + 1:t, k] = v
        F[g, k] = beta
        tlast = t; g += 1
        if k == npiv - 1: rank = g
    return g, rank, done, diag, tlast


def front_qr(F, Stair, npiv, tol, ntol, mode, K=100.0):
    m, n = F.shape
    Tau = np.zeros(n); Rdead = np.zeros(max(npiv, 1), np.int8)
    g = 0; rank = min(m, npiv)
    ntol = min(ntol, npiv)
    for k1 in range(0, n, NB):
        nb = min(NB, n - k1)
        g1 = g
        if g >= m:
            for kk in range(k1, n):
                if kk < npiv:
                    Rdead[kk] = 1; Stair[kk] = 0
                else:
                    Stair[kk] = m
                Tau[kk] = 0
            break
        if mode == "ca":
            g, rank, done, diag, tlast = ca_panel(F, Stair, Tau, Rdead, k1, nb, g, rank, npiv, ntol, tol, K)
        elif mode == "cab":
            g, rank, done, diag, tlast = ca_panel_blocked(F, Stair, Tau, Rdead, k1, nb, g, rank, npiv, ntol, tol, K)
        else:
            g, rank, done, diag, tlast = classic_panel(F, Stair, Tau, Rdead, k1, nb, g, rank, npiv, ntol, tol)
        # trailing update with the explicit V
        k2 = k1 + nb
        if k2 < n and tlast > g1:
            V = np.zeros((tlast - g1, nb))
            for j in range(nb):
                if diag[j] is None or Tau[k1 + j] == 0.0:
                    continue
                d = diag[j] - g1
                V[d, j] = 1.0
                V[d + 1:, j] = F[diag[j] + 1:tlast, k1 + j]
            tau = np.array([Tau[k1 + j] if diag[j] is not None else 0.0 for j in range(nb)])
            Gm = V.T @ V
            T = np.zeros((nb, nb))
            for b in range(nb):
                T[b, b] = tau[b]
                if b > 0 and tau[b] != 0:
                    T[:b, b] = -tau[b] * (T[:b, :b] @ Gm[:b, b])
            Cc = F[g1:tlast, k2:]
            W = V.T @ Cc
            Cc -= V @ (T.T @ W)
        if done:
            break
    return rank, Tau, Rdead


def explicit_Q_apply(F, Stair, Tau, X):
    """X <- Q' X using the reflectors stored in F (columns with Tau != 0); diag rows follow the live order"""
    m, n = F.shape
    g = 0
    for k in range(n):
        t = int(Stair[k])
        if g >= m:
            break
        if t == 0 and Tau[k] == 0:
            # dead or past-the-end: dead pivot has Stair 0
            continue
        if Tau[k] != 0.0:
            v = np.zeros(m); v[g] = 1.0; v[g + 1:t] = F[g + 1:t, k]
            X -= np.outer(v, Tau[k] * (v @ X))
        g += 1
    return X


def main(name=None, K=None, minfn=None):
    """python tests/ca_model.py [fixture] [K] [min columns of a modelled front]: every front of the fixture with at least
    that many columns is factorized by the model next to the restated reference front (oracle); prints and returns the
    comparison (integer outputs, entries that must vanish under Q', R entries reproduced through Q')."""
    name = name or (sys.argv[1] if len(sys.argv) > 1 else "grid20_standin")
    K = K if K is not None else (float(sys.argv[2]) if len(sys.argv) > 2 else 32.0)
    minfn = minfn if minfn is not None else (int(sys.argv[3]) if len(sys.argv) > 3 else 64)
    g = load_golden(name)
    S = Symbolic(g)
    orc = Oracle()
    P = OraclePlan(S, orc)
    L = orc.lib
    tol = scalar(g, "in_tol"); ntol = int(scalar(g, "in_ntol"))
    P.begin(g["in_Ax"], tol, ntol, g["in_Ap"], g["in_Ai"])
    orig_front = L.orc_front
    res = []

    class Hook:
        def __call__(self, fm, fn, fp, tolv, ntolf, ch, Fp, Stp, Rdp, Taup, Wp, flp):
            if fn >= minfn and fm >= 64:
                F0 = np.ctypeslib.as_array(Fp, shape=(fm * fn,)).reshape((fn, fm)).T.copy(order="F")
                St0 = np.ctypeslib.as_array(Stp, shape=(fn,)).copy()
            r = orig_front(fm, fn, fp, tolv, ntolf, ch, Fp, Stp, Rdp, Taup, Wp, flp)
            if fn >= minfn and fm >= 64:
                Fo = np.ctypeslib.as_array(Fp, shape=(fm * fn,)).reshape((fn, fm)).T
                Sto = np.ctypeslib.as_array(Stp, shape=(fn,))
                Fc = F0.copy(order="F"); Stc = St0.copy()
                before = dict(STATS)
                rk, Tauc, Rdc = front_qr(Fc, Stc, fp, tolv, ntolf, MODE, K)
                # compare: integer outputs, R up to sign, backward error
                same_int = (rk == r) and np.array_equal(Stc, Sto)
                # R part: rows < g ... compare |R| upper part rows up to live count
                gl = int(sum(1 for k in range(fn) if Stc[k] != 0 and k < fn))  # rough
                nr = min(fm, fn)
                Ro = np.triu(Fo[:nr, :]); Rc = np.triu(Fc[:nr, :])
                # sign-invariant: row norms
                dn = np.abs(np.linalg.norm(Ro, axis=1) - np.linalg.norm(Rc, axis=1)).max() / (np.linalg.norm(Ro) + 1e-300)
                # backward error through Q' F0 = R
                X = explicit_Q_apply(Fc, Stc, Tauc, F0.copy())
                # after Q', X should equal the R/C content of Fc in the upper part, zero below the reflectors
                Rfull = np.zeros_like(X)
                gg = 0
                # build expected: rows of Fc above/at the staircase diag per column
                # simpler: check orthogonal invariance: norm of X columns equal F0 columns, and X below-diagonal part ~ 0
                colerr = np.abs(np.linalg.norm(X, axis=0) - np.linalg.norm(F0, axis=0)).max() / np.linalg.norm(F0)
                # entries that must vanish: rows > pivot row for each live column
                gg = 0; low = 0.0
                for k in range(fn):
                    if gg >= fm: break
                    if Stc[k] == 0 and k < fp:
                        continue
                    low = max(low, np.abs(X[gg + 1:, k]).max() if gg + 1 < fm else 0.0)
                    gg += 1
                low /= np.linalg.norm(F0)
                # match of R entries (upper part incl. C) against X
                gg = 0; rerr = 0.0
                for k in range(fn):
                    if gg >= fm: break
                    if Stc[k] == 0 and k < fp:
                        continue
                    rerr = max(rerr, np.abs(X[:gg + 1, k] - Fc[:gg + 1, k]).max())
                    gg += 1
                rerr /= np.linalg.norm(F0)
                res.append((fm, fn, fp, same_int, dn, low, rerr, STATS["refresh"] - before["refresh"], STATS["panels"] - before["panels"]))
                if fm >= 400: print(f"front {fm}x{fn} fp={fp} int_ok={same_int} dRnorm={dn:.2e} low={low:.2e} rerr={rerr:.2e} "
                      f"refresh={STATS['refresh'] - before['refresh']}/{STATS['panels'] - before['panels']} panels", flush=True)
            return r

    hook = Hook()

    class LibProxy:
        def __getattr__(self, k):
            if k == "orc_front":
                return hook
            return getattr(L, k)

    P.L = LibProxy()
    P.run_group(0)
    print("total", STATS)
    print("buckets (rows<=64,256,512,1024,2048,more): panels, refreshes", sorted(BUCK.items()))
    if res:
        a = np.array([(r[4], r[5], r[6]) for r in res])
        print("max dRnorm %.2e  max low %.2e  max rerr %.2e; int_ok all: %s" % (a[:, 0].max(), a[:, 1].max(), a[:, 2].max(), all(r[3] for r in res)))
    return res, dict(STATS)


if __name__ == "__main__":
    main()
