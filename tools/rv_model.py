#!/usr/bin/env python3
"""Lane-level numpy model of ONE stage of riccati_backward_rv_kernel (robotoc_amd/csrc/riccati_backward_rv.hpp).

The kernel keeps P+ and s+ in the registers of one wavefront in the f64 MFMA C layout and chains every product through
register layouts (C layout of one product = A or B operand of the next).  This file states those layouts with explicit
64-lane arrays -- mfma16() below is v_mfma_f64_16x16x4_f64 as the hardware deals its operands -- and checks one regular and
one impact stage against the CPU oracle, so that the index algebra of the kernel is proven before it costs GPU time.
Test infrastructure (imports oracle/): run as  python tools/rv_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LANES = np.arange(64)
LI, Q = LANES & 15, LANES >> 4


def mfma16(a, b, c):
    """D = A B + C, 16x16x4: a[lane] = A[m = li][k = q], b[lane] = B[k = q][n = li], c[lane][r] = C[row = q + 4r][col = li]."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LI, Q] = a
    B[Q, LI] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[Q + 4 * r, LI]
    return out


def row_shift(v, n):
    """DPP row_shr:n (n > 0: lane li takes the value of lane li - n of its 16-lane row; lanes li < n get 0) / row_shl:-n."""
    out = np.zeros_like(v)
    for lane in range(64):
        src = (lane & 15) - n
        if 0 <= src < 16:
            out[lane] = v[(lane & ~15) + src]
    return out


def stage(NV, NU, pp, sv, A, Bv, Qxx, Qxu, Quu, Fx, lx, lu, impact, sc=None, sa=False):
    """pp[kt][mt]: (64,4) tiles of P+ in C layout, zero outside NX x NX; sv[c]: (64,4), s+[16c + 4r + q] on lanes li == SCOL.
    sc = (Phix [ns x NX], Phiu [ns x NU], Pres [ns]) on a grid point with a switching constraint (riccati_factorizer.cpp:58-89).
    Returns (pp_new, sv_new, K, k) and, with sc, (.., M, m)."""
    NX = 2 * NV
    T = (NX + 15) // 16                      # tiles of the state
    TM = (NX + NU + 15) // 16                # row tiles of S = [P+; PB^T]
    assert NX + NU == 16 * TM and T == TM    # the stacked operand fills its tiles exactly (ANYmal: 36 + 12 = 48)
    SCOL = NX - 16 * (T - 1)                 # lane of the rider column NX in the last column tile (ANYmal: 4)
    SH = SCOL                                # lane shift of the control columns: u <-> li = u + SH: row NX + u of S is lane u + SH of tile TM - 1
    assert SH == 16 - NU
    G0, G1 = NV // 4, (NX + 3) // 4          # aligned k groups that meet the rows [NV, NX)
    KG = (NX + 3) // 4
    z4 = lambda: np.zeros((64, 4))
    # ---- PB = P+[:, v] Bv with the control columns shifted by SH lanes ----
    acc = [z4() for _ in range(T)]
    gacc = z4()
    if not impact:
        for g in range(G0, G1):
            k = 4 * g + Q - NV
            u = LI - SH
            b = np.where((k >= 0) & (k < NV) & (u >= 0), Bv[np.clip(k, 0, NV - 1), np.clip(u, 0, NU - 1)], 0.0)
            for c in range(T):
                a = pp[g // 4][c][:, g % 4]          # P+[16c + li][4g + q] = P+[4g + q][16c + li]
                acc[c] = mfma16(a, b, acc[c])
        # ---- G = Quu + Bv^T PB[v, :] chained from the accumulators; rider column li = 0: Bv^T s+_v ----
        for g in range(G0, G1):
            k = 4 * g + Q - NV
            a = np.where((k >= 0) & (k < NV) & (LI < NU), Bv[np.clip(k, 0, NV - 1), np.clip(LI, 0, NU - 1)], 0.0)   # Bv^T[u' = li][k]
            srow = row_shift(sv[g // 4][:, g % 4], -SCOL)            # s+[4g + q] from lane li = SCOL to lane li = 0
            b = np.where(LI == 0, np.where(4 * g + Q < NX, srow, 0.0), acc[g // 4][:, g % 4])
            gacc = mfma16(a, b, gacc)
        G = Quu.copy()
        bts = np.zeros(NU)
        for r in range(4):
            u0, u1 = Q + 4 * r, LI - SH
            for lane in range(64):
                if u0[lane] < NU and u1[lane] >= 0:
                    G[u0[lane], u1[lane]] += gacc[lane, r]
                if u0[lane] < NU and LI[lane] == 0:
                    bts[u0[lane]] = gacc[lane, r]
        Y = np.linalg.inv(np.linalg.cholesky(G))   # wave_llt_inv
    # ---- W = [P+; PB^T] [A | Fx]; accumulators of the PB^T rows start from Qxu^T, so that they end as H^T ----
    pa = [[z4() for _ in range(T)] for _ in range(TM)]
    if not impact:
        for c in range(T):
            for r in range(4):
                u, x = 16 * (TM - 1) + 4 * r + Q - NX, 16 * c + LI
                ok = (u >= 0) & (x < NX)
                pa[TM - 1][c][:, r] = np.where(ok, Qxu[np.clip(x, 0, NX - 1), np.clip(u, 0, NU - 1)], 0.0)
    NP = NV - NU
    nmf = 0
    in_R = lambda k: (k < NP) | (k >= NV)          # rows of A that are dense (corner rows, velocity rows)
    for g in range(KG):
        k = 4 * g + Q
        kok = k < NX
        bs, bsm = [], []
        for c in range(T):
            j = 16 * c + LI
            b = np.where(kok & (j < NX), A[np.clip(k, 0, NX - 1), np.clip(j, 0, NX - 1)], 0.0)
            if c == T - 1:
                b = np.where(kok & (LI == SCOL), Fx[np.clip(k, 0, NX - 1)], b)
            bs.append(b)
            bsm.append(np.where(in_R(k), b, 0.0))   # structured rows masked
        for tm in range(TM):
            a = pp[g // 4][tm][:, g % 4] if tm < T else z4()[:, 0]
            if tm == TM - 1:   # lanes li >= SH: PB^T[u = li - SH][4g + q] = acc at the same lane
                a = np.where(LI >= SH, acc[g // 4][:, g % 4], a)
            for c in range(T):
                if sa and c < T - 1 and tm < TM - 1:
                    # structured column tiles, P+ row tiles: only the k groups that meet dense rows; corner rows (g with rows < NP) only
                    # reach the column tiles that hold corner columns -- all of them here (columns [0, NP) in tile 0, [NV, NV + NP) in tile 1)
                    if not (in_R(4 * g + np.arange(4))).any():
                        continue
                    pa[tm][c] = mfma16(a, bsm[c], pa[tm][c])
                else:
                    pa[tm][c] = mfma16(a, bs[c], pa[tm][c])
                nmf += 1
    if sa:
        # structured rows k in [NP, NV) of A: W[:, k] += a S[:, k], W[:, NV + k] += c S[:, k] for the P+ row tiles of the structured
        # column tiles (the PB^T rows and the last column tile went through the dense products above)
        ca, cc = A[NP, NP], A[NP, NV + NP]
        for tm in range(TM - 1):
            for c in range(T - 1):
                for r in range(4):
                    j = 16 * c + LI
                    # a-part: column j = k, same tile, lane and register of P+
                    isa = (j >= NP) & (j < NV)
                    pa[tm][c][:, r] += np.where(isa, ca * pp[tm][c][:, r], 0.0)
                    # c-part: column j = NV + k: k = j - NV = 16 (c - 1) + (li - 2): lanes li >= 2 read lane li - 2 of tile c - 1,
                    # lanes li < 2 read lane li + 14 of tile c - 2
                    isc = (j >= NV + NP) & (j < NX)
                    if c >= 1:
                        src = row_shift(pp[tm][c - 1][:, r], NV % 16)
                        pa[tm][c][:, r] += np.where(isc & (LI >= NV % 16), cc * src, 0.0)
                    if c >= 2:
                        raise NotImplementedError
    # column NX: z = s+ - P+ Fx (rows < NX), lu' = lu - Bv^T s+_v + PB^T Fx (rows NX..)
    lup = np.zeros((64, 4))
    for tm in range(TM):
        for r in range(4):
            row = 16 * tm + 4 * r + Q
            v = pa[tm][T - 1][:, r]
            isz = (LI == SCOL) & (row < NX)
            pa[tm][T - 1][:, r] = np.where(isz, sv[tm][:, r] - v, v)
            if not impact:
                u = row - NX
                for lane in range(64):
                    if LI[lane] == SCOL and 0 <= u[lane] < NU:
                        lup[lane, u[lane] // 4] = lu[u[lane]] - bts[u[lane]] + v[lane]   # u = 4 ks + q
    # ---- F = Qxx + A^T W (upper tiles), column NX starts from -lx and ends as w = A^T z - lx ----
    f = [[None] * T for _ in range(T)]
    for c in range(T):
        for t in range(c, T):
            f[c][t] = z4()
            for r in range(4):
                i, j = 16 * c + 4 * r + Q, 16 * t + LI
                ok = (i < NX) & (j < NX)
                ii, jj = np.clip(i, 0, NX - 1), np.clip(j, 0, NX - 1)
                v = np.where(ok, Qxx[ii, jj], 0.0)
                if t > c:
                    v = 0.5 * (v + np.where(ok, Qxx[jj, ii], 0.0))
                if t == T - 1:
                    v = np.where((LI == SCOL) & (i < NX), -lx[ii], v)
                f[c][t][:, r] = v
    for g in range(KG):
        k = 4 * g + Q
        kok = k < NX
        if sa and not (in_R(4 * g + np.arange(4))).any():
            continue
        for c in range(T):
            m = 16 * c + LI
            a = np.where(kok & (m < NX), A[np.clip(k, 0, NX - 1), np.clip(m, 0, NX - 1)], 0.0)   # A^T[m][k]
            if sa:
                a = np.where(in_R(k), a, 0.0)
                # corner rows only meet the corner columns of A: tiles 0 and 1 here
                if (4 * g + 3 < NP) and c == T - 1:
                    continue
            for t in range(c, T):
                f[c][t] = mfma16(a, pa[g // 4][t][:, g % 4], f[c][t])
                nmf += 1
    if sa:
        ca, cc = A[NP, NP], A[NP, NV + NP]
        for c in range(T):
            for t in range(c, T):
                for r in range(4):
                    i = 16 * c + 4 * r + Q
                    # a-part: F[i][:] += a W[i][:] for i in [NP, NV): same lane and register
                    f[c][t][:, r] += np.where((i >= NP) & (i < NV), ca * pa[c][t][:, r], 0.0)
                    # c-part: F[i][:] += c W[i - NV][:] for i in [NV + NP, NX); i - NV = 4 (g' - 5) + (q + 2) for q < 2, 4 (g' - 4) + (q - 2) else
                    gp = 4 * c + r
                    if 4 * gp + 3 >= NV + NP and 4 * gp < NX:
                        ghi, glo = gp - (NV - 2) // 4, gp - (NV + 2) // 4
                        hi_v = pa[ghi // 4][t][:, ghi % 4] if ghi >= 0 else np.zeros(64)
                        lo_v = pa[glo // 4][t][:, glo % 4] if glo >= 0 else np.zeros(64)
                        send = np.where(Q < 2, hi_v, lo_v)
                        got = send[LANES ^ 32]
                        f[c][t][:, r] += np.where((i >= NV + NP) & (i < NX), cc * got, 0.0)
    stage.mfma_wf = nmf
    K = np.zeros((NU, NX))
    kv = np.zeros(NU)
    if not impact:
        # ---- policy: Z^T = Y [H^T | -lu'],  [K | -k] = -Y^T Z^T ----
        KSU = (NU + 3) // 4
        zt = [z4() for _ in range(T)]
        kk = [z4() for _ in range(T)]
        for ks in range(KSU):
            u = 4 * ks + Q
            a = np.where((u < NU) & (LI < NU), Y[np.clip(LI, 0, NU - 1), np.clip(u, 0, NU - 1)], 0.0)       # Y[i = li][u]
            for c in range(T):
                g = 4 * (TM - 1) + ks + (NX - 16 * (TM - 1)) // 4   # register group of row NX + 4 ks + q in the last row tile
                b = pa[g // 4][c][:, g % 4]
                if c == T - 1:
                    b = np.where(LI < SCOL, b, np.where(LI == SCOL, -lup[:, ks], 0.0))
                b = np.where(u < NU, b, 0.0)
                zt[c] = mfma16(a, b, zt[c])
        ys = zt
        Mout = mv = None
        if sc is not None:
            # ---- switching constraint, factorised: with Zd = Y Phiu^T, S = Zd^T Zd = Ls Ls^T, Ws = Ls^-1,
            #      Eh = Ws (Zd^T Zh - [Phix | -Pres]),  M = -Ws^T Eh,  Ys = Zh + Zd M,  K = -Y^T Ys,  F -= Zh^T Zh - Eh^T Eh ----
            Phix, Phiu, Pres = sc
            ns = Phix.shape[0]
            zd, zdt = z4(), z4()
            for ks in range(KSU):
                u = 4 * ks + Q
                # zd = Y Phiu^T: A = Y[m = i = li][k = u], B = Phiu^T[k = u][n = l = li]
                a = np.where((u < NU) & (LI < NU), Y[np.clip(LI, 0, NU - 1), np.clip(u, 0, NU - 1)], 0.0)
                b = np.where((u < NU) & (LI < ns), Phiu[np.clip(LI, 0, ns - 1), np.clip(u, 0, NU - 1)], 0.0)
                zd = mfma16(a, b, zd)
                # zdt = Phiu Y^T: A = Phiu[m = l = li][k = u'], B = Y^T[k = u'][n = u = li] = Y[li][u']
                a = np.where((u < NU) & (LI < ns), Phiu[np.clip(LI, 0, ns - 1), np.clip(u, 0, NU - 1)], 0.0)
                b = np.where((u < NU) & (LI < NU), Y[np.clip(LI, 0, NU - 1), np.clip(u, 0, NU - 1)], 0.0)
                zdt = mfma16(a, b, zdt)
            S = z4()
            for ks in range(KSU):
                S = mfma16(zd[:, ks], zd[:, ks], S)       # A' = C^T: Zd^T (m = l, k = u); B' = C: Zd (k = u, n = l)
            Sm = np.eye(16)
            for r in range(4):
                for lane in range(64):
                    i, j = Q[lane] + 4 * r, LI[lane]
                    if i < ns and j < ns:
                        Sm[i, j] = S[lane, r]
            Ws = np.linalg.inv(np.linalg.cholesky(Sm[:ns, :ns]))        # wave_llt_inv
            Wsp = np.zeros((16, 16))
            Wsp[:ns, :ns] = Ws
            NSK = (ns + 3) // 4
            t1 = [z4() for _ in range(T)]
            for c in range(T):
                for r in range(4):
                    l, x = Q + 4 * r, 16 * c + LI
                    v = np.where((l < ns) & (x < NX), -Phix[np.clip(l, 0, ns - 1), np.clip(x, 0, NX - 1)], 0.0)
                    if c == T - 1:
                        v = np.where((l < ns) & (LI == SCOL), Pres[np.clip(l, 0, ns - 1)], v)
                    t1[c][:, r] = v
            for ks in range(KSU):
                for c in range(T):
                    t1[c] = mfma16(zd[:, ks], zt[c][:, ks], t1[c])
            eh = [z4() for _ in range(T)]
            mm = [z4() for _ in range(T)]
            for ks in range(NSK):
                l2 = 4 * ks + Q
                a = np.where(l2 < 16, Wsp[LI, np.clip(l2, 0, 15)], 0.0)        # Ws[m = l = li][k = l']
                for c in range(T):
                    eh[c] = mfma16(a, t1[c][:, ks], eh[c])
            for ks in range(NSK):
                l2 = 4 * ks + Q
                a = -Wsp[np.clip(l2, 0, 15), LI]                                # -Ws^T[m = l = li][k = l'] = -Ws[l'][l]
                for c in range(T):
                    mm[c] = mfma16(a, eh[c][:, ks], mm[c])
            ys = [zt[c].copy() for c in range(T)]
            for ks in range(NSK):
                for c in range(T):
                    ys[c] = mfma16(zdt[:, ks], mm[c][:, ks], ys[c])             # A' = C^T of zdt: Zd (m = u, k = l)
            Mout = np.zeros((ns, NX))
            mv = np.zeros(ns)
            for c in range(T):
                for r in range(4):
                    for lane in range(64):
                        l, x = Q[lane] + 4 * r, 16 * c + LI[lane]
                        if l < ns and x < NX:
                            Mout[l, x] = mm[c][lane, r]
                        if l < ns and x == NX:
                            mv[l] = -mm[c][lane, r]
        for ks in range(KSU):
            i = 4 * ks + Q
            a = np.where((i < NU) & (LI < NU), -Y[np.clip(i, 0, NU - 1), np.clip(LI, 0, NU - 1)], 0.0)      # -Y^T[u = li][i]
            for c in range(T):
                kk[c] = mfma16(a, ys[c][:, ks], kk[c])
        for c in range(T):
            for r in range(4):
                u, x = Q + 4 * r, 16 * c + LI
                for lane in range(64):
                    if u[lane] < NU and x[lane] < NX:
                        K[u[lane], x[lane]] = kk[c][lane, r]
                    if u[lane] < NU and x[lane] == NX:
                        kv[u[lane]] = -kk[c][lane, r]
        # ---- F -= Z Z^T; column NX: w - H k ----
        for ks in range(KSU):
            for c in range(T):
                for t in range(c, T):
                    f[c][t] = mfma16(-zt[c][:, ks], zt[t][:, ks], f[c][t])
        if sc is not None:
            for ks in range(NSK):
                for c in range(T):
                    for t in range(c, T):
                        f[c][t] = mfma16(eh[c][:, ks], eh[t][:, ks], f[c][t])
    # ---- s+ <- column NX; P+ <- sym(F): upper tiles as they are, diagonal tiles mirrored, lower tiles transposed ----
    sv_new = [np.where((LI == SCOL)[:, None] & ((16 * c + 4 * np.arange(4)[None, :] + Q[:, None]) < NX), f[c][T - 1], 0.0) for c in range(T)]

    def transpose_tile(tile):
        scr = np.zeros((16, 16))
        for r in range(4):
            scr[Q + 4 * r, LI] = tile[:, r]
        out = np.zeros((64, 4))
        for r in range(4):
            out[:, r] = scr[LI, Q + 4 * r]
        return out

    def mask(tile, c, t):
        out = tile.copy()
        for r in range(4):
            i, j = 16 * c + 4 * r + Q, 16 * t + LI
            out[:, r] = np.where((i < NX) & (j < NX), tile[:, r], 0.0)
        return out
    pn = [[None] * T for _ in range(T)]
    for c in range(T):
        for t in range(c, T):
            if t == c:
                tr = transpose_tile(f[c][c])
                d = z4()
                for r in range(4):
                    d[:, r] = np.where(Q + 4 * r <= LI, f[c][c][:, r], tr[:, r])
                pn[c][c] = mask(d, c, c)
            else:
                pn[c][t] = mask(f[c][t], c, t)
                pn[t][c] = mask(transpose_tile(mask(f[c][t], c, t)), t, c)
    if sc is not None:
        return pn, sv_new, K, kv, Mout, mv
    return pn, sv_new, K, kv


def to_tiles(P, NX):
    T = (NX + 15) // 16
    out = [[np.zeros((64, 4)) for _ in range(T)] for _ in range(T)]
    for kt in range(T):
        for mt in range(T):
            for r in range(4):
                i, j = 16 * kt + 4 * r + Q, 16 * mt + LI
                ok = (i < NX) & (j < NX)
                out[kt][mt][:, r] = np.where(ok, P[np.clip(i, 0, NX - 1), np.clip(j, 0, NX - 1)], 0.0)
    return out


def from_tiles(pp, NX):
    T = (NX + 15) // 16
    P = np.zeros((NX, NX))
    for kt in range(T):
        for mt in range(T):
            for r in range(4):
                for lane in range(64):
                    i, j = 16 * kt + 4 * r + Q[lane], 16 * mt + LI[lane]
                    if i < NX and j < NX:
                        P[i, j] = pp[kt][mt][lane, r]
    return P


def main():
    from oracle import oracle as orc
    from robotoc_amd import problems as pr
    from robotoc_amd.types import GRID_IMPACT, Records
    dims, grids, _ = pr.config_anymal_trot()
    L = orc.layout(dims)
    NV, NU, NX = dims.nv, dims.nu, 2 * dims.nv
    kkt = pr.make_kkt_batch_unique(L, grids, 1, seed=3)[0]
    Kr, Rr = Records(L, "kkt"), Records(L, "ric")
    ric = Rr.zeros(1, len(grids))[0]
    kk = kkt.copy()
    orc.riccati_backward(L, grids, kk, ric)
    worst = 0.0
    SA = "--dense" not in sys.argv
    for st in (45, 35, 33, 20, 15):
        g = grids[st]
        rec, nxt, out = kkt[st], ric[st + 1], ric[st]
        P1 = Rr.f(nxt, "P").copy()
        s1 = Rr.f(nxt, "s").copy()
        T = (NX + 15) // 16
        SCOL = NX - 16 * (T - 1)
        sv = [np.zeros((64, 4)) for _ in range(T)]
        for c in range(T):
            for r in range(4):
                i = 16 * c + 4 * r + Q
                sv[c][:, r] = np.where((LI == SCOL) & (i < NX), s1[np.clip(i, 0, NX - 1)], 0.0)
        f = lambda n: Kr.f(rec, n).copy()
        sc = None
        if g.type != GRID_IMPACT and g.dims > 0:
            sc = (f("Phix")[:g.dims], f("Phiu")[:g.dims], f("Pres")[:g.dims])
        out_ = stage(NV, NU, to_tiles(P1, NX), sv, f("Fxx"), f("Fvu"), f("Qxx"), f("Qxu"), f("Quu"), f("Fx"), f("lx"), f("lu"),
                     g.type == GRID_IMPACT, sc, sa=SA)
        pn, svn, K, k = out_[:4]
        P = from_tiles(pn, NX)
        s = np.zeros(NX)
        for c in range(T):
            for r in range(4):
                for lane in range(64):
                    i = 16 * c + 4 * r + Q[lane]
                    if LI[lane] == SCOL and i < NX:
                        s[i] = svn[c][lane, r]
        eP = np.abs(P - Rr.f(out, "P")).max() / np.abs(Rr.f(out, "P")).max()
        es = np.abs(s - Rr.f(out, "s")).max() / max(np.abs(Rr.f(out, "s")).max(), 1e-300)
        errs = {"P": eP, "s": es, "asym": np.abs(P - P.T).max()}
        if g.type != GRID_IMPACT:
            errs["K"] = np.abs(K.T - Rr.f(out, "K")).max() / np.abs(Rr.f(out, "K")).max()
            errs["k"] = np.abs(k - Rr.f(out, "k")).max() / max(np.abs(Rr.f(out, "k")).max(), 1e-300)
        if sc is not None:
            errs["M"] = np.abs(out_[4] - Rr.f(out, "M")[:g.dims]).max() / np.abs(Rr.f(out, "M")[:g.dims]).max()
            errs["m"] = np.abs(out_[5] - Rr.f(out, "m")[:g.dims]).max() / max(np.abs(Rr.f(out, "m")[:g.dims]).max(), 1e-300)
        print("W + F products: %d MFMAs;" % stage.mfma_wf, end=" ")
        print("stage", st, "type", g.type, "dims", g.dims, {n: float("%.2e" % v) for n, v in errs.items()})
        worst = max(worst, max(v for n, v in errs.items()))
    assert worst < 1e-10, worst
    print("rv lane model: ok")


if __name__ == "__main__":
    main()
