#!/usr/bin/env python3
"""Lane-level numpy model of the product phase of condense_rv_kernel (robotoc_amd/csrc/condense_rv.hpp).

The kernel condenses one (instance, grid point) on ONE wavefront and chains every product of
ContactDynamics::condenseContactDynamics (src/dynamics/contact_dynamics.cpp:55-164) through the register layouts of
v_mfma_f64_16x16x4_f64: the accumulator (C) layout of one product is the A operand (transposed) or the B operand (as is) of the next.
This file states those layouts with explicit 64-lane arrays and checks every output field of a contact grid point against the CPU
oracle, so that the index algebra of the kernel is proven before it costs GPU time.  The saddle inverse MJtJinv itself is taken from
the oracle (the kernel assembles it with the fragment the other condensation kernels use).
Test infrastructure (imports oracle/): run as  python tools/cond_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

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


def c_tile(M, tr, tc):
    """C layout of tile (tr, tc) of M, zero beyond its extents: out[lane][r] = M[q + 4r + 16 tr][li + 16 tc]"""
    out = np.zeros((64, 4))
    for r in range(4):
        row, col = Q + 4 * r + 16 * tr, LI + 16 * tc
        ok = (row < M.shape[0]) & (col < M.shape[1])
        out[ok, r] = M[row[ok], col[ok]]
    return out


def from_right(v, n):
    """lane li reads lane li + n of its 16-lane row (DPP row_shl:n); 0 past the row"""
    out = np.zeros_like(v)
    for lane in range(64):
        s = (lane & 15) + n
        if 0 <= s < 16:
            out[lane] = v[(lane & ~15) + s]
    return out


def from_left(v, n):
    return from_right(v, -n)


def model(NV, NU, NF, nf, ns, g, k0, c0, Lam_full, K, Cd):
    """Returns a dict of the output fields as the kernel stores them."""
    NX, NP, LDV = 2 * NV, NV - NU, NV + NF
    assert 16 <= NV <= 30 and LDV <= 32 and 32 < NX <= 46
    RC = NX - 32                 # lane of the rider column NX in column tile 2
    RU = NV + 0 - 16             # lane of the rider column NV in column tile 1 of W Lam
    dt = g.dt
    inv = 1.0 / g.num_grids_in_phase
    nvf = NV + nf
    z4 = lambda: np.zeros((64, 4))
    # ---- inputs as the kernel holds them ----
    Lam = np.zeros((32, 32))
    Lam[:LDV, :LDV] = Lam_full
    lam = [[c_tile(Lam, tr, tc) for tc in range(2)] for tr in range(2)]
    D = np.zeros((32, 48))
    D[:nvf, :NX] = Cd.f(c0, "dIDCdqv")[:nvf]
    D[:nvf, NX] = Cd.f(c0, "IDC")[:nvf]                                 # rider: LD[:, NX] = MJtJinv_IDC
    dB = [[np.array([D[4 * ks + Q[l], LI[l] + 16 * tc] for l in range(64)]) for ks in range(8)] for tc in range(3)]
    qaa = Cd.f(c0, "Qaa")
    Qff = np.zeros((NF, NF)); Qff[:nf, :nf] = Cd.f(c0, "Qff")[:nf, :nf]
    Qqf = np.zeros((NV, NF)); Qqf[:, :nf] = Cd.f(c0, "Qqf")[:, :nf]
    W1 = np.zeros((16, 16))      # W restricted to the rows / columns 16..31 of [a; f]
    for m in range(NV - 16):
        W1[m, m] = qaa[16 + m]
    W1[NV - 16:NV - 16 + NF, NV - 16:NV - 16 + NF] = Qff
    w1 = [np.array([W1[LI[l], 4 * ks + Q[l]] for l in range(64)]) for ks in range(4)]
    la, lf, ha, hf = Cd.f(c0, "la"), Cd.f(c0, "lf"), Cd.f(c0, "ha"), Cd.f(c0, "hf")
    r36 = np.zeros(32); r36[:NV] = -la; r36[NV:nvf] = lf[:nf]           # -[la; -lf]
    r37 = np.zeros(32); r37[:NV] = -ha; r37[NV:nvf] = hf[:nf]           # -[ha; -hf]
    Ep = np.zeros((32, 32))                                             # E': rows f, columns < NV: Qqf^T
    Ep[NV:NV + NF, :NV] = Qqf.T
    ep = [c_tile(Ep, 1, tc) for tc in range(2)]                         # only row tile 1 is populated (NV >= 16)
    # ---- LD = Lam D (A = Lam through its own C layout: symmetric) ----
    ld = [[z4() for _ in range(3)] for _ in range(2)]
    for tr in range(2):
        for tc in range(3):
            for ks in range(8):
                ld[tr][tc] = mfma16(lam[ks // 4][tr][:, ks % 4], dB[tc][ks], ld[tr][tc])
    # ---- Xn = -Qafqv = W LD + E' (+ riders) ----
    xn = [[z4() for _ in range(3)] for _ in range(2)]
    for tc in range(3):
        for r in range(4):
            xn[0][tc][:, r] = qaa[Q + 4 * r] * ld[0][tc][:, r]
        acc = ep[tc].copy() if tc < 2 else z4()
        for ks in range(4):
            acc = mfma16(w1[ks], ld[1][tc][:, ks], acc)
        xn[1][tc] = acc
    for tr in range(2):
        for r in range(4):
            row = Q + 4 * r + 16 * tr
            xn[tr][2][:, r] = np.where(LI == RC, xn[tr][2][:, r] + r36[row], xn[tr][2][:, r])     # column NX: -laf
            xn[tr][2][:, r] = np.where(LI == RC + 1, r37[row], xn[tr][2][:, r])                    # column NX + 1: -haf
    # LD column NX + 1 := MJtJinv_IDC / dt (A operand of the second term only: the hx rider)
    lda = [[t.copy() for t in row] for row in ld]
    for tr in range(2):
        for r in range(4):
            lda[tr][2][:, r] = np.where(LI == RC + 1, from_left(ld[tr][2][:, r], 1) / dt, ld[tr][2][:, r])
    # ---- W Lam (Qafu_full in its columns < NV); rider columns NV, NV + 1: laf, haf ----
    wl = [[z4() for _ in range(2)] for _ in range(2)]
    for tc in range(2):
        for r in range(4):
            wl[0][tc][:, r] = qaa[Q + 4 * r] * lam[0][tc][:, r]
        acc = z4()
        for ks in range(4):
            acc = mfma16(w1[ks], lam[1][tc][:, ks], acc)
        wl[1][tc] = acc
    for tr in range(2):
        for r in range(4):
            mv = from_right(xn[tr][2][:, r], RC - RU)                 # columns NX, NX + 1 of tile 2 -> lanes RU, RU + 1
            wl[tr][1][:, r] = np.where((LI == RU) | (LI == RU + 1), -mv, wl[tr][1][:, r])
    out = {}
    # ---- V = Xn^T LD + LD^T E': V[m][n] = (Qxx update)[n][m]; rows NX, NX + 1: lx, hx ----
    Qxx0, lx0, hx0 = K.f(k0, "Qxx"), K.f(k0, "lx"), K.f(k0, "hx")
    Qxx, lx, hx = np.zeros((NX, NX)), np.zeros(NX), np.zeros(NX)
    h_new = None
    for tm in range(3):
        for tn in range(3):
            acc = z4()
            for r in range(4):
                m, n = Q + 4 * r + 16 * tm, LI + 16 * tn
                ok = (m < NX) & (n < NX)
                acc[ok, r] = Qxx0[n[ok], m[ok]]
                s = (m == NX) & (n < NX)
                acc[s, r] = lx0[n[s]]
                s = (m == NX + 1) & (n < NX)
                acc[s, r] = hx0[n[s]]
                s = (m == NX + 1) & (n == NX)
                acc[s, r] = K.f(k0, "scal")[2]
            for ks in range(8):
                acc = mfma16(xn[ks // 4][tm][:, ks % 4], ld[ks // 4][tn][:, ks % 4], acc)
            if tn < 2:
                for ks in range(4):
                    acc = mfma16(lda[1][tm][:, ks], ep[tn][:, ks], acc)
            for r in range(4):
                m, n = Q + 4 * r + 16 * tm, LI + 16 * tn
                for l in range(64):
                    if m[l] < NX and n[l] < NX:
                        Qxx[n[l], m[l]] = acc[l, r]
                    elif m[l] == NX and n[l] < NX:
                        lx[n[l]] = acc[l, r]
                    elif m[l] == NX + 1 and n[l] < NX:
                        hx[n[l]] = acc[l, r] * inv
                    elif m[l] == NX + 1 and n[l] == NX:
                        h_new = acc[l, r] * inv
    out.update(Qxx=Qxx, lx=lx, hx=hx, h=h_new)
    # ---- V2 = -(Qxu_full)^T: -old + (W Lam)^T LD + Lam[a, :] E' ----
    Qxu0 = K.f(k0, "Qxu")
    Qxu, Qxup = np.zeros((NX, NU)), np.zeros((NX, NP))
    for tm in range(2):
        for tn in range(3):
            acc = z4()
            for r in range(4):
                i, j = Q + 4 * r + 16 * tm, LI + 16 * tn
                ok = (i >= NP) & (i < NV) & (j < NX)
                acc[ok, r] = -Qxu0[j[ok], i[ok] - NP]
            for ks in range(8):
                acc = mfma16(wl[ks // 4][tm][:, ks % 4], ld[ks // 4][tn][:, ks % 4], acc)
            if tn < 2:
                for ks in range(4):
                    acc = mfma16(lam[1][tm][:, ks], ep[tn][:, ks], acc)
            for r in range(4):
                i, j = Q + 4 * r + 16 * tm, LI + 16 * tn
                for l in range(64):
                    if j[l] < NX and i[l] < NP:
                        Qxup[j[l], i[l]] = -acc[l, r]
                    elif j[l] < NX and i[l] < NV:
                        Qxu[j[l], i[l] - NP] = -acc[l, r]
    out.update(Qxu=Qxu, Qxup=Qxup)
    # ---- QU = Lam[a, :] (W Lam): symmetric in its a x a block -> stored through the mirror index; riders: Lam_a laf, Lam_a haf ----
    Quu0, lu0, hu0, lup0 = K.f(k0, "Quu"), K.f(k0, "lu"), K.f(k0, "hu"), Cd.f(c0, "lu_passive")
    Quu, Quup, lu, hu, lup = np.zeros((NU, NU)), np.zeros((NP, NU)), np.zeros(NU), np.zeros(NU), np.zeros(NP)
    for tm in range(2):
        for tn in range(2):
            acc = z4()
            for r in range(4):
                i, c = Q + 4 * r + 16 * tm, LI + 16 * tn
                ok = (i >= NP) & (i < NV) & (c >= NP) & (c < NV)
                acc[ok, r] = Quu0[c[ok] - NP, i[ok] - NP]
                s = (c == NV) & (i < NP)
                acc[s, r] = lup0[i[s]]
                s = (c == NV) & (i >= NP) & (i < NV)
                acc[s, r] = lu0[i[s] - NP]
                s = (c == NV + 1) & (i >= NP) & (i < NV)
                acc[s, r] = hu0[i[s] - NP]
            for ks in range(8):
                acc = mfma16(lam[ks // 4][tm][:, ks % 4], wl[ks // 4][tn][:, ks % 4], acc)
            for r in range(4):
                i, c = Q + 4 * r + 16 * tm, LI + 16 * tn
                for l in range(64):
                    if NP <= i[l] < NV and NP <= c[l] < NV:
                        Quu[c[l] - NP, i[l] - NP] = acc[l, r]
                    elif NP <= i[l] < NV and c[l] < NP:
                        Quup[c[l], i[l] - NP] = acc[l, r]
                    elif c[l] == NV and i[l] < NP:
                        lup[i[l]] = acc[l, r]
                    elif c[l] == NV and i[l] < NV:
                        lu[i[l] - NP] = acc[l, r]
                    elif c[l] == NV + 1 and NP <= i[l] < NV:
                        hu[i[l] - NP] = acc[l, r] * inv
    out.update(Quu=Quu, Quup=Quup, lu=lu, hu=hu, lup=lup)
    # ---- straight from the registers: Fvu through the mirror index of Lam, the expansion's vectors ----
    Fvu = np.zeros((NV, NU))
    for tr in range(2):
        for tc in range(2):
            for r in range(4):
                col, i = Q + 4 * r + 16 * tr, LI + 16 * tc
                for l in range(64):
                    if i[l] < NV and NP <= col[l] < NV:
                        Fvu[i[l], col[l] - NP] = dt * lam[tr][tc][l, r]
    LD = np.zeros((LDV, NX)); Lr = np.zeros(LDV); laf = np.zeros(LDV)
    for tr in range(2):
        for tc in range(3):
            for r in range(4):
                row, col = Q + 4 * r + 16 * tr, LI + 16 * tc
                for l in range(64):
                    if row[l] < LDV and col[l] < NX:
                        LD[row[l], col[l]] = ld[tr][tc][l, r]
                    elif row[l] < LDV and col[l] == NX:
                        Lr[row[l]] = ld[tr][tc][l, r]
                        laf[row[l]] = -xn[tr][tc][l, r]
    out.update(Fvu=Fvu, LD=LD, Lr=Lr, laf=laf)
    # ---- switching constraint: Phix -= Phia LD[a], Phiu = Phia Lam[a, u]; rider column NX: Phia Lr[a] ----
    if ns:
        Phia = np.zeros((16, 32)); Phia[:ns, :NV] = Cd.f(c0, "Phia")[:ns]
        pa = [np.array([Phia[LI[l], 4 * ks + Q[l]] for l in range(64)]) for ks in range(5)]
        Phix, Phiu = K.f(k0, "Phix").copy(), np.zeros_like(K.f(k0, "Phiu"))
        for tn in range(3):
            acc = z4()
            for ks in range(5):
                acc = mfma16(pa[ks], ld[ks // 4][tn][:, ks % 4], acc)
            for r in range(4):
                s, j = Q + 4 * r, LI + 16 * tn
                for l in range(64):
                    if s[l] < ns and j[l] < NX:
                        Phix[s[l], j[l]] -= acc[l, r]
                    elif s[l] < ns and j[l] == NX:
                        out.setdefault("PhiaLr", np.zeros(ns))[s[l]] = acc[l, r]
        for tn in range(2):
            acc = z4()
            for ks in range(5):
                acc = mfma16(pa[ks], lam[ks // 4][tn][:, ks % 4], acc)
            for r in range(4):
                s, c = Q + 4 * r, LI + 16 * tn
                for l in range(64):
                    if s[l] < ns and NP <= c[l] < NV:
                        Phiu[s[l], c[l] - NP] = acc[l, r]
        out.update(Phix=Phix, Phiu=Phiu)
    return out


def main():
    from oracle import oracle
    from robotoc_amd import problems as pr
    from robotoc_amd.types import GRID_INTERMEDIATE, Grid, Records, anymal_dims
    oracle.build()
    dims = anymal_dims()
    L = oracle.layout(dims)
    NV, NU, NF = dims.nv, dims.nu, dims.nf_max
    NP, NX = NV - NU, 2 * NV
    worst = 0.0
    for nf, ns, seed in ((12, 0, 3), (6, 6, 4), (0, 0, 5), (12, 12, 6), (6, 3, 7)):
        g = Grid(GRID_INTERMEDIATE, 0, 0, int(ns > 0), nf, ns, 3, 5, 0.013)
        gt = Grid(3, 0, 0, 0, nf, 0, 0, 6, 0.0)
        K, Cd = Records(L, "kkt"), Records(L, "cdd")
        kkt, cdd = K.zeros(2), Cd.zeros(2)
        pr.fill_precondense_instance(L, [g, gt], kkt, cdd, np.random.default_rng(seed))
        k0, c0 = kkt[0].copy(), cdd[0].copy()
        assert oracle.condense_stage(L, g, kkt[0], cdd[0]) == 0
        k1, c1 = kkt[0], cdd[0]
        Lam = Cd.f(c1, "MJtJinv")
        o = model(NV, NU, NF, nf, ns, g, k0, c0, Lam, K, Cd)
        nvf = NV + nf
        inv = 1.0 / g.num_grids_in_phase
        chk = [("Qxx", o["Qxx"], K.f(k1, "Qxx")), ("lx", o["lx"], K.f(k1, "lx")), ("hx", o["hx"], K.f(k1, "hx")),
               ("h", np.array([o["h"]]), np.array([K.f(k1, "scal")[2]])),
               ("Qxu", o["Qxu"], K.f(k1, "Qxu")), ("Qxu_passive", o["Qxup"], Cd.f(c1, "Qxu_passive")[:, :NP]),
               ("Quu", o["Quu"], K.f(k1, "Quu")), ("lu", o["lu"], K.f(k1, "lu")), ("hu", o["hu"], K.f(k1, "hu")),
               ("lu_passive", o["lup"], Cd.f(c1, "lu_passive")[:NP]),
               # (stored with leading dimension np: the record view is (8, nu), the raw order r + c np)
               ("Quu_passive_topRight", o["Quup"].T.reshape(-1), np.asarray(Cd.f(c1, "Quu_passive_topRight")).T.reshape(-1)[:NP * NU]),
               ("Fvu", o["Fvu"], K.f(k1, "Fvu")), ("MJtJinv_dIDCdqv", o["LD"][:nvf], Cd.f(c1, "MJtJinv_dIDCdqv")[:nvf]),
               ("MJtJinv_IDC", o["Lr"][:nvf], Cd.f(c1, "MJtJinv_IDC")[:nvf]), ("laf", o["laf"][:nvf], Cd.f(c1, "laf")[:nvf])]
        Fxx = K.f(k0, "Fxx").copy()
        Fxx[NV:, :] = -g.dt * o["LD"][:NV]
        Fxx[NV:, NV:] += np.eye(NV)
        chk.append(("Fxx", Fxx, K.f(k1, "Fxx")))
        chk.append(("Fx", K.f(k0, "Fx")[NV:] - g.dt * o["Lr"][:NV], K.f(k1, "Fx")[NV:]))
        if ns:
            chk += [("Phix", o["Phix"][:ns], K.f(k1, "Phix")[:ns]), ("Phiu", o["Phiu"][:ns], K.f(k1, "Phiu")[:ns]),
                    ("Pres", K.f(k0, "Pres")[:ns] - o["PhiaLr"], K.f(k1, "Pres")[:ns])]
        for name, got, want in chk:
            got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
            e = float(np.abs(got - want).max() / max(np.abs(want).max(), 1.0))
            worst = max(worst, e)
            assert e < 1e-10, (nf, ns, name, e)
        print("dimf %2d dims %2d: %d fields agree with the oracle" % (nf, ns, len(chk)))
    print("lane model of condense_rv_kernel vs oracle.condense_stage: worst relative error %.2e" % worst)


if __name__ == "__main__":
    main()
