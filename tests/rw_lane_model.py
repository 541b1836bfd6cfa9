"""Lane-level numpy model of ONE stage of riccati_backward_rw_kernel (robotoc_amd/csrc/riccati_backward_rw.hpp): the
register-resident backward Riccati step of the iCub-size shapes (nx = 64 / 70: T = 4 / 5 state tiles, one wavefront per OCP
instance, the whole 512-entry register file).  TEST INFRASTRUCTURE: tests/test_rw_lane_model.py checks it against the CPU oracle,
so that the index algebra of the kernel -- which operand of which MFMA sits in which lane, which rows of Fxx are skipped, which
lane shift a structured row costs -- is proven before it costs GPU time.

mfma16() is v_mfma_f64_16x16x4_f64 as the hardware deals its operands: a[lane] = A[m = li][k = q], b[lane] = B[k = q][n = li],
c[lane][r] = C[row = q + 4r][col = li]  (li = lane & 15, q = lane >> 4).
"""
import numpy as np

LANES = np.arange(64)
LI, Q = LANES & 15, LANES >> 4


def mfma16(a, b, c):
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    A[LI, Q] = a
    B[Q, LI] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[Q + 4 * r, LI]
    return out


def z4():
    return np.zeros((64, 4))


def lane_rot(v, n16):
    """out[lane] = v[(lane + 16 n16) & 63]  (ds_bpermute with a constant rotation: lane (li, q) reads lane (li, q + n16))."""
    return v[(LANES + 16 * n16) & 63]


def row_shr(v, n):
    """DPP row_shr:n -- lane li takes lane li - n of its 16-lane row, lanes li < n get 0."""
    out = np.zeros_like(v)
    ok = LI >= n
    out[ok] = v[LANES[ok] - n]
    return out


def row_shl(v, n):
    """DPP row_shl:n -- lane li takes lane li + n of its row, lanes li + n > 15 get 0."""
    out = np.zeros_like(v)
    ok = LI + n < 16
    out[ok] = v[LANES[ok] + n]
    return out


def qsum(v):
    """sum over the four lanes (li, 0..3): v += shfl_xor(v, 16); v += shfl_xor(v, 32)."""
    v = v + v[LANES ^ 16]
    return v + v[LANES ^ 32]


def transpose_tile(tile):
    scr = np.zeros((16, 16))
    for r in range(4):
        scr[Q + 4 * r, LI] = tile[:, r]
    out = z4()
    for r in range(4):
        out[:, r] = scr[LI, Q + 4 * r]
    return out


class Cfg:
    def __init__(self, NV, NU):
        self.NV, self.NU, self.NP, self.NX = NV, NU, NV - NU, 2 * NV
        NX = self.NX
        self.T = (NX + 15) // 16
        self.TU = (NU + 15) // 16
        self.KG = (NX + 3) // 4
        self.KGU = (NU + 3) // 4
        self.G0 = NV // 4                       # first aligned k group that meets the velocity rows
        self.G1 = (self.NP + 3) // 4            # k groups [0, G1) meet the corner rows
        self.NUC = NU - 16 * (self.TU - 1)      # lane of the rider column NU in the last control tile
        assert 0 < self.NUC < 16
        self.TS, self.LS = NV // 16, NV % 16    # column NV + k of a tile row = TS tiles and LS lanes to the right of column k
        self.DG = [g for g in range(self.KG) if g < self.G1 or g >= self.G0]   # k groups with a dense row of A

    def row_dense(self, k):
        return (k < self.NP) | ((k >= self.NV) & (k < self.NX))

    def row_struct(self, k):
        return (k >= self.NP) & (k < self.NV)


def to_tiles(P, NX):
    T = (NX + 15) // 16
    out = [[z4() for _ in range(T)] for _ in range(T)]
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
                i, j = 16 * kt + 4 * r + Q, 16 * mt + LI
                ok = (i < NX) & (j < NX)
                P[i[ok], j[ok]] = pp[kt][mt][ok, r]
    return P


def struct_rows_add(c_, dst, src, ca, cc):
    """dst[c][.] (a column of C tiles, rows = state rows) += the structured rows' part of A^T src:
         rows k in [NP, NV):       dst[k]      += ca src[k]          (same tile, register, lane)
         rows NV + k:              dst[NV + k] += cc src[k]          (TS tiles down; LS rows down: LS % 4 q-groups + LS / 4 registers)
       src[c][.]: the same column of tiles of the operand (W or PB), all T row tiles."""
    T, TS, LS = c_.T, c_.TS, c_.LS
    qs, rs = LS % 4, LS // 4
    # rotated copies: rot[c][:, r] at lane (li, q) = src[c][:, r] at lane (li, (q - qs) mod 4), i.e. row 16c + 4r + ((q - qs) mod 4)
    rot = [lane_rot(src[c], (4 - qs) % 4) for c in range(T)] if qs else src
    for c in range(T):
        for r in range(4):
            i = 16 * c + 4 * r + Q
            dst[c][:, r] += np.where(c_.row_struct(i), ca * src[c][:, r], 0.0)
            k = i - c_.NV
            # row k = 16 (c - TS) + 4 r + q - LS = 16 (c - TS) + 4 (r - rs - borrow) + ((q - qs) mod 4), borrow = (q < qs)
            e_hi = 4 * (c - TS) + r - rs          # flat register index (4 tile + r) of the source when q >= qs
            e_lo = e_hi - 1                       # ... when q < qs
            hi = rot[e_hi // 4][:, e_hi % 4] if 0 <= e_hi < 4 * T else np.zeros(64)
            lo = rot[e_lo // 4][:, e_lo % 4] if 0 <= e_lo < 4 * T else np.zeros(64)
            got = np.where(Q >= qs, hi, lo)
            dst[c][:, r] += np.where(c_.row_struct(k) & (i < c_.NX), cc * got, 0.0)


def stage(c_, pp, s_next, A, Bv, Qxx, Qxu, Quu, Fx, lx, lu, impact, direct_ht=False):
    """pp[kt][mt]: (64,4) C tiles of P+ (zero outside NX x NX); s_next: s+ [NX] (lives in LDS).
    Returns (pp_new, s_new, K [NU x NX], k [NU]); stage.mfma counts the matrix instructions."""
    NV, NU, NP, NX, T, TU, KG, KGU, G0, NUC = c_.NV, c_.NU, c_.NP, c_.NX, c_.T, c_.TU, c_.KG, c_.KGU, c_.G0, c_.NUC
    nmf = 0
    ca, cc = A[NP, NP], A[NP, NV + NP]
    # A fragment of k group g and column tile c -- the B operand of [.] A and the A operand of A^T [.] alike:
    # A[4g + q][16c + li], structured rows masked (LDS holds the dense k groups only)
    def afrag(g, c):
        k, j = 4 * g + Q, 16 * c + LI
        ok = c_.row_dense(k) & (j < NX)
        return np.where(ok, A[np.clip(k, 0, NX - 1), np.clip(j, 0, NX - 1)], 0.0)

    def row_layout(v, n):   # v[4g + q] for g = 0.. : what a lane reads from LDS
        return lambda g: np.where(4 * g + Q < n, v[np.clip(4 * g + Q, 0, n - 1)], 0.0)

    def col_layout_to_vec(tiles, n):   # tiles[c][lane] = v[16c + li] (replicated over q)
        v = np.zeros(n)
        for c in range(len(tiles)):
            j = 16 * c + LI
            ok = (j < n) & (Q == 0)
            v[j[ok]] = tiles[c][ok]
        return v

    # ---- 1. z = s+ - P+ Fx (brrf.cpp:86): per-lane partial sums over the rows a lane holds, then the q-reduction ----
    fx_row = row_layout(Fx, NX)
    zc = []
    for mt in range(T):
        part = np.zeros(64)
        for kt in range(T):
            for r in range(4):
                part += pp[kt][mt][:, r] * fx_row(4 * kt + r)
        j = 16 * mt + LI
        zc.append(np.where(j < NX, s_next[np.clip(j, 0, NX - 1)], 0.0) - qsum(part))
    z = col_layout_to_vec(zc, NX)         # -> LDS
    z_row = row_layout(z, NX)
    K = np.zeros((NU, NX))
    kvec = np.zeros(NU)
    zt = None
    if not impact:
        # ---- 2. PB = P+[:, v] Bv: acc[c][tu], rows x = 16c + .., columns u = 16 tu + li ----
        def bvfrag(g, tu):   # Bv[4g + q - NV][16 tu + li]: B operand of PB, A operand (Bv^T) of G
            k, u = 4 * g + Q - NV, 16 * tu + LI
            ok = (k >= 0) & (k < NV) & (u < NU)
            return np.where(ok, Bv[np.clip(k, 0, NV - 1), np.clip(u, 0, NU - 1)], 0.0)
        acc = [[z4() for _ in range(TU)] for _ in range(T)]
        for g in range(G0, KG):
            for tu in range(TU):
                b = bvfrag(g, tu)
                for c in range(T):
                    acc[c][tu] = mfma16(pp[g // 4][c][:, g % 4], b, acc[c][tu])
                    nmf += 1
        # ---- G = Quu + Bv^T PB[v, :]; rider column NU (lane NUC of the last control tile): Bv^T z_v ----
        gacc = [[z4() for _ in range(TU)] for _ in range(TU)]
        for tr in range(TU):
            for tu in range(TU):
                for r in range(4):
                    u0, u1 = 16 * tr + 4 * r + Q, 16 * tu + LI
                    ok = (u0 < NU) & (u1 < NU)
                    gacc[tr][tu][:, r] = np.where(ok, Quu[np.clip(u0, 0, NU - 1), np.clip(u1, 0, NU - 1)], 0.0)
        for g in range(G0, KG):
            for tu in range(TU):
                b = acc[g // 4][tu][:, g % 4]
                if tu == TU - 1:
                    b = np.where(LI == NUC, z_row(g), b)
                for tr in range(TU):
                    gacc[tr][tu] = mfma16(bvfrag(g, tr), b, gacc[tr][tu])
                    nmf += 1
        G = np.zeros((NU, NU))
        lup = np.zeros(NU)
        for tr in range(TU):
            for tu in range(TU):
                for r in range(4):
                    u0, u1 = 16 * tr + 4 * r + Q, 16 * tu + LI
                    ok = (u0 < NU) & (u1 < NU)
                    G[u0[ok], u1[ok]] = gacc[tr][tu][ok, r]
                    if tu == TU - 1:
                        okr = (u0 < NU) & (LI == NUC)
                        lup[u0[okr]] = lu[u0[okr]] - gacc[tr][tu][okr, r]     # lu' = lu - Bv^T z_v
        # ---- 3. LLT(G), Y = L^-1 (wave_llt_inv_blocked); t = Y lu', k = -Y^T t ----
        Y = np.linalg.inv(np.linalg.cholesky(G))
        tvec = Y @ lup
        kvec = -Y.T @ tvec
        if direct_ht:
            # ---- 4'. H^T = Qxu^T + PB^T A DIRECTLY (riccati_backward_rw2.hpp): the accumulators of PB are, lane for lane, the A fragments
            #      of PB^T (lane (li, q) <-> PB[4g + q][16 tu + li]); the dense k groups take A from LDS, the structured rows a SYNTHESISED
            #      B fragment (ca on the diagonal, cc NV columns to the right) -- 2 or 3 column tiles per group, no transposes, no lane
            #      rotations; z in the idle column NU of PB makes row NU of H^T the rider A^T z ----
            for g in range(KG):
                accz = acc[g // 4][TU - 1][:, g % 4]
                acc[g // 4][TU - 1][:, g % 4] = np.where(LI == NUC, z_row(g), accz)
            hT = [[z4() for _ in range(T)] for _ in range(TU)]
            for tu in range(TU):
                for c in range(T):
                    for r in range(4):
                        u, x = 16 * tu + 4 * r + Q, 16 * c + LI
                        ok = (u < NU) & (x < NX)
                        hT[tu][c][:, r] = np.where(ok, Qxu[np.clip(x, 0, NX - 1), np.clip(u, 0, NU - 1)], 0.0)
            for g in range(KG):
                k = 4 * g + Q
                for c in range(T):
                    x = 16 * c + LI
                    if g in c_.DG:
                        b = afrag(g, c)
                    else:
                        b = np.zeros(64)
                    # the structured rows of this group (also those inside a dense group: afrag masks them)
                    bs = np.where(c_.row_struct(k) & (x == k), ca, 0.0) + np.where(c_.row_struct(k) & (x == NV + k) & (x < NX), cc, 0.0)
                    if not (np.any(b != 0) or np.any(bs != 0)) and g not in c_.DG:
                        continue
                    for tu in range(TU):
                        if g in c_.DG:
                            hT[tu][c] = mfma16(acc[g // 4][tu][:, g % 4], b, hT[tu][c])
                            nmf += 1
                        if np.any(bs != 0):
                            hT[tu][c] = mfma16(acc[g // 4][tu][:, g % 4], bs, hT[tu][c])
                            nmf += 1
            w0 = np.zeros(NX)
            rr_, qq_ = (NU - 16 * (TU - 1)) // 4, (NU - 16 * (TU - 1)) % 4
            for c in range(T):
                x = 16 * c + LI
                ok = (x < NX) & (Q == qq_)
                w0[x[ok]] = hT[TU - 1][c][ok, rr_]
            hT_direct, w0_direct = hT, w0
        # ---- 4. H = A^T PB (rows x, columns u) with z in the idle column NU of PB: column NU of H is A^T z ----
        for g in range(KG):
            accz = acc[g // 4][TU - 1][:, g % 4]
            acc[g // 4][TU - 1][:, g % 4] = np.where(LI == NUC, z_row(g), accz)
        hx = [[z4() for _ in range(TU)] for _ in range(T)]
        for g in c_.DG:
            for c in range(T):
                a = afrag(g, c)
                for tu in range(TU):
                    hx[c][tu] = mfma16(a, acc[g // 4][tu][:, g % 4], hx[c][tu])
                    nmf += 1
        for tu in range(TU):
            col = [hx[c][tu] for c in range(T)]
            struct_rows_add(c_, col, [acc[c][tu] for c in range(T)], ca, cc)
        # the rider: w0 = A^T z (rows x on lanes li == NUC of the last control tile) -> LDS
        w0 = np.zeros(NX)
        for c in range(T):
            for r in range(4):
                i = 16 * c + 4 * r + Q
                ok = (i < NX) & (LI == NUC)
                w0[i[ok]] = hx[c][TU - 1][ok, r]
        # ---- 5. H^T = transpose(H) + Qxu^T: hT[tu][c], rows u, columns x (li along the contiguous index of Qxu) ----
        hT = [[None] * T for _ in range(TU)]
        for tu in range(TU):
            for c in range(T):
                tr_ = transpose_tile(hx[c][tu])
                for r in range(4):
                    u, x = 16 * tu + 4 * r + Q, 16 * c + LI
                    ok = (u < NU) & (x < NX)
                    tr_[:, r] = np.where(ok, tr_[:, r] + Qxu[np.clip(x, 0, NX - 1), np.clip(u, 0, NU - 1)], 0.0)
                hT[tu][c] = tr_
        if direct_ht:   # (the transposed path above ran on the same operands: the two must agree; the direct one is used)
            for tu in range(TU):
                for c in range(T):
                    for r in range(4):
                        u, x = 16 * tu + 4 * r + Q, 16 * c + LI
                        ok = (u < NU) & (x < NX)
                        assert np.abs(np.where(ok, hT_direct[tu][c][:, r] - hT[tu][c][:, r], 0.0)).max() <= 1e-9 * max(1.0, np.abs(hT[tu][c]).max())
                        hT_direct[tu][c][:, r] = np.where(ok, hT_direct[tu][c][:, r], 0.0)   # (row NU carried the rider)
            assert np.abs(w0_direct - w0).max() <= 1e-9 * max(1.0, np.abs(w0).max())
            hT, w0 = hT_direct, w0_direct
        # ---- 6. Z^T = Y H^T (Y lower triangular: the tile above the diagonal is skipped) ----
        def yfrag(tu, gj):      # A operand Y[m = 16 tu + li][k = 4 gj + q]
            m, k = 16 * tu + LI, 4 * gj + Q
            ok = (m < NU) & (k < NU)
            return np.where(ok, Y[np.clip(m, 0, NU - 1), np.clip(k, 0, NU - 1)], 0.0)

        def ytfrag(tu, gj):     # A operand -Y^T[m = 16 tu + li][k = 4 gj + q] = -Y[k][m]
            m, k = 16 * tu + LI, 4 * gj + Q
            ok = (m < NU) & (k < NU)
            return np.where(ok, -Y[np.clip(k, 0, NU - 1), np.clip(m, 0, NU - 1)], 0.0)
        # riccati_backward_rw2.hpp: lu' rides in the idle column NX of H^T's last column tile -> t = Y lu' in Z^T, k = -Y^T t in K
        RID = NX % 16
        riders = direct_ht and RID != 0
        if riders:
            for tu in range(TU):
                for r in range(4):
                    u = 16 * tu + 4 * r + Q
                    sel = (LI == RID) & (u < NU)
                    hT[tu][T - 1][:, r] = np.where(sel, lup[np.clip(u, 0, NU - 1)], hT[tu][T - 1][:, r])
        zt = [[z4() for _ in range(T)] for _ in range(TU)]
        for tu in range(TU):
            for gj in range(min(KGU, 4 * (tu + 1))):
                a = yfrag(tu, gj)
                for c in range(T):
                    zt[tu][c] = mfma16(a, hT[gj // 4][c][:, gj % 4], zt[tu][c])
                    nmf += 1
        # ---- 7. K = -Y^T Z^T (Y^T upper triangular), tile by tile -> HBM ----
        for c in range(T):
            for tu in range(TU):
                kk = z4()
                for gj in range(4 * tu, KGU):
                    kk = mfma16(ytfrag(tu, gj), zt[gj // 4][c][:, gj % 4], kk)
                    nmf += 1
                for r in range(4):
                    u, x = 16 * tu + 4 * r + Q, 16 * c + LI
                    ok = (u < NU) & (x < NX)
                    K[u[ok], x[ok]] = kk[ok, r]
                    if riders and c == T - 1:   # the riders against the vector forms of step 3
                        sel = (LI == RID) & (u < NU)
                        assert np.abs(kk[sel, r] - kvec[u[sel]]).max() <= 1e-9 * max(1.0, np.abs(kvec).max())
                        assert np.abs(zt[tu][c][sel, r] - tvec[u[sel]]).max() <= 1e-9 * max(1.0, np.abs(tvec).max())
    # ---- 8. F starts from Qxx (upper tiles, off-diagonal ones symmetrised: brrf.cpp:85 folded into the start value), F -= Z Z^T ----
    f = [[None] * T for _ in range(T)]
    for c in range(T):
        for t in range(c, T):
            d = z4()
            for r in range(4):
                i, j = 16 * c + 4 * r + Q, 16 * t + LI
                ok = (i < NX) & (j < NX)
                ic, jc = np.clip(i, 0, NX - 1), np.clip(j, 0, NX - 1)
                v = Qxx[jc, ic] if t == c else 0.5 * (Qxx[ic, jc] + Qxx[jc, ic])
                d[:, r] = np.where(ok, v, 0.0)
            f[c][t] = d
    if not impact:
        for gu in range(KGU):
            for c in range(T):
                for t in range(c, T):
                    f[c][t] = mfma16(-zt[gu // 4][c][:, gu % 4], zt[gu // 4][t][:, gu % 4], f[c][t])
                    nmf += 1
    # ---- s = A^T z - lx - H k = w0 - lx + Z t: column layout, per-lane partial sums + q-reduction ----
    if impact:
        # no H product on an impact grid point: A^T z as a mat-vec from the (dense-row) LDS copy of A plus the structured rows
        w0 = A.T @ z
    s_new = w0 - lx
    if not impact:
        t_row = row_layout(tvec, NU)
        for c in range(T):
            part = np.zeros(64)
            for tu in range(TU):
                for r in range(4):
                    part += zt[tu][c][:, r] * t_row(4 * tu + r)
            part = qsum(part)
            j = 16 * c + LI
            ok = (j < NX) & (Q == 0)
            s_new[j[ok]] += part[ok]
    # ---- 9. column tile by column tile: W[:, t] = P+ A[:, t], F[c][t] += A^T[c] W[:, t] ----
    for t in range(T):
        w = [z4() for _ in range(T)]
        for g in c_.DG:
            b = afrag(g, t)
            for tm in range(T):
                w[tm] = mfma16(pp[g // 4][tm][:, g % 4], b, w[tm])
                nmf += 1
        # structured rows k of A: W[:, k] += ca P+[:, k] (same tile / lane), W[:, NV + k] += cc P+[:, k] (TS tiles, LS lanes to the left)
        j = 16 * t + LI
        for tm in range(T):
            for r in range(4):
                w[tm][:, r] += np.where(c_.row_struct(j), ca * pp[tm][t][:, r], 0.0)
                k = j - NV
                src = np.zeros(64)
                if 0 <= t - c_.TS < T:
                    src = src + (row_shr(pp[tm][t - c_.TS][:, r], c_.LS) if c_.LS else pp[tm][t - c_.TS][:, r])
                if c_.LS and 0 <= t - c_.TS - 1 < T:
                    src = src + row_shl(pp[tm][t - c_.TS - 1][:, r], 16 - c_.LS)
                w[tm][:, r] += np.where(c_.row_struct(k) & (j < NX), cc * src, 0.0)
        for g in c_.DG:
            for c in range(t + 1):
                f[c][t] = mfma16(afrag(g, c), w[g // 4][:, g % 4], f[c][t])
                nmf += 1
        col = [f[c][t] if c <= t else z4() for c in range(T)]
        struct_rows_add(c_, col, w, ca, cc)
        for c in range(t + 1):
            f[c][t] = col[c]
    # ---- 10. P = sym(F): upper tiles as computed, diagonal tiles mirrored, lower tiles transposed ----
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
                tr_ = transpose_tile(f[c][c])
                d = z4()
                for r in range(4):
                    d[:, r] = np.where(Q + 4 * r <= LI, f[c][c][:, r], tr_[:, r])
                pn[c][c] = mask(d, c, c)
            else:
                pn[c][t] = mask(f[c][t], c, t)
                pn[t][c] = mask(transpose_tile(mask(f[c][t], c, t)), t, c)
    stage.mfma = nmf
    return pn, s_new, K, kvec
