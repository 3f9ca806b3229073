"""CPU emulation of the data movement of csrc/ampb_f16x3.hip (one workgroup): the permuted A rows that make the MFMA result land in
the P layout, the halo exchange (v_permlane32_swap + LDS slots), the 8 x 8 DPP lane transposes and the operand-tile addresses, and the conv's fragment reads -- every
index formula of the kernel restated over numpy "registers" [wave][lane][...] and checked against the tensor coordinates it is meant
to hold.  The hardware semantics it assumes (permlane32_swap, DPP row_shl/shr/quad_perm, the MFMA operand mirror) are the ones
tests/experiments/ampb_primitives.hip checks on the GPU.      python tests/experiments/ampb_layout_emulation.py"""
import numpy as np

WM, WN, G = 2, 4, 32
W, WL = 128 * WN, 128 * WN + 2 * G + 1          # + 1 pad column: WL = 1 (mod 8), see the bank check of the scatter below
NCH, CHS = 2 * WM, 4 * WL
NW = WM * WN


def f(ch, col):            # the tensor value at (channel, tile column): unique per coordinate
    return ch * 10000.0 + col


PT = lambda c: c >> 4
PR = lambda c: c & 15


def arow_col(i, t):          # A row i of MFMA tile t reads wave column ...
    return 64 * ((i >> 2) & 1) + 16 * t + 4 * (i >> 3) + (i & 3)


# --- conv output (transposed product): lane (m, h), acc[t][r] = MFMA row 8 (r >> 2) + 4 h + (r & 3), i.e. the column that row read
acc = np.zeros((NW, 64, 4, 16))
for w in range(NW):
    wm, wn = divmod(w, WN)
    for lane in range(64):
        m, h = lane & 31, lane >> 5
        for t in range(4):
            for r in range(16):
                acc[w, lane, t, r] = f(32 * wm + m, 128 * wn + arow_col(8 * (r >> 2) + 4 * h + (r & 3), t))


def permlane32_swap(a, b):  # a, b: [64]; lanes 32-63 of a swap with lanes 0-31 of b
    a2, b2 = a.copy(), b.copy()
    a2[32:], b2[:32] = b[:32], a[32:]
    return a2, b2


# --- the result is the P layout
for w in range(NW):
    wm, wn = divmod(w, WN)
    for lane in range(64):
        m, h = lane & 31, lane >> 5
        for c in range(64):
            assert acc[w, lane, PT(c), PR(c)] == f(32 * wm + m, 128 * wn + 64 * h + c), (w, lane, c)
print("P layout ok")

# --- halo exchange
xch = np.zeros(NW * 320)
for w in range(NW):
    for lane in range(64):
        m, h = lane & 31, lane >> 5
        for i in range(5):
            xch[((w * 2 + h) * 32 + m) * 5 + i] = acc[w, lane, PT(59 + i), PR(59 + i)] if h else acc[w, lane, PT(i), PR(i)]
for w in range(NW):
    wm, wn = divmod(w, WN)
    wl, wr = (w - 1 if wn > 0 else w), (w + 1 if wn + 1 < WN else w)
    for i in range(5):
        s0, s1 = permlane32_swap(acc[w, :, PT(i), PR(i)], acc[w, :, PT(59 + i), PR(59 + i)])
        for lane in range(64):
            m, h = lane & 31, lane >> 5
            nb = xch[((wr * 2 + 0) * 32 + m) * 5 + i] if h else xch[((wl * 2 + 1) * 32 + m) * 5 + i]
            hl = s0[lane] if h else nb
            hr = nb if h else s1[lane]
            base = 128 * wn + 64 * h
            if base - 5 + i >= 0 and (h or wn > 0):
                assert hl == f(32 * wm + m, base - 5 + i), ("hl", w, lane, i)
            if base + 64 + i < W and (not h or wn + 1 < WN):
                assert hr == f(32 * wm + m, base + 64 + i), ("hr", w, lane, i)
print("halo exchange ok")


# --- transposes + operand-tile writes
def dpp(v, kind):           # v: [64]
    out = np.zeros(64)
    for l in range(64):
        row, i = l & ~15, l & 15
        if kind == "shr4":
            out[l] = v[row + i - 4] if i >= 4 else 0.0
        elif kind == "shl4":
            out[l] = v[row + i + 4] if i + 4 < 16 else 0.0
        elif kind == "x1":
            out[l] = v[l ^ 1]
        elif kind == "x2":
            out[l] = v[l ^ 2]
    return out


lanes = np.arange(64)
b4, b2, b1 = (lanes & 4) != 0, (lanes & 2) != 0, (lanes & 1) != 0


def transpose8(R):          # R: [8][64]
    for j in range(4):
        lo, hi = R[j].copy(), R[j + 4].copy()
        R[j] = np.where(b4, dpp(hi, "shr4"), lo)
        R[j + 4] = np.where(b4, hi, dpp(lo, "shl4"))
    for jj in range(4):
        j = (jj & 1) + 4 * (jj >> 1)
        lo, hi = R[j].copy(), R[j + 2].copy()
        R[j] = np.where(b2, dpp(hi, "x2"), lo)
        R[j + 2] = np.where(b2, hi, dpp(lo, "x2"))
    for jj in range(4):
        j = 2 * jj
        lo, hi = R[j].copy(), R[j + 1].copy()
        R[j] = np.where(b1, dpp(hi, "x1"), lo)
        R[j + 1] = np.where(b1, hi, dpp(lo, "x1"))
    return R


smem = np.full((NCH * CHS, 8), -1.0)     # uint4 units x 8 f16 (hi plane values; the lo plane sits 2 * WL further)
for w in range(NW):
    wm, wn = divmod(w, WN)
    for b in range(8):
        R = [acc[w, :, PT(8 * b + j), PR(8 * b + j)].copy() for j in range(8)]
        R = transpose8(R)
        for lane in range(64):
            h, e8, o4 = lane >> 5, lane & 7, (lane >> 3) & 3
            colb = G + 128 * wn + 64 * h + e8
            dst = (2 * wm + (o4 >> 1)) * CHS + (o4 & 1) * WL + colb
            smem[dst + 8 * b] = [R[j][lane] for j in range(8)]
            smem[dst + 8 * b + 2 * WL] = [-R[j][lane] for j in range(8)]      # "lo plane": tagged by sign
for c in range(NCH):
    for oct_ in range(2):
        for col in range(W):
            for e in range(8):
                assert smem[c * CHS + oct_ * WL + G + col, e] == f(16 * c + 8 * oct_ + e, col), (c, oct_, col, e)
                assert smem[c * CHS + 2 * WL + oct_ * WL + G + col, e] == -f(16 * c + 8 * oct_ + e, col)
print("operand tile ok (register transposes: AMP_AMPB_SCATTER = 0)")

# --- the default form: every lane scatters its own 64 columns as 2-byte stores (ds_write_b16), no lane transposes
smem2 = np.full((NCH * CHS, 8), -1.0)
worst = 0
for w in range(NW):
    wm, wn = divmod(w, WN)
    for c in range(64):
        banks = {}
        for lane in range(64):
            h, e8, o4 = lane >> 5, lane & 7, (lane >> 3) & 3
            unit = (2 * wm + (o4 >> 1)) * CHS + (o4 & 1) * WL + G + 128 * wn + 64 * h + c
            smem2[unit, e8] = acc[w, lane, PT(c), PR(c)]
            smem2[unit + 2 * WL, e8] = -acc[w, lane, PT(c), PR(c)]
            byte = unit * 16 + 2 * e8
            banks.setdefault((h, (byte // 4) % 32), set()).add(byte // 4)     # per half-wave: distinct dwords on one bank
        worst = max(worst, max(len(v) for v in banks.values()))
assert np.array_equal(smem2, smem)
print(f"operand tile ok (LDS scatter); distinct dwords per bank and half-wave: {worst} (1 = conflict-free; two lanes share each dword)")

# --- the conv's fragment reads (A operand of the transposed product: row = time = lane & 31, k-block = lane >> 5)
H2, d = 5, 5
for w in range(NW):
    wm, wn = divmod(w, WN)
    for lane in range(64):
        m, h = lane & 31, lane >> 5
        rd = h * WL + G + 128 * wn + 64 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3) - H2 * d
        for c in range(NCH):
            for g in (0, 5, 10):
                for t in range(4):
                    col = 128 * wn + arow_col(m, t) + (g - H2) * d
                    got = smem[c * CHS + rd + g * d + 16 * t]
                    if 0 <= col < W:
                        assert all(got[e] == f(16 * c + 8 * h + e, col) for e in range(8)), (w, lane, c, g, t)
                    else:
                        assert all(got[e] == -1.0 for e in range(8))           # guard columns (zero in the kernel)
print("fragment reads ok")
# ds_read_b128 is served in four 16-lane groups; each must touch 16 distinct 16-B slots (mod 16: 64 banks)
for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
    assert len({arow_col(i, 0) % 16 for i in grp}) == 16
print("fragment reads conflict-free")
# --- the edge patch address: channel ch, tile column col -> f16 index
for ch in (0, 7, 8, 15, 16, 40, 63):
    for col in (0, 3, 100):
        rowoff = (((ch >> 4) * CHS + ((ch >> 3) & 1) * WL + G) << 3) + (ch & 7)
        idx = rowoff + (col << 3)
        assert smem.reshape(-1)[idx] == f(ch, col), (ch, col)
        assert smem.reshape(-1)[rowoff + ((col + 2 * WL) << 3)] == -f(ch, col)
print("edge patch address ok")
print("ALL OK")
