"""Lane-level numpy restatement of disn_amd/csrc/conv_h2w.hip (test infrastructure, CPU).

Follows the batched convolution kernel's index arithmetic one to one: tile decode, halo loader units (with the
repeat-the-last-unit clamp), byte-addressed LDS with the padded pixel / padded row layout, the window-major
row <-> pixel map, tap shifts as byte offsets, k-waves and their hand-over, the C layout of
v_mfma_f32_32x32x16_f16, the epilogue's quad -> window map and the in-register 2x2 pool.  An indexing mistake
shows here, before a GPU minute is spent.  Numerics as in conv_h2_emulation.py: numpy float16 split, products and
accumulation in float64 (the MFMA's fp32 accumulation rounding is not modelled).
"""
import numpy as np

from conv_h2_emulation import pack, pow2_scale, quad_row, sigma, split  # noqa: F401  (same weight image, same sigma)

# variant -> MB, MWV, NWV, WK, TH, TW   (conv_h2w_launch's switch)
VARIANTS = {1: (4, 2, 2, 1, 8, 32), 2: (7, 1, 4, 1, 8, 28), 3: (7, 1, 2, 1, 8, 28), 4: (7, 1, 4, 2, 8, 28),
            5: (7, 1, 2, 2, 8, 28), 6: (7, 1, 4, 1, 8, 28), 7: (7, 1, 2, 2, 8, 28)}
# round 6, the SEGMENTED variants (SEG = 2): the k16 blocks in two halves (lower -> p0, upper -> p1), each half in segments
# of two chunks whose sums are added in order, p0 + p1 at the end.  6 (PARK): ONE k-wave walks every block in order and
# parks p0 at the midpoint; 7 (HALVES): k-wave 0 walks the lower half, k-wave 1 the upper half -- chunk i holds blocks i and
# Cin / 32 + i.  Same (block, tap) sequence per accumulator and the same segment boundaries: the same bits on the GPU.
SEGMENTED = {6: "park", 7: "halves"}


def segment_plan(variant, cin):
    """-> [p0, p1], each a list of segments, each segment the k16 blocks it sums (in order) -- per accumulator"""
    kb = cin // 16
    assert variant in SEGMENTED and kb % 4 == 0 and kb >= 8, "two halves of whole two-chunk segments"
    halves = [list(range(0, kb // 2)), list(range(kb // 2, kb))]
    return [[h[i:i + 2] for i in range(0, len(h), 2)] for h in halves]


def chunk_blocks(variant, cin, c):
    """k16 blocks that chunk c of the walk stages in LDS, per k-wave: {wk: block}"""
    kb = cin // 16
    if SEGMENTED.get(variant) == "halves":
        return {0: c, 1: kb // 2 + c}
    WK = VARIANTS[variant][3]
    return {wk: WK * c + wk for wk in range(WK)}


_r, _lane = np.meshgrid(np.arange(16), np.arange(64), indexing="ij")
C_ROW = (_r & 3) + 8 * (_r >> 2) + 4 * (_lane >> 5)   # [r][lane] -> row of the 32 x 32 block
C_COL = _lane & 31


def row_bytes(raw):
    r = (raw // 256) * 256 + 128
    return r if r >= raw else r + 256


def geometry(variant):
    MB, MWV, NWV, WK, TH, TW = VARIANTS[variant]
    CK = 16 * WK
    KPIX = CK * 4 + 16
    RP, HR = TW + 2, TH + 2
    ROWB = row_bytes(RP * KPIX)
    return dict(MB=MB, MWV=MWV, NWV=NWV, WK=WK, TH=TH, TW=TW, CK=CK, KPIX=KPIX, UPP=CK // 4, RP=RP, HR=HR, ROWB=ROWB,
                BUF=HR * ROWB, NT=64 * MWV * NWV * WK, UNITS=HR * RP * (CK // 4), WPR=TW // 2)


def kwaves(H, W, cout):
    return 1 if H * W * cout >= 56 * 56 * 256 else 2


def arow(G, lane, wm, wk, mb):
    """byte offset of the top-left tap's h fragment of this lane's A row of block mb"""
    i, g = lane & 31, lane >> 5
    m = 32 * (wm * G["MB"] + mb) + sigma(i)
    w, e = m >> 2, m & 3
    wy, wx = divmod(w, G["WPR"])
    return (2 * wy + (e >> 1)) * G["ROWB"] + (2 * wx + (e & 1)) * G["KPIX"] + (16 * wk + 8 * g) * 2


def conv_tile(x, img, s_w, bias, variant, tile, relu=True):
    """one workgroup: x [H,W,Cin] fp32, tile = (tyi, txi, nt) -> dict (y, x, n) -> value, pooled dict, vmax"""
    G = geometry(variant)
    MB, MWV, NWV, WK, TH, TW = (G[k] for k in ("MB", "MWV", "NWV", "WK", "TH", "TW"))
    CK, KPIX, UPP, RP, ROWB, BUF, NT, UNITS, WPR = (G[k] for k in ("CK", "KPIX", "UPP", "RP", "ROWB", "BUF", "NT", "UNITS", "WPR"))
    H, W, Cin = x.shape
    KB, NC = Cin // 16, Cin // CK
    LP = (UNITS + NT - 1) // NT
    tyi, txi, nt = tile
    y0, x0 = tyi * TH, txi * TW
    sa = pow2_scale(np.abs(x).max(), 14)
    descale = np.float32(1.0) / sa * (np.float32(1.0) / np.asarray(s_w, np.float32))     # [Cout]: per column
    acc = np.zeros((MWV, NWV, WK, MB, 16, 64), np.float64)
    visited = {}
    for c in range(NC):
        # ---- the halo in byte-addressed LDS (one buffer; f16 element a at byte 2a), NaN = never written ----
        lds = np.full(BUF // 2, np.nan, np.float64)
        written = np.zeros(BUF // 2, bool)
        for tid in range(NT):
            for k in range(LP):
                u = min(tid + k * NT, UNITS - 1)
                hp, c4 = divmod(u, UPP)
                hy, hx = divmod(hp, RP)
                yy, xx = y0 - 1 + hy, x0 - 1 + hx
                ok = 0 <= yy < H and 0 <= xx < W
                if SEGMENTED.get(variant) == "halves":   # two 64-byte pieces of two cache lines per halo pixel
                    ch0 = 16 * c + 4 * c4 if c4 < 4 else Cin // 2 + 16 * c + 4 * (c4 - 4)
                else:
                    ch0 = CK * c + 4 * c4
                v = (x[yy, xx, ch0:ch0 + 4] * sa).astype(np.float32) if ok else np.zeros(4, np.float32)
                h, l = split(v)
                woff = hy * ROWB + hx * KPIX + 8 * c4
                for e in range(4):
                    for plane, val in ((0, h[e]), (CK * 2, l[e])):
                        a = (woff + plane) // 2 + e
                        assert not written[a] or lds[a] == float(val), "two units store different values at one address"
                        lds[a] = float(val)
                        written[a] = True
        for wm in range(MWV):
            for wn in range(NWV):
                n0 = (nt * NWV + wn) * 32
                for wk in range(WK):
                    for t in range(9):
                        blk = chunk_blocks(variant, Cin, c)[wk]
                        visited.setdefault((wm, wn, wk), []).append(blk) if t == 0 else None
                        f = ((n0 >> 5) * KB + blk) * 9 + t
                        bh, bl = img[f, 0].astype(np.float64), img[f, 1].astype(np.float64)
                        Bh = np.zeros((16, 32)); Bl = np.zeros((16, 32))
                        for lane in range(64):
                            j, g = lane & 31, lane >> 5
                            Bh[8 * g:8 * g + 8, j] = bh[lane]
                            Bl[8 * g:8 * g + 8, j] = bl[lane]
                        off = (t // 3) * ROWB + (t % 3) * KPIX
                        for mb in range(MB):
                            Ah = np.zeros((32, 16)); Al = np.zeros((32, 16))
                            for lane in range(64):
                                i, g = lane & 31, lane >> 5
                                a = arow(G, lane, wm, wk, mb) + off
                                Ah[i, 8 * g:8 * g + 8] = lds[a // 2:a // 2 + 8]
                                Al[i, 8 * g:8 * g + 8] = lds[(a + CK * 2) // 2:(a + CK * 2) // 2 + 8]
                            assert not np.isnan(Ah).any() and not np.isnan(Al).any(), "a fragment read LDS bytes nobody wrote"
                            Dm = Al @ Bh + Ah @ Bl + Ah @ Bh
                            acc[wm, wn, wk, mb] += Dm[C_ROW, C_COL]      # C layout: lane (j, g), register r
    if variant in SEGMENTED:   # every accumulator's walk is the plan's: PARK one k-wave (p0 then p1), HALVES k-wave w = half w
        plan = segment_plan(variant, Cin)
        for (wm, wn, wk), blocks in visited.items():
            want = [b for seg in (plan[0] + plan[1] if SEGMENTED[variant] == "park" else plan[wk]) for b in seg]
            assert blocks == want, (variant, wk, blocks, want)
    out, pooled, vmax = {}, {}, 0.0
    MBH = (MB + 1) // 2 if WK == 2 else MB
    for wm in range(MWV):
        for wn in range(NWV):
            n0 = (nt * NWV + wn) * 32
            tot = acc[wm, wn, 0] + acc[wm, wn, 1] if WK == 2 else acc[wm, wn, 0]
            for wk in range(WK):
                lo, hi = (MBH, MB) if (WK == 2 and wk == 1) else (0, MBH)
                for mb in range(lo, hi):
                    for lane in range(64):
                        j, g = lane & 31, lane >> 5
                        for q in range(4):
                            m = 32 * (wm * MB + mb) + quad_row(q, g)
                            wy, wx = divmod(m >> 2, WPR)
                            yy, xx = y0 + 2 * wy, x0 + 2 * wx
                            v = []
                            for e in range(4):
                                t = tot[mb, 4 * q + e, lane] * float(descale[n0 + j]) + float(bias[n0 + j])
                                if relu:
                                    t = max(t, 0.0)
                                v.append(t)
                                py, px = yy + (e >> 1), xx + (e & 1)
                                if py < H and px < W:
                                    assert (py, px, n0 + j) not in out, "two lanes store the same element"
                                    out[(py, px, n0 + j)] = t
                                    vmax = max(vmax, abs(t))
                            if yy + 1 < H and xx + 1 < W:
                                assert (yy >> 1, xx >> 1, n0 + j) not in pooled
                                pooled[(yy >> 1, xx >> 1, n0 + j)] = max(v)
    return out, pooled, vmax


def conv(x, w_hwio, bias, variant, relu=True):
    """whole layer, every workgroup -> (out [H,W,Cout], pooled [H/2,W/2,Cout] (nan where not written), vmax)"""
    G = geometry(variant)
    H, W, _ = x.shape
    cout = w_hwio.shape[-1]
    img, s_w = pack(w_hwio)
    out = np.full((H, W, cout), np.nan)
    pooled = np.full((H // 2, W // 2, cout), np.nan)
    vmax = 0.0
    for tyi in range((H + G["TH"] - 1) // G["TH"]):
        for txi in range((W + G["TW"] - 1) // G["TW"]):
            for nt in range(cout // (32 * G["NWV"])):
                o, p, m = conv_tile(x, img, s_w, bias, variant, (tyi, txi, nt), relu)
                for k, v in o.items():
                    assert np.isnan(out[k]), "two workgroups store the same element"
                    out[k] = v
                for k, v in p.items():
                    assert np.isnan(pooled[k])
                    pooled[k] = v
                vmax = max(vmax, m)
    return out, pooled, vmax


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in grp] for grp in B128_GROUPS]


def lds_read_conflicts(variant):
    """(worst, total extra cycles, reads): extra LDS cycles of the A-fragment ds_read_b128s of one chunk
    (MI355X_MICROARCH.md, LDS: four 16-lane groups per instruction, a 16-byte access covers four of the 64 banks)"""
    G = geometry(variant)
    worst = total = reads = 0
    for wm in range(G["MWV"]):
        for wk in range(G["WK"]):
            for mb in range(G["MB"]):
                for t in range(9):
                    for plane in (0, G["CK"] * 2):
                        for grp in B128_GROUPS:
                            slots = {}
                            for lane in grp:
                                a = arow(G, lane, wm, wk, mb) + (t // 3) * G["ROWB"] + (t % 3) * G["KPIX"] + plane
                                assert a % 16 == 0
                                slots.setdefault((a // 16) % 16, set()).add(a)
                            extra = max(len(v) for v in slots.values()) - 1
                            worst = max(worst, extra)
                            total += extra
                            reads += 1
    return worst, total, reads
