"""Lane-level numpy restatement of disn_amd/csrc/dense_h2.hip (test infrastructure, CPU) -- the 1x1 sibling of
tests/conv_h2_emulation.py.  It follows the kernel's index arithmetic one to one: the weight image (the conv_h2
image with one tap), the row loader's unit -> (row, float4) map and LDS addresses, the hardware row <-> logical row
map sigma of the A fragments, the four k-waves' k16 blocks, the C layout of v_mfma_f32_32x32x16_f16, the
(w0 + w2) + (w1 + w3) reduction and the epilogue's row map; two sources read in place, the deferred bias + ReLU on
load, per-image maxima.  The two-term split is numpy float16; products and sums are float64 (the MFMA's fp32
accumulation is not modelled).

`presplit=True` restates the NEXT step of DESIGN.md section 6 as an executable specification: the A operand arrives
already split -- per 4-channel unit the 16 bytes hold h0..h3 | l0..l3 (float16) instead of four floats -- with the
power-of-two scale its producer chose from a rigorous bound, so the loader is a plain copy (no VALU), and a second
source with another scale is folded in by rescaling the accumulators exactly between the two K ranges."""
import numpy as np

from conv_h2_emulation import pow2_scale, sigma, split

WK = 4
TILES = {(1, 2): "<1,2,KPW>", (2, 2): "<2,2,KPW>", (2, 4): "<2,4,KPW>", (4, 2): "<4,2,KPW>"}


def pack(w_kn):
    """[K][N] -> (image [frags][plane][lane][8] float16, s_w [N]): frag index nb * (K / 16) + kb (one tap); one
    power-of-two scale per output column"""
    K, N = w_kn.shape
    s = np.array([pow2_scale(m, 13) for m in np.abs(w_kn).max(axis=0)], np.float32)
    KB = K // 16
    img = np.zeros(((N // 32) * KB, 2, 64, 8), np.float16)
    for f in range(img.shape[0]):
        kb, nb = f % KB, f // KB
        for lane in range(64):
            j, g = lane & 31, lane >> 5
            v = (w_kn[16 * kb + 8 * g + np.arange(8), 32 * nb + j] * s[32 * nb + j]).astype(np.float32)
            img[f, 0, lane], img[f, 1, lane] = split(v)
    return img, s


def to_split_form(x, s):
    """fp32 [M][K] -> the split-form tensor (same bytes: per 4 channels h0..h3 | l0..l3 as float16), scale s"""
    h, l = split((x * np.float32(s)).astype(np.float32))
    M, K = x.shape
    out = np.zeros((M, K // 4, 8), np.float16)
    out[:, :, :4] = h.reshape(M, K // 4, 4)
    out[:, :, 4:] = l.reshape(M, K // 4, 4)
    return out.reshape(M, 2 * K)


def dense_tile(a1, a2, img, s_w, bias, MB, NW, KPW, tile, amax, in_bias=None, relu=True, presplit=None):
    """one workgroup.  a1 [M][k1], a2 [M][k2] or None; tile = (mt, nt); amax = (max|a1|, max|a2|) of the tile's
    image.  presplit = (s1, s2): a1 / a2 are split-form tensors with these scales.  -> {(m, n): value}, vmax"""
    CK, BM = 16 * WK * KPW, 32 * MB
    KPIX, UPP = CK * 4 + 16, (16 * WK * KPW) // 4
    NT = 256 * NW
    M = a1.shape[0]
    k1 = a1.shape[1] // (2 if presplit else 1)
    k2 = 0 if a2 is None else a2.shape[1] // (2 if presplit else 1)
    K = k1 + k2
    KB = K // 16
    mt, nt = tile
    m0 = mt * BM
    assert (BM * UPP) % NT == 0 and K % CK == 0 and k1 % CK == 0
    LP = BM * UPP // NT
    bmax = 0.0 if in_bias is None else float(np.abs(in_bias).max())
    sa = None if presplit else pow2_scale(np.float32(max(amax)) + np.float32(bmax), 14)
    out, vmax = {}, 0.0
    for wn in range(NW):
        n0 = (nt * NW + wn) * 32
        acc = np.zeros((WK, MB, 16, 64), np.float64)
        cur_scale = None
        for c in range(K // CK):
            # ---- row loader: LDS image of the chunk, byte-addressed like the kernel (h plane, l plane per row) ----
            lds = np.zeros(BM * KPIX // 2, np.float64)          # one float16 slot per 2 bytes
            first = c * CK < k1
            src, k0 = (a1, c * CK) if first else (a2, c * CK - k1)
            for tid in range(NT):
                for k in range(LP):
                    u = tid + k * NT
                    r, c4 = divmod(u, UPP)
                    ok = m0 + r < M
                    row = m0 + r if ok else 0
                    woff = r * KPIX + 8 * c4
                    if presplit:
                        unit = src[row, 2 * (k0 + 4 * c4):2 * (k0 + 4 * c4) + 8].astype(np.float64) * (1.0 if ok else 0.0)
                        h4, l4 = unit[:4], unit[4:]              # a plain copy: no conversion
                    else:
                        x = src[row, k0 + 4 * c4:k0 + 4 * c4 + 4].astype(np.float32)
                        if in_bias is not None:
                            x = np.maximum(x + in_bias[c * CK + 4 * c4:c * CK + 4 * c4 + 4], 0).astype(np.float32)
                        h4, l4 = split((x * (sa if ok else np.float32(0))).astype(np.float32))
                    lds[woff // 2:woff // 2 + 4] = h4
                    lds[(woff + CK * 2) // 2:(woff + CK * 2) // 2 + 4] = l4
            if presplit:
                s_now = presplit[0] if first else presplit[1]
                if cur_scale is not None and s_now != cur_scale:
                    acc *= s_now / cur_scale                    # exact: both are powers of two
                cur_scale = s_now
            # ---- the four k-waves ---------------------------------------------------------------------------
            for wk in range(WK):
                for t in range(KPW):
                    kb = (c * KPW + t) * WK + wk
                    f = (n0 >> 5) * KB + kb
                    Bh, Bl = np.zeros((16, 32)), np.zeros((16, 32))
                    for lane in range(64):
                        j, g = lane & 31, lane >> 5
                        Bh[8 * g:8 * g + 8, j] = img[f, 0, lane]
                        Bl[8 * g:8 * g + 8, j] = img[f, 1, lane]
                    for mb in range(MB):
                        Ah, Al = np.zeros((32, 16)), np.zeros((32, 16))
                        for lane in range(64):
                            i, g = lane & 31, lane >> 5
                            addr = (mb * 32 + sigma(i)) * KPIX + (16 * wk + 8 * g) * 2 + t * WK * 32
                            Ah[i, 8 * g:8 * g + 8] = lds[addr // 2:addr // 2 + 8]
                            Al[i, 8 * g:8 * g + 8] = lds[(addr + CK * 2) // 2:(addr + CK * 2) // 2 + 8]
                        Dm = Al @ Bh + Ah @ Bl + Ah @ Bh
                        for lane in range(64):
                            j, g = lane & 31, lane >> 5
                            for r in range(16):
                                acc[wk, mb, r, lane] += Dm[(r & 3) + 8 * (r >> 2) + 4 * g, j]
        tot = (acc[0] + acc[2]) + (acc[1] + acc[3])
        descale = (1.0 / float(cur_scale if presplit else sa)) * (1.0 / np.asarray(s_w, np.float64))   # [N]: per column
        for wk in range(WK):                                    # wave wk finishes register quad wk
            for lane in range(64):
                j, g = lane & 31, lane >> 5
                L0 = sigma(8 * wk + 4 * g)
                for mb in range(MB):
                    for e in range(4):
                        r, m = 4 * wk + e, m0 + mb * 32 + L0 + e
                        v = tot[mb, r, lane] * descale[n0 + j] + float(bias[n0 + j])
                        if relu:
                            v = max(v, 0.0)
                        if m < M:
                            assert (m, n0 + j) not in out, "two lanes store the same element"
                            out[(m, n0 + j)] = v
                            vmax = max(vmax, abs(v))
    return out, vmax


def dense(a1, a2, w_kn, bias, MB, NW, KPW, in_bias=None, relu=True, rows_per_image=0, presplit=None):
    """whole layer -> out [M][N]; rows_per_image > 0: maxima (scales) per image"""
    M, N = a1.shape[0], w_kn.shape[1]
    img, s_w = pack(w_kn)
    out = np.full((M, N), np.nan)
    BM = 32 * MB
    for mt in range((M + BM - 1) // BM):
        lo, hi = (0, M) if rows_per_image <= 0 else ((mt * BM) // rows_per_image * rows_per_image,
                                                     (mt * BM) // rows_per_image * rows_per_image + rows_per_image)
        if presplit:
            amax = (0.0, 0.0)
        else:
            amax = (float(np.abs(a1[lo:hi]).max()), 0.0 if a2 is None else float(np.abs(a2[lo:hi]).max()))
        for nt in range(N // (32 * NW)):
            o, _ = dense_tile(a1, a2, img, s_w, bias, MB, NW, KPW, (mt, nt), amax, in_bias, relu, presplit)
            for k, v in o.items():
                assert np.isnan(out[k]), "two workgroups store the same element"
                out[k] = v
    return out


def lds_conflicts(MB, KPW):
    """extra LDS cycles of (a) one A-fragment ds_read_b128 and (b) the loader's two ds_write_b64 per unit, with the
    banking rules of MI355X_MICROARCH.md (64 banks x 4 bytes; a b128 access is served in four 16-lane groups, a b64
    in two 32-lane halves): 0 = conflict-free"""
    CK = 16 * WK * KPW
    KPIX, UPP = CK * 4 + 16, CK // 4
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups = groups + [[l + 32 for l in grp] for grp in groups]
    worst_r = 0
    for wk in range(WK):
        for mb in range(MB):
            for t in range(KPW):
                for plane in (0, CK * 2):
                    for grp in groups:
                        slots = {}
                        for lane in grp:
                            i, g = lane & 31, lane >> 5
                            a = (mb * 32 + sigma(i)) * KPIX + (16 * wk + 8 * g) * 2 + t * WK * 32 + plane
                            slots.setdefault((a // 16) % 16, set()).add(a)
                        worst_r = max(worst_r, max(len(v) for v in slots.values()) - 1)
    worst_w = 0
    for base in range(0, 64 * 4, 64):                            # a few waves' worth of units
        for half in (0, 32):
            for plane in (0, CK * 2):
                banks = {}
                for lane in range(half, half + 32):
                    u = base + lane
                    r, c4 = divmod(u, UPP)
                    a = r * KPIX + 8 * c4 + plane
                    for w in (0, 4):
                        banks.setdefault(((a + w) // 4) % 64, set()).add(a + w)
                worst_w = max(worst_w, max(len(v) for v in banks.values()) - 1)
    return worst_r, worst_w
