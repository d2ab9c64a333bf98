"""Lane-level numpy emulation of disn_amd/csrc/mlp_fused.hip (test infrastructure, CPU only).

It restates, index for index, what the HIP kernel does with its registers -- the packed weight
stream (fm_pair_coords / fm_pack_kernel), the v_mfma_f32_32x32x16_f16 operand and result layouts,
the slot <-> feature map phi that turns an output tile into the next layer's operand without
cross-lane traffic, the power-of-two scales and the two-term fp16 split -- so that the layout
contract can be checked against a plain matrix product WITHOUT a GPU, and the device pack
kernel can be compared bit for bit with ``pack_image`` on the GPU.

MFMA semantics used (cdna_hip_programming.md section 3):
  A operand: lane l holds A[i = l & 31][k = 8 (l >> 5) + t], t = 0..7
  B operand: lane l holds B[k = 8 (l >> 5) + t][j = l & 31]
  C/D:       lane l, register r holds D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]
"""
from __future__ import annotations

import numpy as np

PAIRS_L2, PAIRS_A, PAIRS_B = 32, 48, 16
PAIRS = PAIRS_L2 + 16 * PAIRS_A + 16 * PAIRS_B      # 1056
LAYER_DIMS = [(64, 256), (256, 512), (512, 512), (512, 256)]   # conv2, conv3, fold2/conv1 (point rows), fold2/conv2


PAIRS_A2 = 96                                        # FEAT form: phase A2, 16 iterations of six feature blocks
PAIRS_FEAT = PAIRS + 16 * PAIRS_A2                   # 2592
FEAT_REAL, FEAT_COLS = 1472, 1536


def pair_coords(p: int, feat: bool = False):
    """fm_pair_coords: stream position -> (layer, output tile, reduction block)"""
    if p < PAIRS_L2:
        return 0, p >> 2, p & 3
    p -= PAIRS_L2
    if p < 16 * PAIRS_A:
        it, r = divmod(p, PAIRS_A)
        if r < 16:
            return 1, it, r
        return 2, (r - 16) & 15, 2 * it + ((r - 16) >> 4)
    p -= 16 * PAIRS_A
    if feat:
        if p < 16 * PAIRS_A2:
            it2, r = divmod(p, PAIRS_A2)
            return 2, r & 15, 32 + 6 * it2 + 2 * (r >> 5) + ((r >> 4) & 1)
        p -= 16 * PAIRS_A2
    return 3, p & 7, 2 * (p >> 4) + ((p >> 3) & 1)


def phi(kb: int, g, t):
    """feature held in reduction slot (g, t) of block kb"""
    return 16 * kb + (t & 3) + 8 * (t >> 2) + 4 * g


def pow2_scale_for(amax: float, target_exp: int) -> float:
    if not (amax > 0) or not np.isfinite(amax):
        return 1.0
    e = int(np.floor(np.log2(np.float32(amax))))
    return float(2.0 ** (target_exp - e))


def split16(v: np.ndarray):
    """two-term fp16 split of float32 values (round to nearest even, like v_cvt_f16_f32)"""
    v = v.astype(np.float32)
    h = v.astype(np.float16)
    l = (v - h.astype(np.float32)).astype(np.float16)
    return h, l


ISW_OFF = (64, 64 + 256, 64 + 256 + 512, 64 + 256 + 512 + 512)   # fm::mS2 .. mS5
META = 64 + 256 + 512 + 512 + 256


def pack_image(w2, w3, w4p, w5):
    """-> (image [PAIRS, 2, 64, 8] float16, meta float32[META]) exactly as fm_meta_kernel + fm_pack_kernel.
    EQUALISED hidden features: layer l's feature f carries the power-of-two factor c_l[f] = 2^-e(max_k |W_l[k, f]| /
    c_{l-1}[k]) (c_1 = 1; the gathered feature rows of the FEAT form have no row factor): the image holds
    W~_l = diag(1 / c_{l-1}) W_l diag(c_l) 2^13.  meta[8] / [9] / [10]: the largest column 1-norms of W~4's point rows /
    W~3 / W~4's feature rows (the bounds), meta[11] = max c_4, from float 64 on c_2[256], c_3[512], c_4[512], c_5[256]"""
    ws = [np.asarray(w, np.float32) for w in (w2, w3, w4p, w5)]
    feat = ws[2].shape[0] > 512              # the FEAT image: w4p is the whole [512 + 1472][512] matrix
    meta = np.zeros(META, np.float32)
    rrow = np.ones(512, np.float32)
    wt = []
    for i, w in enumerate(ws):
        K = min(w.shape[0], 512) if i == 2 else w.shape[0]          # rows that are hidden features
        r = np.ones(w.shape[0], np.float32)
        r[:K] = rrow[:K]
        wr = (w * r[:, None]).astype(np.float32)
        cm = np.abs(wr).max(axis=0)
        cm = np.maximum(cm, np.float32(cm.max() * np.float32(2.0 ** -20)))   # fm::kColFloor: a dead column is not blown up
        c = np.array([pow2_scale_for(float(m), 0) for m in cm], np.float32)
        meta[ISW_OFF[i]:ISW_OFF[i] + w.shape[1]] = c
        l1p = (np.abs(wr[:K]).sum(axis=0, dtype=np.float32) * c).max()
        if i == 1:
            meta[9] = np.float32(l1p * np.float32(1.0001))
        if i == 2:
            meta[8] = np.float32(l1p * np.float32(1.0001))
            meta[11] = c.max()
            if feat:
                meta[10] = np.float32((np.abs(wr[K:]).sum(axis=0, dtype=np.float32) * c).max() * np.float32(1.0001))
        wt.append((wr * c[None, :] * np.float32(8192.0)).astype(np.float32))
        rrow = np.ones(512, np.float32)
        rrow[:w.shape[1]] = np.float32(1.0) / c
    if feat:
        wt[2] = np.concatenate([wt[2], np.zeros((512 + FEAT_COLS - wt[2].shape[0], 512), np.float32)])
    npairs = PAIRS_FEAT if feat else PAIRS
    img = np.zeros((npairs, 2, 64, 8), np.float16)
    lane = np.arange(64)
    i, g = lane & 31, lane >> 5
    t = np.arange(8)
    for p in range(npairs):
        layer, nt, kb = pair_coords(p, feat)
        w = wt[layer]
        k = phi(kb, g[:, None], t[None, :])                       # [64, 8]
        if layer == 2 and kb >= 32:                               # feature blocks: natural slot order
            k = 16 * kb + 8 * g[:, None] + t[None, :]
        img[p, 0], img[p, 1] = split16(w[k, (32 * nt + i)[:, None]])
    return img, meta


def mfma(acc, a_frag, b_frag):
    """acc [64 lanes, 16 regs] += A (lanes (i,g) x 8) . B (lanes (j,g) x 8), fp32 accumulation of exact products"""
    a = a_frag.astype(np.float32).reshape(2, 32, 8)       # [g, i, t]
    b = b_frag.astype(np.float32).reshape(2, 32, 8)       # [g, j, t]
    d = np.einsum("git,gjt->ij", a, b, dtype=np.float32)  # D[i][j]
    lane = np.arange(64)
    j, g = lane & 31, lane >> 5
    r = np.arange(16)
    rows = (r[None, :] & 3) + 8 * (r[None, :] >> 2) + 4 * g[:, None]   # [64, 16]
    return acc + d[rows, j[:, None]]


def exp_of(m):
    m = np.maximum(m.astype(np.float32), np.float32(0))
    e = np.where(m > 0, np.floor(np.log2(np.maximum(m, np.float32(1e-45)))), -127).astype(np.int32)
    return np.clip(e, -100, 100)


def feat_split_scale(featmax: float) -> np.float32:
    """the image's power-of-two feature scale (elementwise.hip split_pow2_scale o feat_split_amax)"""
    return np.float32(pow2_scale_for(max(float(featmax), 2.0 ** -20), 14))


def split_rows(feat: np.ndarray, featmax: float) -> np.ndarray:
    """project_gather_taps_kernel's SPLIT form of feature rows [n][1472] -> bytes [n][1536 * 4]: every 8 channels as
    [h8 | l8] (f16 planes of feature * scale), the padding columns zero"""
    n = feat.shape[0]
    x = np.zeros((n, FEAT_COLS), np.float32)
    x[:, :FEAT_REAL] = feat.astype(np.float32) * feat_split_scale(featmax)
    h, l = split16(x)
    out = np.zeros((n, FEAT_COLS // 8, 2, 8), np.float16)
    out[:, :, 0] = h.reshape(n, -1, 8)
    out[:, :, 1] = l.reshape(n, -1, 8)
    return out.reshape(n, -1).view(np.uint8)


def fused_stream(img, meta, consts, pts, add4=None, feat_rows=None, featmax=None):
    """One MLP stream for ONE wave (32 points): consts = dict(w1[3,64], b1, b2, b3, b4[512], b5, w6[256], b6);
    add4: optional [32 points, 512] additive term of fold2/conv1 (the resampled pmap rows); its |max| bound is
    consts['addmax4'].  FEAT form: feat_rows = split_rows(...) of the wave's 32 points (uint8 [32][6144]), featmax the
    image's bound; `img` then is the FEAT image.  Returns the 32 per-point sums (fold2/conv5 output incl. b6)."""
    featf = feat_rows is not None
    npairs = PAIRS_FEAT if featf else PAIRS
    lane = np.arange(64)
    j, g = lane & 31, lane >> 5
    f32 = np.float32
    c2, c3, c4, c5 = [meta[o:o + n] for o, n in zip(ISW_OFF, (256, 512, 512, 256))]     # the equalisation factors
    inv_sw = f32(1.0 / 8192.0)
    b2, b3, b4, b5 = consts["b2"] * c2, consts["b3"] * c3, consts["b4"] * c4, consts["b5"] * c5
    w6 = (consts["w6"] / c5).astype(f32)
    if add4 is not None:
        add4 = (add4 * c4[None, :]).astype(f32)
    cw4, cw3 = meta[8], meta[9]
    addmax4 = np.float32(np.float32(consts.get("add4max", 0.0)) * meta[11] + np.float32(np.abs(b4).max()))
    if featf:
        fmax = np.float32(max(float(featmax), 2.0 ** -20))
        sfeat = feat_split_scale(featmax)
        addmax4 = np.float32(fmax * meta[10] + np.float32(np.abs(b4).max()))
    x, y, z = (pts[j, c].astype(f32) for c in range(3))
    t = np.arange(8)
    r16 = np.arange(16)
    feat_of_reg = lambda nt: 32 * nt + (r16[None, :] & 3) + 8 * (r16[None, :] >> 2) + 4 * g[:, None]   # [64,16]

    # fold1/conv1 in slot order
    e1 = np.zeros((4, 64, 8), f32)
    for kb in range(4):
        f = phi(kb, g[:, None], t[None, :])
        e1[kb] = np.maximum(x[:, None] * consts["w1"][0, f] + y[:, None] * consts["w1"][1, f]
                            + z[:, None] * consts["w1"][2, f] + consts["b1"][f], 0).astype(f32)
    m = e1.max(axis=(0, 2))
    m = np.maximum(m, m[lane ^ 32])
    e = exp_of(m)
    s = (2.0 ** (14 - e)).astype(f32)
    inv2 = (2.0 ** (e - 14)).astype(f32) * inv_sw
    x1 = [split16(e1[kb] * s[:, None]) for kb in range(4)]

    p = 0   # stream position

    def pair(acc, xh, xl):
        nonlocal p
        wh, wl = img[p, 0], img[p, 1]
        p += 1
        acc = mfma(acc, wl, xh)
        acc = mfma(acc, wh, xl)
        return mfma(acc, wh, xh)

    # fold1/conv2
    z2 = np.zeros((8, 64, 16), f32)
    for nt in range(8):
        for kb in range(4):
            assert pair_coords(p, featf) == (0, nt, kb)
            z2[nt] = pair(z2[nt], *x1[kb])
    for nt in range(8):
        z2[nt] = np.maximum(z2[nt] * inv2[:, None] + b2[feat_of_reg(nt)], 0)
    m = z2.max(axis=(0, 2))
    m = np.maximum(m, m[lane ^ 32])
    e2 = exp_of(m)
    s2 = (2.0 ** (14 - e2)).astype(f32)
    inv3 = (2.0 ** (e2 - 14)).astype(f32) * inv_sw
    bound3 = (m * cw3 + np.abs(b3).max()).astype(f32)
    e3 = exp_of(bound3)
    s3 = (2.0 ** (14 - e3)).astype(f32)
    inv4 = (2.0 ** (e3 - 14)).astype(f32) * inv_sw
    e4 = exp_of((bound3 * cw4 + addmax4).astype(f32))
    s4 = (2.0 ** (14 - e4)).astype(f32)
    inv5 = (2.0 ** (e4 - 14)).astype(f32) * inv_sw
    x2 = []
    for nt in range(8):
        for hf in range(2):
            x2.append(split16(z2[nt][:, 8 * hf:8 * hf + 8] * s2[:, None]))

    def tile_to_frags(acc, bias, inv, sc, nt, add=None):
        v = acc * inv[:, None] + bias[feat_of_reg(nt)]
        if add is not None:
            v = v + add[j[:, None], feat_of_reg(nt)]
        v = (np.maximum(v, 0) * sc[:, None]).astype(f32)
        assert np.abs(v).max() < 65504, "scaled activation left fp16's range: the bound is wrong"
        return [split16(v[:, 0:8]), split16(v[:, 8:16])]

    # phase A
    acc4 = np.zeros((16, 64, 16), f32)
    for it in range(16):
        acc = np.zeros((64, 16), f32)
        for kb in range(16):
            assert pair_coords(p, featf) == (1, it, kb)
            acc = pair(acc, *x2[kb])
        fr = tile_to_frags(acc, b3, inv3, s3, it)
        for r in range(32):
            assert pair_coords(p, featf) == (2, r & 15, 2 * it + (r >> 4))
            acc4[r & 15] = pair(acc4[r & 15], *fr[r >> 4])
    # phase A2 (FEAT): one exact rescale s_feat / s3 per point, then the 96 feature blocks from the split rows
    if featf:
        acc4 = acc4 * (sfeat / s3)[None, :, None]
        inv4 = np.full(64, f32(1.0) / sfeat * inv_sw, f32)
        rows16 = feat_rows.view(np.float16).reshape(32, FEAT_COLS // 8, 2, 8)
        for it2 in range(16):
            fr6 = []
            for b in range(6):                       # lane (j, g): channel group 2 (6 it2 + b) + g of its row
                grp = 2 * (6 * it2 + b) + g
                fr6.append((rows16[j, grp, 0], rows16[j, grp, 1]))
            for r in range(96):
                blk = 2 * (r >> 5) + ((r >> 4) & 1)
                assert pair_coords(p, True) == (2, r & 15, 32 + 6 * it2 + blk)
                acc4[r & 15] = pair(acc4[r & 15], *fr6[blk])
    # phase B
    acc5 = np.zeros((8, 64, 16), f32)
    for it in range(16):
        fr = tile_to_frags(acc4[it], b4, inv4, s4, it, add4)
        for r in range(16):
            assert pair_coords(p, featf) == (3, r & 7, 2 * it + (r >> 3))
            acc5[r & 7] = pair(acc5[r & 7], *fr[r >> 3])
    assert p == npairs
    dot = np.zeros(64, f32)
    for nt in range(8):
        h5 = np.maximum(acc5[nt] * inv5[:, None] + b5[feat_of_reg(nt)], 0)
        dot += (h5 * w6[feat_of_reg(nt)]).sum(axis=1, dtype=f32)
    dot = dot + dot[lane ^ 32] + f32(consts["b6"])
    return dot[:32]
