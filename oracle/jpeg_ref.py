"""TEST INFRASTRUCTURE (oracle): plain numpy / Python restatement of the baseline-JPEG decode that csrc/jpeg.hip performs, reading
the same tables sgic_amd/jpeg.py builds.  It exists to pin the ALGORITHM -- libjpeg-turbo's default decoder, which is what the
reference's `Image.open(path).convert("RGB")` (compress.py:160) runs: jdhuff.c (Huffman), jidctint.c (islow IDCT),
jdsample.c (fancy upsampling h2v1 / h2v2 / h1v2), jdcolor.c (YCbCr -> RGB tables) -- against the installed Pillow on small images
in the CPU tests; the GPU tests then compare the kernels with Pillow directly.  Never imported by the product."""
import numpy as np

NATURAL = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
                    56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63] + [63] * 16)


class _Bits:
    def __init__(self, data):
        self.d, self.pos, self.acc, self.cnt = data, 0, 0, 0

    def fill(self, n):
        while self.cnt < n:
            b = int(self.d[self.pos]) if self.pos < len(self.d) else 0
            self.pos += 1
            self.acc = ((self.acc << 8) | b) & ((1 << 64) - 1)
            self.cnt += 8

    def peek(self, n):
        self.fill(n)
        return (self.acc >> (self.cnt - n)) & ((1 << n) - 1)

    def skip(self, n):
        self.cnt -= n
        self.acc &= (1 << self.cnt) - 1

    def seek(self, byte_off):
        self.pos, self.acc, self.cnt = byte_off, 0, 0


def _decode_sym(br, tab):
    fast = tab[:1024].view(np.uint16)
    maxcode = tab[1024:1096].view(np.int32)
    valoff = tab[1096:1164].view(np.int32)
    huffval = tab[1168:1424]
    e = int(fast[br.peek(9)])
    if e:
        br.skip(e >> 8)
        return e & 255
    code16 = br.peek(16)
    l = 10
    while l <= 16 and (code16 >> (16 - l)) > maxcode[l]:
        l += 1
    if l > 16:
        raise ValueError("invalid Huffman code")
    br.skip(l)
    return int(huffval[(valoff[l] + (code16 >> (16 - l))) & 255])


def _receive(br, s):
    v = br.peek(s)
    br.skip(s)
    return v - (1 << s) + 1 if v < (1 << (s - 1)) else v


def huffman_decode(p):
    """Parsed (sgic_amd.jpeg.parse) -> list of per-component coefficient arrays (bh, bw, 64) int16, natural order"""
    br = _Bits(p.scan)
    coefs = [np.zeros((c["bh"], c["bw"], 64), dtype=np.int16) for c in p.comps]
    pred = [0] * p.ncomp
    nseg = 1
    for mcu in range(p.mcus_x * p.mcus_y):
        if p.restart and mcu and mcu % p.restart == 0:
            br.seek(int(p.segs[nseg]))
            nseg += 1
            pred = [0] * p.ncomp
        my, mx = divmod(mcu, p.mcus_x)
        for c, d in enumerate(p.comps):
            hs, vs = (1, 1) if p.ncomp == 1 else (d["h"], d["v"])
            for by in range(vs):
                for bx in range(hs):
                    blk = coefs[c][my * vs + by, mx * hs + bx]
                    s = _decode_sym(br, p.tabs[d["dc"]])
                    if s:
                        pred[c] += _receive(br, s)
                    blk[0] = pred[c]
                    k = 1
                    while k < 64:
                        rs = _decode_sym(br, p.tabs[2 + d["ac"]])
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r != 15:
                                break
                            k += 16
                            continue
                        k += r
                        blk[NATURAL[k]] = _receive(br, s)
                        k += 1
    return coefs


def _idct_1d(x, shift):
    """jidctint.c: one 1-D pass of the LL&M integer IDCT over the last axis (8), int64 arithmetic"""
    x = x.astype(np.int64)
    i = [x[..., k] for k in range(8)]
    z2, z3 = i[2], i[6]
    z1 = (z2 + z3) * 4433
    tmp2 = z1 + z3 * -15137
    tmp3 = z1 + z2 * 6270
    z2, z3 = i[0], i[4]
    tmp0, tmp1 = (z2 + z3) << 13, (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = i[7], i[5], i[3], i[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * 9633
    tmp0, tmp1, tmp2, tmp3 = tmp0 * 2446, tmp1 * 16819, tmp2 * 25172, tmp3 * 12299
    z1, z2, z3, z4 = z1 * -7373, z2 * -20995, z3 * -16069 + z5, z4 * -3196 + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    out = np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1)
    return (out + (1 << (shift - 1))) >> shift


def idct_blocks(coef, quant):
    """(bh, bw, 64) int16, (64,) u16 natural order -> (bh*8, bw*8) u8 sample plane"""
    bh, bw, _ = coef.shape
    x = (coef.astype(np.int64) * quant.astype(np.int64)).reshape(bh, bw, 8, 8)          # [.., y, x]
    ws = _idct_1d(x.transpose(0, 1, 3, 2), 11).transpose(0, 1, 3, 2)                    # pass 1 down the columns
    o = _idct_1d(ws, 18)                                                                # pass 2 along the rows
    i = o & 1023
    lim = np.where(i < 128, i + 128, np.where(i < 512, 255, np.where(i < 896, 0, i - 896)))
    return lim.astype(np.uint8).transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample(pl, cw, ch, hs, vs, W, H):
    """jdsample.c fancy upsampling of a (ch, cw) chroma plane to (H, W)"""
    a = pl[:ch, :cw].astype(np.int32)
    if hs == 1 and vs == 1:
        return a[:H, :W]
    if vs == 2:
        up = np.concatenate([a[:1], a[:-1]], axis=0)      # row above (replicated at the top)
        dn = np.concatenate([a[1:], a[-1:]], axis=0)      # row below (replicated at the bottom)
        if hs == 1:
            out = np.empty((2 * ch, cw), dtype=np.int32)
            out[0::2] = (3 * a + up + 1) >> 2
            out[1::2] = (3 * a + dn + 2) >> 2
            return out[:H, :W]
        rows = np.empty((2 * ch, cw), dtype=np.int32)     # column sums 3 * near + far
        rows[0::2] = 3 * a + up
        rows[1::2] = 3 * a + dn
        prev = np.concatenate([rows[:, :1], rows[:, :-1]], axis=1)
        nxt = np.concatenate([rows[:, 1:], rows[:, -1:]], axis=1)
        out = np.empty((2 * ch, 2 * cw), dtype=np.int32)
        out[:, 0::2] = (rows * 3 + prev + 8) >> 4
        out[:, 1::2] = (rows * 3 + nxt + 7) >> 4
        out[:, 0] = (rows[:, 0] * 4 + 8) >> 4
        out[:, -1] = (rows[:, -1] * 4 + 7) >> 4
        return out[:H, :W]
    prev = np.concatenate([a[:, :1], a[:, :-1]], axis=1)
    nxt = np.concatenate([a[:, 1:], a[:, -1:]], axis=1)
    out = np.empty((ch, 2 * cw), dtype=np.int32)
    out[:, 0::2] = (3 * a + prev + 1) >> 2
    out[:, 1::2] = (3 * a + nxt + 2) >> 2
    out[:, 0] = a[:, 0]
    out[:, -1] = a[:, -1]
    return out[:H, :W]


def decode(data):
    """bytes of a baseline JPEG -> (H, W, 3) u8, as `np.asarray(Image.open(...).convert("RGB"))`"""
    import sgic_amd  # noqa: F401
    from sgic_amd import jpeg as J
    p = J.parse(data)
    coefs = huffman_decode(p)
    planes = [idct_blocks(coefs[c], p.quant[d["tq"]]) for c, d in enumerate(p.comps)]
    H, W = p.H, p.W
    y = planes[0][:H, :W].astype(np.int32)
    if p.ncomp == 1:
        return np.stack([y, y, y], axis=-1).astype(np.uint8)
    d1 = p.comps[1]
    hs, vs = p.hmax // d1["h"], p.vmax // d1["v"]
    cb = upsample(planes[1], d1["cw"], d1["ch"], hs, vs, W, H) - 128
    cr = upsample(planes[2], d1["cw"], d1["ch"], hs, vs, W, H) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)
