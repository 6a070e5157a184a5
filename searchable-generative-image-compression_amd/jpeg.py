"""Host side of the GPU JPEG decoder (csrc/jpeg.hip): marker parsing, Huffman lookup tables, removal of the 0xFF00
byte stuffing / RSTn markers -- byte shuffling only, no entropy decoding -- and the batch descriptor the kernels read.

Supported on the GPU (everything Pillow's `save(..., "JPEG")` and ordinary cameras / encoders write): baseline or extended-sequential
Huffman JPEG, 8-bit, one interleaved scan, greyscale or YCbCr with 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 sampling, restart intervals,
custom Huffman / quantisation tables.  Anything else (progressive, arithmetic coding, CMYK / YCCK, RGB-coded, 12-bit, multi-scan,
chroma planes too narrow for the fancy upsampler) raises `Unsupported`, and the ingest falls back to the host decoder for that
batch -- the result is the same pixels either way, the GPU path being bit-exact with Pillow (tests/test_gpu_jpeg.py)."""
import struct

import numpy as np

NP = 64
LOOK = 9
TAB_BYTES = 1424
CHUNK = 2048
P_SCAN_OFF, P_SCAN_LEN, P_TAB_OFF, P_QUANT_OFF, P_NCOMP, P_W, P_H, P_HMAX, P_VMAX, P_MCUS_X, P_MCUS_Y, P_RESTART, P_SEG_OFF, P_NSEG = range(14)
P_COMP0, P_CSTRIDE = 16, 12

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
                   56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)


class Unsupported(ValueError):
    """a JPEG variant the GPU decoder does not take (the caller decodes it on the host instead)"""


def huff_table(bits, vals):
    """JPEG DHT (16 code-length counts + symbols) -> the 1424-byte table of csrc/jpeg.hip:
    u16 fast[512] (nbits << 8 | symbol for codes of <= 9 bits, else 0) | i32 maxcode[18] | i32 valoff[17] | pad | u8 huffval[256]
    (the canonical code assignment of ITU T.81 Annex C, as jdhuff.c jpeg_make_d_derived_tbl)"""
    fast = np.zeros(1 << LOOK, dtype=np.uint16)
    maxcode = np.full(18, -1, dtype=np.int32)
    valoff = np.zeros(17, dtype=np.int32)
    huffval = np.zeros(256, dtype=np.uint8)
    huffval[:len(vals)] = vals
    code, k = 0, 0
    for l in range(1, 17):
        n = int(bits[l - 1])
        if n:
            valoff[l] = k - code
            for _ in range(n):
                if l <= LOOK:
                    lo = code << (LOOK - l)
                    fast[lo:lo + (1 << (LOOK - l))] = (l << 8) | int(vals[k])
                code += 1
                k += 1
            maxcode[l] = code - 1
        if code > (1 << l):
            raise Unsupported("bad Huffman table")
        code <<= 1
    maxcode[17] = 0x7fffffff
    out = np.zeros(TAB_BYTES, dtype=np.uint8)
    out[0:1024] = fast.view(np.uint8)
    out[1024:1096] = maxcode.view(np.uint8)
    out[1096:1164] = valoff.view(np.uint8)
    out[1168:1424] = huffval
    return out


class Parsed:
    __slots__ = ("W", "H", "ncomp", "comps", "hmax", "vmax", "quant", "tabs", "scan", "segs", "restart", "mcus_x", "mcus_y")


def parse(data):
    """bytes of one JPEG file -> Parsed (tables, geometry, cleaned scan); raises Unsupported for variants outside the GPU path"""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("not a JPEG")
    qt = {}
    ht = {}
    restart = 0
    frame = None
    adobe_transform = None
    pos = 2
    n = len(data)
    while True:
        if pos + 4 > n:
            raise Unsupported("truncated before SOS")
        if data[pos] != 0xFF:
            raise Unsupported("marker expected")
        while data[pos + 1] == 0xFF:
            pos += 1
        m = data[pos + 1]
        pos += 2
        if m in (0x01,) or 0xD0 <= m <= 0xD7:
            continue
        ln = struct.unpack(">H", data[pos:pos + 2])[0]
        seg = data[pos + 2:pos + ln]
        if m == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                if pq:
                    t = np.frombuffer(seg[i:i + 128], dtype=">u2").astype(np.uint16)
                    i += 128
                else:
                    t = np.frombuffer(seg[i:i + 64], dtype=np.uint8).astype(np.uint16)
                    i += 64
                nat = np.zeros(64, dtype=np.uint16)
                nat[ZIGZAG] = t
                qt[tq] = nat
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                bits = np.frombuffer(seg[i + 1:i + 17], dtype=np.uint8)
                cnt = int(bits.sum())
                vals = np.frombuffer(seg[i + 17:i + 17 + cnt], dtype=np.uint8)
                i += 17 + cnt
                if th > 1 or tc > 1:
                    raise Unsupported("more than two Huffman tables per class")
                ht[(tc, th)] = huff_table(bits, vals)
        elif m in (0xC0, 0xC1):
            prec, H, W, nc = struct.unpack(">BHHB", seg[:6])
            if prec != 8:
                raise Unsupported("not 8-bit")
            frame = (H, W, [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(nc)])
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise Unsupported("progressive / lossless / arithmetic JPEG")
        elif m == 0xDD:
            restart = struct.unpack(">H", seg[:2])[0]
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            adobe_transform = seg[11]
        elif m == 0xDA:
            if frame is None:
                raise Unsupported("SOS before SOF")
            ns = seg[0]
            H, W, comps = frame
            if ns != len(comps):
                raise Unsupported("non-interleaved multi-scan file")
            sel = {}
            for c in range(ns):
                sel[seg[1 + 2 * c]] = (seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15)
            ss, se, ahal = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]
            if (ss, se, ahal) != (0, 63, 0):
                raise Unsupported("not a sequential scan")
            pos += ln
            break
        elif m == 0xD9:
            raise Unsupported("EOI before SOS")
        pos += ln
    H, W, comps = frame
    nc = len(comps)
    if nc not in (1, 3):
        raise Unsupported(f"{nc} components")
    if nc == 3:
        ids = [c[0] for c in comps]
        if adobe_transform == 0 or (adobe_transform is None and ids == [ord("R"), ord("G"), ord("B")]):
            raise Unsupported("RGB-coded JPEG")
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    if nc == 3:
        if (comps[0][1], comps[0][2]) != (hmax, vmax) or (comps[1][1:3] != comps[2][1:3]) or comps[1][1:3] != (1, 1) or hmax > 2 or vmax > 2:
            raise Unsupported("sampling factors outside 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0")
    else:
        hmax = vmax = 1                      # a single-component scan is never interleaved (T.81 A.2.2)
        comps = [(comps[0][0], 1, 1, comps[0][3])]
    p = Parsed()
    p.W, p.H, p.ncomp, p.hmax, p.vmax, p.restart = W, H, nc, hmax, vmax, restart
    p.mcus_x, p.mcus_y = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    p.comps = []
    for (cid, h, v, tq) in comps:
        if tq not in qt or cid not in sel or (0, sel[cid][0]) not in ht or (1, sel[cid][1]) not in ht:
            raise Unsupported("missing table")
        cw, ch = -(-W * h // hmax), -(-H * v // vmax)
        if (h < hmax or v < vmax) and cw <= 2:
            raise Unsupported("chroma plane too narrow for the fancy upsampler")
        p.comps.append(dict(h=h, v=v, tq=tq, dc=sel[cid][0], ac=sel[cid][1], bw=p.mcus_x * h, bh=p.mcus_y * v, cw=cw, ch=ch))
    p.quant = np.zeros((4, 64), dtype=np.uint16)
    for k, t in qt.items():
        if k < 4:
            p.quant[k] = t
    p.tabs = np.zeros((4, TAB_BYTES), dtype=np.uint8)
    for (tc, th), t in ht.items():
        p.tabs[2 * tc + th] = t
    # entropy-coded segment: up to the marker that is not RSTn / stuffing; un-stuff, drop the RSTn markers and remember where each
    # restart interval starts in the cleaned stream
    raw = np.frombuffer(data, dtype=np.uint8, offset=pos)
    ff = np.nonzero(raw[:-1] == 0xFF)[0]
    nxt = raw[ff + 1]
    stop = ff[(nxt != 0) & ~((nxt >= 0xD0) & (nxt <= 0xD7)) & (nxt != 0xFF)]
    end = int(stop[0]) if len(stop) else len(raw)
    raw = raw[:end]
    ff = ff[ff < end - 1] if end > 0 else ff[:0]
    nxt = raw[ff + 1] if len(ff) else nxt[:0]
    keep = np.ones(len(raw), dtype=bool)
    stuffed = ff[nxt == 0]
    keep[stuffed + 1] = False
    rst = ff[(nxt >= 0xD0) & (nxt <= 0xD7)]
    keep[rst] = False
    keep[rst + 1] = False
    keep[ff[nxt == 0xFF]] = False                   # fill bytes (an 0xFF in front of another 0xFF) are not data (T.81 B.1.1.2)
    newpos = np.cumsum(keep) - keep                 # position of every original byte in the cleaned stream
    p.scan = raw[keep]
    p.segs = np.concatenate([[0], newpos[rst]]).astype(np.int32) if len(rst) else np.zeros(1, dtype=np.int32)
    if restart and len(p.segs) < -(-p.mcus_x * p.mcus_y // restart):
        raise Unsupported("restart markers missing")
    return p


class JpegBatch:
    """host-side descriptor of a batch of equal-geometry JPEGs: ONE contiguous blob
        params (B x 64 i32) | restart offsets (i32) | quant tables (B x 256 u16) | Huffman lookup tables (B x 4 x 1424 B) | cleaned scans
    so that the batch crosses PCIe in a single copy.  `alloc(nbytes) -> uint8 torch tensor` supplies the staging memory: the ingest
    passes PINNED slot buffers -- a copy from pageable memory is synchronous in HIP and, measured in the CLI trace, waited for the
    previous batch's GPU work (the decode then ran alone instead of under it)."""

    def __init__(self, datas, pool=None, alloc=None):
        import torch
        ps = list(pool.map(parse, datas)) if pool is not None else [parse(d) for d in datas]
        self.last_err = None
        H, W = ps[0].H, ps[0].W
        if any((p.H, p.W) != (H, W) for p in ps):
            raise ValueError("a JPEG batch shares one geometry")
        B = len(ps)
        self.B, self.H, self.W = B, H, W
        nseg = sum(len(p.segs) for p in ps)
        scan_len = [len(p.scan) + ((-len(p.scan)) % CHUNK or (CHUNK if len(p.scan) == 0 else 0)) for p in ps]
        al = lambda n: (n + 255) & ~255
        self.off_params, n = 0, al(B * NP * 4)
        self.off_segs, n = n, n + al(nseg * 4)
        self.off_quant, n = n, n + al(B * 256 * 2)
        self.off_tabs, n = n, n + al(B * 4 * TAB_BYTES)
        self.off_scan, n = n, n + sum(scan_len)
        self.nseg, self.nbytes = nseg, n
        blob = alloc(n + al(4 * B)) if alloc is not None else torch.empty(n + al(4 * B), dtype=torch.uint8)
        self.blob = blob[:n]
        self.err_host = blob[n:n + 4 * B].view(torch.int32)     # the error codes come back into the same (pinned) staging buffer
        host = self.blob.numpy()
        host[:self.off_scan] = 0
        params = host[self.off_params:self.off_params + B * NP * 4].view(np.int32).reshape(B, NP)
        segs = host[self.off_segs:self.off_segs + nseg * 4].view(np.int32)
        quant = host[self.off_quant:self.off_quant + B * 512].view(np.uint16).reshape(B, 256)
        tabs = host[self.off_tabs:self.off_tabs + B * 4 * TAB_BYTES].reshape(B, 4 * TAB_BYTES)
        scan = host[self.off_scan:]
        scan_off = seg_off = blk_off = plane_off = 0
        self.max_blocks = 0
        for b, p in enumerate(ps):
            r = params[b]
            r[P_SCAN_OFF], r[P_SCAN_LEN], r[P_TAB_OFF], r[P_QUANT_OFF] = scan_off, scan_len[b], b * 4 * TAB_BYTES, b * 256
            r[P_NCOMP], r[P_W], r[P_H], r[P_HMAX], r[P_VMAX] = p.ncomp, W, H, p.hmax, p.vmax
            r[P_MCUS_X], r[P_MCUS_Y], r[P_RESTART], r[P_SEG_OFF], r[P_NSEG] = p.mcus_x, p.mcus_y, p.restart, seg_off, len(p.segs)
            nblk = 0
            for c, d in enumerate(p.comps):
                o = P_COMP0 + c * P_CSTRIDE
                r[o:o + 11] = [d["h"], d["v"], d["tq"], d["dc"], d["ac"], d["bw"], d["bh"], d["cw"], d["ch"], blk_off + nblk, plane_off]
                nblk += d["bw"] * d["bh"]
                plane_off += (d["bw"] * 8 * d["bh"] * 8 + 15) & ~15
            self.max_blocks = max(self.max_blocks, nblk)
            blk_off += nblk
            scan[scan_off:scan_off + len(p.scan)] = p.scan
            scan[scan_off + len(p.scan):scan_off + scan_len[b]] = 0
            segs[seg_off:seg_off + len(p.segs)] = p.segs
            quant[b] = p.quant.reshape(-1)
            tabs[b] = p.tabs.reshape(-1)
            scan_off += scan_len[b]
            seg_off += len(p.segs)
        self.params, self.tabs, self.scan = params, tabs.reshape(-1), scan     # host views (tests)
        self.total_blocks, self.plane_bytes = blk_off, plane_off

    def decode(self, device, out=None, check=True):
        """-> (B, H, W, 3) u8 device tensor on the current stream.  check=True synchronises and raises if a stream is corrupt;
        check=False leaves the per-image error codes in self.last_err (device int32 tensor) for the caller to read later"""
        import torch
        from . import ops
        # a PINNED blob is read by the kernels in place (device-accessible host memory: no H2D copy to schedule -- an SDMA copy would
        # queue behind the previous batch's result copies and start the decode only when that batch has finished); else one copy
        dev = torch.device(device)
        with torch.cuda.device(dev):
            return self._decode(self.blob if self.blob.is_pinned() else self.blob.to(dev, non_blocking=True), out, check)

    def _decode(self, d, out, check):
        from . import ops
        B = self.B
        view = lambda off, nbytes: d[off:off + nbytes]
        params = view(self.off_params, B * NP * 4)
        segs = view(self.off_segs, max(4, self.nseg * 4))
        quant = view(self.off_quant, B * 512)
        tabs = view(self.off_tabs, B * 4 * TAB_BYTES)
        scan = d[self.off_scan:]
        out, self.last_err = ops.jpeg_decode_batch(params, scan, tabs, segs, quant, B, self.H, self.W, self.total_blocks,
                                                   self.plane_bytes, self.max_blocks, out=out, check=check)
        return out
