"""Seeded JPEG test files (bytes) written with the installed Pillow: every sampling layout / table variant the GPU decoder takes"""
import io

import numpy as np


def natural_like(h, w, rng, grey=False):
    """low-pass noise + fine noise: all 64 DCT coefficients get exercised, chroma has structure"""
    from PIL import Image
    a = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.uint8)
    im = np.asarray(Image.fromarray(a).resize((w, h), Image.BICUBIC)).astype(np.int32) + rng.integers(-12, 13, (h, w, 3))
    im = np.clip(im, 0, 255).astype(np.uint8)
    return im[:, :, 0] if grey else im


def cases(small=True):
    """-> list of (name, bytes)"""
    from PIL import Image
    rng = np.random.default_rng(7)
    spec = [(48, 64, dict(quality=75)), (37, 53, dict(quality=90, subsampling=0)), (40, 72, dict(quality=60, subsampling=1)),
            (33, 47, dict(quality=85, subsampling=2)), (64, 64, dict(quality=30, optimize=True)),
            (50, 50, dict(quality=95, subsampling=2, restart_marker_blocks=3)), (41, 29, dict(quality=80)), (16, 16, dict(quality=100, subsampling=0))]
    if not small:
        spec += [(256, 256, dict(quality=75)), (256, 256, dict(quality=92, subsampling=0, optimize=True)), (859, 1000, dict(quality=85)),
                 (511, 257, dict(quality=50, subsampling=1, restart_marker_rows=2)), (1024, 1024, dict(quality=98, subsampling=0))]
    out = []
    for h, w, kw in spec:
        buf = io.BytesIO()
        Image.fromarray(natural_like(h, w, rng)).save(buf, "JPEG", **kw)
        out.append((f"{h}x{w}_" + "_".join(f"{k}{v}" for k, v in kw.items()), buf.getvalue()))
    buf = io.BytesIO()
    Image.fromarray(natural_like(45, 61, rng, grey=True)).save(buf, "JPEG", quality=70)
    out.append(("grey_45x61", buf.getvalue()))
    # optional 0xFF fill bytes in front of a marker inside the entropy-coded segment (T.81 B.1.1.2): legal, rare, must be skipped
    buf = io.BytesIO()
    Image.fromarray(natural_like(48, 48, rng)).save(buf, "JPEG", quality=85, restart_marker_blocks=2)
    d = buf.getvalue()
    i = d.find(b"\xff\xd0", d.find(b"\xff\xda"))
    out.append(("fill_bytes_before_rst", d[:i] + b"\xff\xff" + d[i:]))
    return out
