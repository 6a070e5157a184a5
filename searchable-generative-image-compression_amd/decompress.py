#!/usr/bin/env python3
"""Counterpart of the reference driver src/decompress.py (same flags; writes save_dir/results/<stem>.png),
running the MI355X decode path.  Files with identical shapes are decoded as one batch."""
import argparse
import os
import sys
from glob import glob

import numpy as np
import torch

torch.set_grad_enabled(False)


def to_u8_hwc(x_chw_01):
    """torchvision.utils.save_image semantics for one image: mul(255).add_(0.5).clamp_(0,255) -> u8 (decompress.py:114)"""
    return x_chw_01.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()


def save_png(x_chw_01, path):
    from PIL import Image
    Image.fromarray(to_u8_hwc(x_chw_01)).save(path)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_config", type=str, default=None)
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--dataset_dir", type=str, required=True, help="directory with *.c2df")
    ap.add_argument("--save_dir", type=str, required=True)
    ap.add_argument("--gpu_idx", type=int, default=0)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args(argv)

    from . import weights as W
    from .codec import Codec
    from .compress import load_state
    from .config import LARGE, SMALL
    from .filemaker import unpack_c2df

    torch.cuda.set_device(args.gpu_idx)
    dev = torch.device("cuda", args.gpu_idx)
    cfg = SMALL if args.small else LARGE
    sd = load_state(args.ckpt_path, W.full_spec, cfg, 1234)
    model = Codec(sd, cfg, dev)
    model.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    model.hybrid_codec.quantize_feat.update(force=True)
    out_dir = os.path.join(args.save_dir, "results")
    os.makedirs(out_dir, exist_ok=True)

    files = sorted(glob(os.path.join(args.dataset_dir, "*.c2df")))
    items = []
    for fp in files:
        enc, header = unpack_c2df(fp)
        items.append((fp, enc, header))
    groups = {}
    for i, (_, enc, _) in enumerate(items):
        groups.setdefault(tuple(int(v) for v in enc["img_shape"]), []).append(i)
    # PNG encoding (zlib, releases the GIL) runs on a thread pool underneath the next batch's GPU decode
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        pending = []
        for shape, idxs in groups.items():
            for s in range(0, len(idxs), args.batch_size):
                chunk = idxs[s:s + args.batch_size]
                x_hat = model.decode_batch([items[i][1] for i in chunk])
                for j, i in enumerate(chunk):
                    fp, _, header = items[i]
                    pl, pr, pt, pb = header.get("padding", [0, 0, 0, 0])
                    H, Wd = x_hat.shape[2] - pt - pb, x_hat.shape[3] - pl - pr   # negative-pad crop (decompress.py:110-112)
                    a = to_u8_hwc(x_hat[j, :, pt:pt + H, pl:pl + Wd].clamp(-1, 1) * 0.5 + 0.5)
                    path = os.path.join(out_dir, os.path.splitext(os.path.basename(fp))[0] + ".png")
                    pending.append(pool.submit(lambda arr, p: Image.fromarray(arr).save(p), a, path))
        for f in pending:
            f.result()
    return 0


if __name__ == "__main__":
    sys.exit(main())
