#!/usr/bin/env python3
"""Counterpart of the reference driver src/compress.py (same flags, same outputs:
save_dir/{bitstreams/<stem>.c2df, clip_vecs/<stem>.npy, faiss/index.faiss, faiss/ids.txt}), running the
MI355X path.  Differences, all MI355X-native by design:
  * images of equal padded size are batched (--batch_size, default 32) instead of B=1;
  * multi-GPU: one process per GPU (torchrun), every rank loads its own weights (no DDP broadcast,
    compress.py:242), takes a contiguous shard of the sorted file list, and the CLIP vectors are
    all-gathered with RCCL instead of going through *.npy files on a shared file system (compress.py:295-306);
  * without --ckpt_path / --clip_ckpt the deterministic synthetic weights are used (there is no
    checkpoint offline); a real checkpoint with the reference's state_dict keys loads unchanged.
"""
import argparse
import os
import sys
from glob import glob

import numpy as np
import torch

torch.set_grad_enabled(False)


def get_padding_size(height, width, p=64):
    """entropy/compression_model.py:13-22"""
    new_h = (height + p - 1) // p * p
    new_w = (width + p - 1) // p * p
    return 0, new_w - width, 0, new_h - height


def load_image(path):
    from PIL import Image
    a = np.array(Image.open(path).convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0) * 2.0 - 1.0   # ToTensor, *2-1 (compress.py:161-164)


def load_state(path, spec_fn, cfg, seed):
    from . import weights as W
    if path:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        if "visual.conv1.weight" in sd or "token_embedding.weight" in sd:   # a bare open_clip CLIP state_dict
            sd = {f"clip.{k}": v for k, v in sd.items()}
        return sd
    return W.synth_weights(spec_fn(cfg), seed=seed)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_config", type=str, default=None, help="accepted for CLI compatibility (architecture is fixed)")
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--clip_ckpt", type=str, default=None, help="open_clip ViT-B-32 state_dict (visual.* keys)")
    ap.add_argument("--dataset_dir", type=str, required=True)
    ap.add_argument("--save_dir", type=str, required=True)
    ap.add_argument("--gpu_idx", type=int, default=0)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--small", action="store_true", help="SMALL/TINY test architectures")
    args = ap.parse_args(argv)

    import torch.distributed as dist
    from . import weights as W
    from .codec import ClipCodec, Codec
    from .config import CLIP_B32, CLIP_TINY, LARGE, SMALL
    from .dist import gather_vectors, shard_range
    from .faiss_io import FaissDB
    from .filemaker import pack_c2df

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="nccl", init_method="env://")
        local = int(os.environ.get("LOCAL_RANK", "0"))
    else:
        local = args.gpu_idx
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfg, ccfg = (SMALL, CLIP_TINY) if args.small else (LARGE, CLIP_B32)
    sd = load_state(args.ckpt_path, lambda c: W.encoder_spec(c) + W.codec_misc_spec(c) + W.bottleneck_spec(c), cfg, 1234)
    csd = load_state(args.clip_ckpt, W.clip_spec, ccfg, 4321)
    if args.clip_ckpt and not any(k.startswith("clip.") for k in csd):
        csd = {f"clip.{k}": v for k, v in csd.items()}
    model = Codec(sd, cfg, dev)
    model.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    model.hybrid_codec.quantize_feat.update(force=True)
    clipc = ClipCodec(csd, ccfg, dev)

    bit_dir, index_dir, clip_dir = (os.path.join(args.save_dir, d) for d in ("bitstreams", "faiss", "clip_vecs"))
    if rank == 0:
        for d in (args.save_dir, bit_dir, index_dir, clip_dir):
            os.makedirs(d, exist_ok=True)
    if world > 1:
        dist.barrier()

    files = sorted(glob(os.path.join(args.dataset_dir, "*.*")))
    lo, hi = shard_range(len(files), rank, world)
    mine = files[lo:hi]
    # group by original size so that a batch shares padding and CLIP resize geometry
    # image decode on a small thread pool (PIL releases the GIL while decoding): at 300+ images/s per GPU a serial
    # Image.open loop would be the bottleneck of the driver
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        imgs = list(zip(mine, pool.map(load_image, mine)))
    groups = {}
    for i, (p, im) in enumerate(imgs):
        groups.setdefault(tuple(im.shape[1:]), []).append(i)
    vecs = np.zeros((len(mine), ccfg.embed_dim), dtype=np.float32)
    from . import ops
    from .pipeline import CompressPipeline
    pipe = CompressPipeline(model, clipc, dev, want_unit=True)

    def write_out(job):
        """host side of one batch: .c2df container + clip vector per image (compress.py:268-291)"""
        h, chunk, (H, Wd), pad = job
        pl, pr, pt, pb = pad
        for j, (i, streams) in enumerate(zip(chunk, pipe.finish(h))):
            stem = os.path.splitext(os.path.basename(imgs[i][0]))[0]
            enc = pipe.enc_result(h, j, streams)
            enc["clip_stream"] = streams["clip_stream"]
            enc["clip_meta"] = clipc.meta(ccfg.embed_dim)
            header = {"version": 2, "model_id": enc["clip_meta"]["model_id"], "embed_dim": int(ccfg.embed_dim),
                      "quant_type": "u8_symmetric_-1_1", "image_hw": [int(H), int(Wd)],
                      "padding": [int(pl), int(pr), int(pt), int(pb)]}
            with open(os.path.join(bit_dir, f"{stem}.c2df"), "wb") as f:
                f.write(pack_c2df(enc, header))
            np.save(os.path.join(clip_dir, f"{stem}.npy"), streams["clip_unit"])
            vecs[i] = streams["clip_unit"]

    # two-deep pipeline: the GPU works on batch k+1 while the host packs and writes batch k
    pending = None
    for (H, Wd), idxs in groups.items():
        pad = get_padding_size(H, Wd, p=256)
        pl, pr, pt, pb = pad
        for s in range(0, len(idxs), args.batch_size):
            chunk = idxs[s:s + args.batch_size]
            x = torch.stack([imgs[i][1] for i in chunk]).to(dev)
            xp = ops.pad_replicate(x.contiguous(), pl, pr, pt, pb)                               # compress.py:258-261
            h = pipe.submit(xp, clip_hw=(H, Wd))     # CLIP sees the UNPADDED top-left H x W region (compress.py:266)
            if pending is not None:
                write_out(pending)
            pending = (h, chunk, (H, Wd), pad)
    if pending is not None:
        write_out(pending)
    allv = gather_vectors(torch.from_numpy(vecs).to(dev), len(files), rank, world).cpu().numpy()
    if rank == 0 and len(files) > 0:
        db = FaissDB(index_dir, ccfg.embed_dim)
        for i, p in enumerate(files):
            stem = os.path.splitext(os.path.basename(p))[0]
            db.add(allv[i], os.path.join(bit_dir, f"{stem}.c2df"))
        db.persist()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
