#!/usr/bin/env python3
"""Counterpart of the reference driver src/compress.py (same flags, same outputs:
save_dir/{bitstreams/<stem>.c2df, clip_vecs/<stem>.npy, faiss/index.faiss, faiss/ids.txt}), running the
MI355X path.  Differences, all MI355X-native by design:
  * the shard is STREAMED: header-only size pass, bounded read-ahead of decoded u8 batches in pinned memory, u8 H2D
    copies overlapped with the previous batch's GPU work, ToTensor / *2-1 / replicate padding on the GPU (ingest.py);
  * images of equal size are batched (--batch_size, default 32) instead of B=1;
  * multi-GPU: one process per GPU (torchrun or `bench.py`-style launch), every rank loads its own weights (no DDP
    broadcast, compress.py:242), takes a contiguous shard of the sorted file list, and the CLIP vectors are
    all-gathered with RCCL instead of going through *.npy files on a shared file system (compress.py:295-306);
    a rank that fails still reaches the collective with an error flag, so its peers exit instead of hanging;
  * without --ckpt_path / --clip_ckpt the deterministic synthetic weights are used (there is no
    checkpoint offline); a real checkpoint with the reference's state_dict keys loads unchanged.
The last line on stdout of rank 0 is a JSON record with the CLI-level rate (files -> .c2df, decode and writes included).
"""
import argparse
import json
import os
import sys
import time
from glob import glob

import numpy as np
import torch

from .entropy.compression_model import get_padding_size

torch.set_grad_enabled(False)


def load_image(path):
    """one image as the reference's Test_Dataset yields it (compress.py:158-165): CHW fp32 in [-1,1] on the host.
    The batch driver below does not use it (it ingests u8 and converts on the GPU); search.py query-image does."""
    from PIL import Image
    a = np.array(Image.open(path).convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0) * 2.0 - 1.0


def load_state(path, spec_fn, cfg, seed):
    from . import weights as W
    if path:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
        if "visual.conv1.weight" in sd or "token_embedding.weight" in sd:   # a bare open_clip CLIP state_dict
            sd = {f"clip.{k}": v for k, v in sd.items()}
        return sd
    return W.synth_weights(spec_fn(cfg), seed=seed)


def _write_outputs(c2df_path, blob, npy_path, vec):
    """the two per-image files of compress.py:282-286; runs on the writer threads (file I/O releases the GIL)"""
    with open(c2df_path, "wb") as f:
        f.write(blob)
    np.save(npy_path, vec)


def stem_of(path):
    return os.path.splitext(os.path.basename(path))[0]


def unique_stems(files):
    """Inputs that share a stem (a.jpg, a.png) write the SAME <stem>.c2df / <stem>.npy.  The reference's sequential loop leaves the
    later file's outputs on disk for both (compress.py:248-291); a batched, multi-threaded, sharded driver must not leave that to
    the order batches or writer threads happen to finish in, so only the sorted-LAST file of a stem is encoded at all: the same
    final files and index entry, decided up front.  -> the sorted list without the shadowed files."""
    last = {}
    for i, p in enumerate(files):
        last[stem_of(p)] = i
    return [p for i, p in enumerate(files) if last[stem_of(p)] == i]


def assemble_index(files, vecs, bit_dir, index_dir, dim):
    """Rank-0 index assembly (compress.py:295-306).  The reference walks `sorted(glob(clip_dir/*.npy))` -- i.e. the
    UNIQUE stems ordered by the string "<stem>.npy" -- and adds a vector for every stem whose .c2df exists, with the
    doc id `os.path.join(bit_dir, "<stem>.c2df")`.  `vecs[i]` belongs to `files[i]`; two inputs that share a stem
    (a.jpg, a.png) collapse to one entry like their .npy files do (the later file in sorted order wins)."""
    from .faiss_io import FaissDB
    by_stem = {}
    for i, p in enumerate(files):
        by_stem[stem_of(p)] = i
    order = sorted(by_stem, key=lambda s: s + ".npy")
    if not order:
        return []
    db = FaissDB(index_dir, dim)
    ids = []
    for s in order:
        doc_id = os.path.join(bit_dir, f"{s}.c2df")
        if os.path.exists(doc_id):
            db.add(vecs[by_stem[s]], doc_id)
            ids.append(doc_id)
    db.persist()
    return ids


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--base_config", type=str, default=None, help="accepted for CLI compatibility (architecture is fixed)")
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--clip_ckpt", type=str, default=None, help="open_clip ViT-B-32 state_dict (visual.* keys)")
    ap.add_argument("--dataset_dir", type=str, required=True)
    ap.add_argument("--save_dir", type=str, required=True)
    ap.add_argument("--gpu_idx", type=int, default=0)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--workers", type=int, default=None, help="image decode threads (default: min(16, cores))")
    ap.add_argument("--prefetch", type=int, default=3, help="decoded batches held ahead of the GPU (bounds host memory)")
    ap.add_argument("--small", action="store_true", help="SMALL/TINY test architectures")
    args = ap.parse_args(argv)

    import torch.distributed as dist
    from . import ops
    from . import weights as W
    from .codec import ClipCodec, Codec
    from .config import CLIP_B32, CLIP_TINY, LARGE, SMALL
    from .dist import gather_vectors, shard_range
    from .filemaker import pack_c2df
    from .ingest import DeviceIngest, ShardLoader
    from .pipeline import CompressPipeline

    # launched by torchrun / bench.py-style launchers (WORLD_SIZE set, even to 1): one rank per GPU over RCCL, and every
    # collective below runs -- a 1-rank job takes the code path of an 8-rank job.  Plain `python compress.py`: no process group.
    distributed = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=os.environ.get("SGIC_DIST_BACKEND", "nccl"), init_method="env://", rank=rank, world_size=world)
        local = int(os.environ.get("LOCAL_RANK", str(args.gpu_idx)))
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
    if distributed:
        dist.barrier()

    files = unique_stems(sorted(glob(os.path.join(args.dataset_dir, "*.*"))))
    lo, hi = shard_range(len(files), rank, world)
    mine = files[lo:hi]
    vecs = np.zeros((len(mine), ccfg.embed_dim), dtype=np.float32)
    pipe = CompressPipeline(model, clipc, dev, want_unit=True)
    ingest = DeviceIngest(dev)
    clip_meta = clipc.meta(ccfg.embed_dim)
    failed = None
    host_ms = {}      # where the launching thread spends a batch (milliseconds, summed): reported in the JSON record
    gpu_evs = []      # (start, end) events around each batch's launch-stream work
    t_start = time.perf_counter()

    def write_out(job):
        """host side of one batch: .c2df container + clip vector per image (compress.py:268-291)"""
        h, batch, pad, copied = job
        copied.synchronize()
        if batch.jpeg is not None:            # decoded on the GPU: a corrupt entropy-coded segment shows up as an error code
            codes = batch.jpeg.err_host.numpy()[:len(batch.paths)]          # pinned, filled on the copy stream ahead of `copied`
            if codes.any():
                raise RuntimeError(f"corrupt JPEG data in {[p for p, c in zip(batch.paths, codes) if c]} (codes {codes[codes != 0].tolist()})")
        batch.release()                       # the pinned u8 buffer goes back to the decode threads
        pl, pr, pt, pb = pad
        for j, (i, streams) in enumerate(zip(batch.indices, pipe.finish(h))):
            stem = stem_of(mine[i])
            enc = pipe.enc_result(h, j, streams)
            enc["clip_stream"] = streams["clip_stream"]
            enc["clip_meta"] = clip_meta
            header = {"version": 2, "model_id": clip_meta["model_id"], "embed_dim": int(ccfg.embed_dim),
                      "quant_type": "u8_symmetric_-1_1", "image_hw": [int(batch.H), int(batch.W)],
                      "padding": [int(pl), int(pr), int(pt), int(pb)]}
            vecs[i] = streams["clip_unit"]
            writes.append(io_pool.submit(_write_outputs, os.path.join(bit_dir, f"{stem}.c2df"), pack_c2df(enc, header),
                                         os.path.join(clip_dir, f"{stem}.npy"), vecs[i].copy()))
        while len(writes) > 4 * args.batch_size:      # bounded: file-system stalls push back on the producer
            writes.popleft().result()

    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    io_pool, writes = ThreadPoolExecutor(max_workers=4), deque()
    loader = None
    try:
        loader = ShardLoader(mine, args.batch_size, workers=args.workers, depth=args.prefetch)
        # two-deep pipeline: the GPU works on batch k+1 while the host packs and writes batch k, and the decode threads
        # are already filling the pinned buffers of batches k+2 .. k+1+prefetch
        pending = None
        it = iter(loader)
        batch = next(it, None)
        tok = ingest.start(batch) if batch is not None else None               # decode / H2D of the first batch
        while batch is not None:
            t_a = time.perf_counter()
            nxt = next(it, None)
            t_b = time.perf_counter()
            # one batch AHEAD: the next batch's GPU JPEG decode (or H2D copy) is enqueued before this batch's kernels, so it runs under them
            ntok = ingest.start(nxt) if nxt is not None else None
            pad = get_padding_size(batch.H, batch.W, p=256)                    # compress.py:257
            x, copied = ingest.finish(tok, pad)                                # ToTensor*2-1 + replicate pad on the GPU
            t_c = time.perf_counter()
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            h = pipe.submit(x, clip_hw=(batch.H, batch.W))   # CLIP sees the UNPADDED top-left H x W region (compress.py:266)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            gpu_evs.append((ev0, ev1))
            t_d = time.perf_counter()
            if pending is not None:
                write_out(pending)
            t_e = time.perf_counter()
            for k, v in (("wait_loader", t_b - t_a), ("ingest", t_c - t_b), ("submit", t_d - t_c), ("write_out_incl_gpu_wait", t_e - t_d)):
                host_ms[k] = host_ms.get(k, 0.0) + v * 1e3
            host_ms["batches"] = host_ms.get("batches", 0) + 1
            pending = (h, batch, pad, copied)
            batch, tok = nxt, ntok
        if pending is not None:
            write_out(pending)
        while writes:
            writes.popleft().result()             # every file is on disk before the index is assembled; errors surface here
    except BaseException as e:   # noqa: BLE001 -- reported below, after the collective the peers are waiting in
        failed = e
    finally:
        if loader is not None:
            loader.close()
        io_pool.shutdown(wait=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    if len(gpu_evs) > 2:      # launch-stream time of a batch (first batch excluded: autotune / cold caches) and the idle time between batches
        host_ms["gpu_launch_stream_busy"] = sum(a.elapsed_time(b) for a, b in gpu_evs[1:]) / (len(gpu_evs) - 1) * host_ms.get("batches", 1)
        host_ms["gpu_gap_between_batches"] = sum(gpu_evs[i][1].elapsed_time(gpu_evs[i + 1][0]) for i in range(1, len(gpu_evs) - 1)) / max(1, len(gpu_evs) - 2) * host_ms.get("batches", 1)

    # every rank reaches this point, failed or not: the error flag travels first so that nobody blocks in the gather
    # (collective operands live on the device for RCCL; the gloo backend -- CPU rehearsals, several ranks sharing one card -- takes host tensors)
    cdev = dev if not distributed or dist.get_backend() == "nccl" else torch.device("cpu")
    if distributed:
        flag = torch.tensor([1.0 if failed is not None else 0.0], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) > 0:
            dist.destroy_process_group()
            if failed is not None:
                raise failed
            print(f"[rank {rank}] another rank failed; exiting without writing the index", file=sys.stderr)
            return 1
    elif failed is not None:
        raise failed
    allv = gather_vectors(torch.from_numpy(vecs).to(cdev), len(files), rank, world).cpu().numpy()
    rate = torch.tensor([len(mine) / dt if dt > 0 else 0.0], device=cdev, dtype=torch.float64)
    if distributed:
        dist.all_reduce(rate, op=dist.ReduceOp.SUM)
    if rank == 0:
        assemble_index(files, allv, bit_dir, index_dir, ccfg.embed_dim)
        print(json.dumps({"cli_images_per_s": round(float(rate.item()), 2), "images": len(files), "n_gpus": world,
                          "collectives": (dist.get_backend() if distributed else None),
                          "seconds_rank0": round(dt, 3), "batch_size": args.batch_size,
                          "host_ms_per_batch": {k: round(v / max(1, host_ms.get("batches", 1)), 2) for k, v in host_ms.items() if k != "batches"},
                          "gpu_jpeg_batches": getattr(loader, "gpu_batches", 0), "host_decoded_batches": getattr(loader, "host_batches", 0),
                          "note": "files -> .c2df: header pass, JPEG/PNG decode, H2D, encoder+entropy+CLIP, container + .npy writes"}),
              flush=True)
    ops.save_tile_cache()
    if distributed:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
