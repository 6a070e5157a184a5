#!/usr/bin/env python3
"""Benchmark of the compress hot path: images/s end-to-end compress (hybrid encoder + entropy coding + CLIP)
at 256x256, batch 32 per GPU (BASELINE.json configs[1]), synthetic images and synthetic weights of the
exact production architecture (TiTok ViT-L hybrid encoder, 64-ch bottleneck, OpenCLIP ViT-B/32).

A step = one batch of 32 device-resident fp32 images -> 32 x (z_bit_stream, h_bit_stream, zstd'd CLIP code)
as host byte strings.  One process per GPU (torch.distributed / RCCL), images sharded across ranks with no
data-path collective except the all-gather of the CLIP vectors (for the FAISS index).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

GFLOP_PER_IMAGE = 366.3        # compress at 256x256 (355.3 enc + 2.16 bottleneck + 8.8 CLIP + VQ)
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def cpu_baseline(sd, clip_sd, cfg, clip_cfg, n_images):
    """The oracle (kind "port"): torch-CPU fp32 restatement + C rANS, B=1 loop like compress.py:248."""
    from oracle import orc
    from oracle import torch_ref as TR
    from sgic_amd.data import synth_images
    # threads actually usable: the cgroup/affinity share of this process, not the host's core count
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("SGIC_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    t = np.load(os.path.join(ROOT, "tests", "golden", "cdf_table.npz"))
    tab = orc.Table(t["cdf"], t["cdf_length"], t["offset"])
    x = synth_images(n_images, 256, 256, seed=99)
    pre = torch.randn(1, 3, 224, 224)
    with torch.no_grad():
        t0 = time.perf_counter()
        done = 0
        for b in range(n_images):
            if b > 0 and time.perf_counter() - t0 > 25.0:   # bounded sample: ~10-30 s of CPU work
                break
            print(f"[cpu_baseline] image {b} t={time.perf_counter() - t0:.1f}s", file=sys.stderr, flush=True)
            xb = x[b:b + 1]
            z, h, _ = TR.encoder_forward(xb * 0.5 + 0.5, sd, cfg)
            idx = TR.vq_indices(z, sd)
            orc.pack12(idx.numpy().astype(np.int16))
            y = TR.bottleneck_analysis(h, sd)
            sym, ind, _, _ = TR.four_part_prior_write(y, sd, cfg.force_zero_thres)
            orc.rans_encode(sym.numpy(), ind.numpy(), tab)
            u8 = orc.resize_bicubic_u8((xb[0].clamp(-1, 1).mul(0.5).add(0.5)).mul(255).byte().numpy(), 224, 224)
            TR.clip_tower(pre, clip_sd, clip_cfg)
            done += 1
        dt = time.perf_counter() - t0
    n_images = done
    return {"value": round(n_images / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_images} images 256x256, B=1 loop, torch-CPU fp32 restatement + C rANS oracle (oracle/)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--small", action="store_true", help="debug: SMALL/TINY configs (not a valid bench)")
    ap.add_argument("--mode", choices=["compress", "decompress"], default="compress",
                    help="compress = the headline metric (BASELINE.json configs[1]); decompress = configs[2]/[4] secondary line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="also copy the input batch host->device inside every step "
                    "(PCIe-inclusive rate for DESIGN.md; never the headline value)")
    ap.add_argument("--cpu-images", type=int, default=32, help="CPU-baseline sample: images of the same workload, B=1 loop, cut off after 25 s")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import sgic_amd  # noqa: F401
    from sgic_amd import ops
    from sgic_amd import weights as W
    from sgic_amd.codec import ClipCodec, Codec
    from sgic_amd.config import CLIP_B32, CLIP_TINY, LARGE, SMALL
    from sgic_amd.data import synth_images

    cfg, clip_cfg = (SMALL, CLIP_TINY) if args.small else (LARGE, CLIP_B32)
    spec = W.full_spec(cfg) if args.mode == "decompress" else W.encoder_spec(cfg) + W.codec_misc_spec(cfg) + W.bottleneck_spec(cfg)
    sd = W.synth_weights(spec, seed=1234)
    clip_sd = W.synth_weights(W.clip_spec(clip_cfg), seed=4321)
    codec = Codec(sd, cfg, dev)
    codec.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    codec.hybrid_codec.quantize_feat.update(force=True)
    clipc = ClipCodec(clip_sd, clip_cfg, dev)

    B, S = args.batch, args.size
    x = synth_images(B, S, S, seed=1000 + rank).to(dev)   # inputs resident in HBM before the timed region
    gathered = torch.empty(world * B, clip_cfg.embed_dim, device=dev) if world > 1 else None

    # Two-deep software pipeline over steps (sgic_amd.pipeline.CompressPipeline, the same object compress.py drives):
    # the GPU work of step i (encoder, entropy coding, CLIP, async D2H into pinned buffers) is enqueued before the
    # host finishes step i-1 (slice the streams, zstd the CLIP codes), so the host-side byte work hides under the GPU.
    # Every step still ends as 32 x (z, h, clip) host byte strings inside the timed region.  The rANS kernel (serial,
    # ~1 ms) runs on a side HIP stream under the CLIP tower.
    from sgic_amd.pipeline import CompressPipeline

    def gather(unit):
        if world > 1:
            dist.all_gather_into_tensor(gathered, unit)   # CLIP vectors for the FAISS index (RCCL over xGMI)

    pipe = CompressPipeline(codec, clipc, dev, on_unit=gather)
    x_host = x.cpu().pin_memory() if args.h2d else None

    def enqueue(slot):
        xin = x
        if args.h2d:
            xin = torch.empty_like(x)
            xin.copy_(x_host, non_blocking=True)
        return pipe.submit(xin)

    def finalize(h):
        return [(d["z_bit_stream"], d["h_bit_stream"], d["clip_stream"]) for d in pipe.finish(h)]

    class _Pipe:
        prev, i, last = None, 0, None

    def step():
        cur = enqueue(_Pipe.i & 1)
        _Pipe.i += 1
        if _Pipe.prev is not None:
            _Pipe.last = finalize(_Pipe.prev)
        _Pipe.prev = cur
        return _Pipe.last

    def drain():
        if _Pipe.prev is not None:
            _Pipe.last = finalize(_Pipe.prev)
            _Pipe.prev = None
        return _Pipe.last

    if args.mode == "decompress":
        encs = codec.encode_batch(x)          # outside the timed region: the bitstreams to decode

        def step():   # noqa: F811  -- one batch of .c2df payloads -> reconstructed pixels (device-resident fp32)
            x_hat = codec.decode_batch(encs)
            return [(b"", b"", b"")] * B if x_hat is not None else None

    # one untimed priming step (independent of --warmup): first sight of every GEMM / attention shape runs the
    # per-shape tile autotuner, and lazily-built tables (Swin row maps, CLIP resize coefficients, prior cache) fill
    if args.mode == "compress":
        step()
        drain()
        for _ in range(args.warmup):
            step()
        out = drain()
    else:
        drain = lambda: None   # noqa: E731
        step()
        for _ in range(args.warmup):
            out = step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # profile window: closes pending in-context tile races (no tuning inside the timed region) and times every GEMM /
    # conv launch with its own dispatch-level event pair (hipExtLaunchKernel, no extra packets on the stream)
    ops.profile_begin()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    out = drain() or out      # the last step's host work is inside the timed region too
    sync()
    dt = time.perf_counter() - t0
    prof = ops.profile_end()      # [(flops, ms, shape key)] per launch of the timed region
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        gemm_flops = sum(p[0] for p in prof)
        gemm_ms = sum(p[1] for p in prof)
        n_launch = len(prof)
        if os.environ.get("SGIC_BENCH_SHAPES"):   # per-shape breakdown of the dominant kernel (stderr)
            agg = {}
            for fl, t_, key in prof:
                v = agg.setdefault(key, [0.0, 0.0, 0])
                v[0] += fl; v[1] += t_; v[2] += 1
            for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                tile = ops._TILE.get(key + (str(dev),))
                print(f"[gemm] M,N,K,res,act={key} tile={tile} calls={v[2]} total_ms={v[1]:.2f} share={v[1]/gemm_ms:.3f} "
                      f"TF={v[0]/v[1]/1e9:.1f}", file=sys.stderr)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        total_bytes = sum(len(a) + len(b) + len(c) for a, b, c in out)
        # HBM-side bytes per GEMM launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2-corrected + WRITE_SIZE,
        # MI355X_MICROARCH.md) committed under profiles/; PMC counters cannot be read live from inside this process.
        traffic = None
        pmc_file = "round1_pmc_gemm_summary.json" if args.mode == "compress" else "round1_pmc_decompress_gemm_summary.json"
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            traffic = int((pj["hbm_fetch_MB_per_launch_x2_corrected"] + pj["hbm_write_MB_per_launch"]) * 1e6)
        except Exception:
            pass
        res = {
            "metric": ("images/sec end-to-end compress (enc+entropy+CLIP) at 256x256" if args.mode == "compress" else
                       f"images/sec decompress (entropy decode + hybrid decoder + generative decoder) at {S}x{S}"),
            "value": round(world * B * args.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"configs[1]: batch={B} {S}x{S} encoder+entropy+CLIP compress per GPU, "
                                    f"{'SMALL debug model' if args.small else 'TiTok ViT-L hybrid encoder + ViT-B/32 CLIP'}, synthetic weights")
                       if args.mode == "compress" else
                       (f"configs[2]: batch={B} {S}x{S} decompress per GPU (rANS decode chain + hybrid decoder + FeatMerge + "
                        f"taming VQGAN decoder), {'SMALL debug model' if args.small else 'production architecture'}, synthetic weights"),
                       "global_batch": world * B, "bytes_per_image": round(total_bytes / B, 1)},
            "roofline": {"bound": "mfma", "kernel": "gemm_f32_kernel", "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": f"bytes per launch (rocprofv3 PMC, profiles/{pmc_file})",
                         "launches_per_step": n_launch // max(1, args.steps),
                         "avg_launch_us": round(gemm_ms * 1e3 / max(1, n_launch), 2),
                         "gflop_per_launch": round(gemm_flops / max(1, n_launch) / 1e9, 3),
                         "gemm_share_of_step": round(gemm_ms / (dt * 1e3), 4),
                         "end_to_end_frac": round((GFLOP_PER_IMAGE if args.mode == "compress" else 654.3) * (S / 256.0) ** 2 * B *
                                                  args.steps / dt / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline and not args.small and args.mode == "compress":
            res["cpu_baseline"] = cpu_baseline(sd, clip_sd, cfg, clip_cfg, args.cpu_images)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
