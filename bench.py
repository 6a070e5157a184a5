#!/usr/bin/env python3
"""Benchmark of the compress hot path: images/s end-to-end compress (hybrid encoder + entropy coding + CLIP)
at 256x256, batch 32 per GPU (BASELINE.json configs[1]), synthetic images and synthetic weights of the
exact production architecture (TiTok ViT-L hybrid encoder, 64-ch bottleneck, OpenCLIP ViT-B/32).

A step = one batch of 32 device-resident fp32 images -> 32 x (z_bit_stream, h_bit_stream, zstd'd CLIP code)
as host byte strings.  One process per GPU (torch.distributed / RCCL), images sharded across ranks with no
data-path collective except the all-gather of the CLIP vectors (for the FAISS index).

    python bench.py                       # 1 GPU
    python bench.py --gpus 4              # self-launching: starts 4 ranks (children of this process), rank 0 prints
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 bench.py --gpus 4
Prints ONE JSON line on rank 0.  `--gpus N` must equal WORLD_SIZE when a launcher set it (loud failure otherwise).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 366.3        # compress at 256x256 (355.3 enc + 2.16 bottleneck + 8.8 CLIP + VQ)
GFLOP_PER_IMAGE_DEC = 654.3    # decompress at 256x256
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: ~2.5 PF dense = 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
S3_MFMA_PER_MAC = 6             # csrc/gemm_split.hip: six bf16 MFMAs per fp32-equivalent multiply-add (bf16x3 split)
PEAK_S3_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / S3_MFMA_PER_MAC, 1)   # fp32-equivalent ceiling of the split GEMM on the bf16 pipe


def _is_s3(key):
    return len(key) > 5 and key[5] == "s3"


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain shell: start N ranks as CHILD processes (one per GPU, env:// rendezvous
    on 127.0.0.1) and wait for them.  This parent never touches the GPU (no HIP call, no torch.cuda call) and never
    exec()s; rank 0's JSON line goes straight to the inherited stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in alive:       # a failed rank would leave its peers blocked in the next collective
                    q.terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, clip_sd, cfg, clip_cfg, budget_s=28.0):
    """The oracle (kind "port": torch-CPU fp32 restatement + C rANS / Pillow-resample oracle) timed on the host cores,
    BASELINE.md §3 protocol on a bounded sample: B=1 loop like compress.py:248 (1 warm-up + median of 5 images) and one
    B=32 batched pass (as many images of it as the time budget allows, at least 8)."""
    import numpy as np
    import torch
    from oracle import orc
    from oracle import torch_ref as TR
    from sgic_amd.data import synth_images
    # all the cores this process may use -- its affinity mask, cut by the cgroup's CPU quota when one is set
    # (a 1-GPU box gives a job a share of the host, not the whole EPYC); SGIC_CPU_THREADS overrides.  The round-2 figure was
    # taken on 16 threads, so that leg is reported beside it whenever more are available.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = max(1, int(os.environ.get("SGIC_CPU_THREADS", str(avail))))
    torch.set_num_threads(cores)
    t = np.load(os.path.join(ROOT, "tests", "golden", "cdf_table.npz"))
    tab = orc.Table(t["cdf"], t["cdf_length"], t["offset"])
    x = synth_images(32, 256, 256, seed=99)
    mean = torch.tensor(clip_cfg.mean).view(1, 3, 1, 1)
    std = torch.tensor(clip_cfg.std).view(1, 3, 1, 1)
    S = clip_cfg.image_size

    def compress(xb):
        """xb (b,3,256,256) -> per-image byte strings, the same stages bench's GPU step runs"""
        z, h, _ = TR.encoder_forward(xb * 0.5 + 0.5, sd, cfg)
        idx = TR.vq_indices(z, sd).view(xb.shape[0], -1)
        y = TR.bottleneck_analysis(h, sd)
        sym, ind, _, _ = TR.four_part_prior_write(y, sd, cfg.force_zero_thres)
        sym, ind = sym.numpy().reshape(xb.shape[0], -1), ind.numpy().reshape(xb.shape[0], -1)
        u8 = np.stack([orc.resize_bicubic_u8(xb[b].clamp(-1, 1).mul(0.5).add(0.5).mul(255).byte().numpy(), S, S)
                       for b in range(xb.shape[0])])
        pre = (torch.from_numpy(u8).float().div(255.0) - mean) / std      # the RESAMPLED image feeds the CLIP tower
        unit = TR.clip_tower(pre, clip_sd, clip_cfg)
        out = []
        for b in range(xb.shape[0]):
            q = np.clip(np.round((unit[b].numpy() * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
            out.append((orc.pack12(idx[b].numpy().astype(np.int16)), orc.rans_encode(sym[b], ind[b], tab), q.tobytes()))
        return out

    t_start = time.perf_counter()
    legs = {}
    with torch.no_grad():
        compress(x[:1])                                       # warm-up
        t1 = []
        for b in range(1, 6):
            t0 = time.perf_counter()
            compress(x[b:b + 1])
            t1.append(time.perf_counter() - t0)
            print(f"[cpu_baseline] B=1 image {b} ({cores} threads): {t1[-1]:.3f}s", file=sys.stderr, flush=True)
        b1 = 1.0 / statistics.median(t1)
        if cores > 16:                                        # the round-2 configuration beside it
            torch.set_num_threads(16)
            compress(x[:1])
            t16 = []
            for b in range(6, 9):
                t0 = time.perf_counter()
                compress(x[b:b + 1])
                t16.append(time.perf_counter() - t0)
            legs["b1_loop_16_threads_median_of_3"] = round(1.0 / statistics.median(t16), 4)
            torch.set_num_threads(cores)
        left = budget_s - (time.perf_counter() - t_start)
        nb = int(max(8, min(32, left * b1 * 1.3)))            # batched leg sized to the remaining budget
        t0 = time.perf_counter()
        compress(x[:nb])
        tb = time.perf_counter() - t0
        print(f"[cpu_baseline] B={nb} batched: {tb:.3f}s", file=sys.stderr, flush=True)
    bb = nb / tb
    legs.update({"b1_loop_median_of_5": round(b1, 4), f"b{nb}_batched_one_pass": round(bb, 4)})
    return {"value": round(max(b1, bb), 4), "unit": "images/s", "cores": cores, "cores_available": avail,
            "host_logical_cpus": os.cpu_count(), "kind": "port", "cpu_model": _cpu_model(), "legs": legs,
            "sample": f"256x256 compress (encoder + entropy + resample + CLIP): 1 warm-up + 5 single images (median) and one "
                      f"batch of {nb}; value = the faster leg; torch-CPU fp32 restatement + C rANS oracle (oracle/), {cores} threads"}


# ------------------------------------------------------------------------------------------------ CPU rehearsal
def rehearse_cpu(args, world, rank):
    """--rehearse-cpu: the launcher / rendezvous / shard / all-gather / max-over-ranks protocol of the real bench on
    the gloo backend with NO GPU work (tests/test_bench_launcher.py; also usable on a 1-GPU box to rehearse N ranks).
    Prints the same JSON skeleton with value = 0 and data = "rehearsal"; never a benchmark result."""
    import torch
    import torch.distributed as dist
    if os.environ.get("SGIC_BENCH_FAIL_RANK") == str(rank):     # test hook: a rank that dies before the rendezvous
        return 3
    if world > 1:
        dist.init_process_group(backend="gloo", init_method="env://")
    B, D = args.batch, 512
    gathered = torch.empty(world * B, D)
    g = torch.Generator().manual_seed(rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    tg = 0.0
    for _ in range(args.steps):
        unit = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
        tq = time.perf_counter()
        if world > 1:
            dist.all_gather_into_tensor(gathered, unit)
            assert torch.equal(gathered[rank * B:(rank + 1) * B], unit)
        tg += time.perf_counter() - tq
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    if world > 1:
        tt = torch.tensor([dt, tg], dtype=torch.float64)
        allt = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [float(a[0]) for a in allt]
        dt = max(per_rank)
    if rank == 0:
        print(json.dumps({"metric": "launcher rehearsal (no GPU work)", "value": 0.0, "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "rehearsal",
                          "config": {"workload": "rendezvous + all-gather protocol only", "global_batch": world * B},
                          "per_rank_seconds": [round(t, 4) for t in per_rank],
                          "allgather_ms_per_step": round(tg / max(1, args.steps) * 1e3, 3)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------ roofline helpers
def by_shape_table(prof, ops, dev, top=10, pmc_shapes="round3_pmc_gemm_by_shape.json"):
    """per-shape roofline of the GEMM launches from the per-launch event pairs of the timed region; `frac` is against the
    peak of the kernel the shape ran on (split GEMM: bf16 peak / 6; fp32 MFMA kernel: 157.3)"""
    agg = {}
    for fl, ms, key in prof:
        v = agg.setdefault(key, [0.0, 0.0, 0])
        v[0] += fl
        v[1] += ms
        v[2] += 1
    total_ms = sum(v[1] for v in agg.values()) or 1.0
    pmc = {}
    try:   # HBM-side counter bytes per launch and shape, from the committed rocprofv3 --pmc passes (tools/pmc_by_shape.py)
        pmc = {tuple(r["shape"]): r for r in json.load(open(os.path.join(ROOT, "profiles", pmc_shapes)))["shapes"]}
    except Exception:
        pass
    rows = []
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        if key and key[0] == "batched":
            _, nb, M, N, K, res, act = key
        else:
            nb = 1
            M, N, K, res, act = key[:5]
        s3 = _is_s3(key)
        # algorithmic bytes: fp32 operands and result; the split kernel reads its operands as 3 bf16 planes (6 B per element)
        conv = len(key) > 6 and key[6] == "conv"     # implicit GEMM: the input is read once (M x Cin), not as the M x 9 Cin im2col matrix
        algo = (6.0 if s3 else 4.0) * nb * (M * (K // 9 if conv else K) + N * K) + 4.0 * nb * M * N * (2 if res else 1)
        peak = PEAK_S3_TFLOPS if s3 else PEAK_FP32_MFMA_TFLOPS
        r = {"M": M, "N": N, "K": K, "batch": nb, "residual": bool(res), "act": act, "calls": v[2], "kernel": ("split3" if s3 else "f32") + (" conv3x3" if conv else ""),
             "tflops": round(v[0] / v[1] / 1e9, 1), "frac": round(v[0] / v[1] / 1e9 / peak, 3),
             "share_of_gemm_time": round(v[1] / total_ms, 4), "avg_us": round(v[1] / v[2] * 1e3, 1),
             "algorithmic_MB": round(algo / 1e6, 1), "tile_mode": ops.tile_of(key, dev)}
        p = pmc.get((M, N, K))
        if p and p.get("kernel", "f32") == r["kernel"].split()[0]:
            r["counter_MB"] = p["counter_MB"]
        rows.append(r)
    return rows


def roofline_block(prof, ops, dev, dt, steps, flops_per_step, pmc_file, pmc_shapes=None):
    """dominant kernel = the GEMM family that holds most of the GEMM time: the bf16x3 split GEMM (priced against the bf16 dense
    MFMA peak / 6 MFMAs per multiply-add) or the fp32-input MFMA GEMM (157.3)"""
    s3 = [p for p in prof if _is_s3(p[2])]
    f32 = [p for p in prof if not _is_s3(p[2])]
    dom_s3 = sum(p[1] for p in s3) > sum(p[1] for p in f32)
    dom, other = (s3, f32) if dom_s3 else (f32, s3)
    peak = PEAK_S3_TFLOPS if dom_s3 else PEAK_FP32_MFMA_TFLOPS
    gemm_flops = sum(p[0] for p in dom)
    gemm_ms = sum(p[1] for p in dom)
    all_ms = sum(p[1] for p in prof)
    n_launch = len(dom)
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    traffic = None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        if pj.get("kernel", "gemm_f32_kernel") == ("gemm_split3_kernel" if dom_s3 else "gemm_f32_kernel"):
            traffic = int((pj["hbm_fetch_MB_per_launch_x2_corrected"] + pj["hbm_write_MB_per_launch"]) * 1e6)
    except Exception:
        pass
    blk = {"bound": "mfma", "kernel": "gemm_split3_dma_kernel (+ gemm_split3_kernel / gemm_split3_ring_kernel for the small tiles)" if dom_s3 else "gemm_f32_kernel", "achieved": round(achieved, 2), "peak": peak,
           "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
           "traffic_unit": f"bytes per launch (rocprofv3 PMC, profiles/{pmc_file})",
           "launches_per_step": n_launch // max(1, steps),
           "avg_launch_us": round(gemm_ms * 1e3 / max(1, n_launch), 2),
           "gflop_per_launch": round(gemm_flops / max(1, n_launch) / 1e9, 3),
           "gemm_share_of_step": round(all_ms / (dt * 1e3), 4),
           "end_to_end_frac": round(flops_per_step * steps / dt / 1e12 / peak, 4),
           "by_shape": by_shape_table(prof, ops, dev, pmc_shapes=pmc_shapes or pmc_file.replace("_summary", "_by_shape"))}
    if dom_s3:
        blk["peak_note"] = (f"fp32-equivalent ceiling of the bf16x3 split GEMM: bf16 dense MFMA peak {PEAK_BF16_MFMA_TFLOPS} TFLOP/s / "
                            f"{S3_MFMA_PER_MAC} bf16 MFMAs per multiply-add; `achieved` counts ALGORITHMIC fp32 flops (2 M N K)")
        blk["bf16_mfma_tflops"] = round(achieved * S3_MFMA_PER_MAC, 1)
        blk["frac_of_fp32_mfma_peak"] = round(achieved / PEAK_FP32_MFMA_TFLOPS, 4)
    if other:
        oms = sum(p[1] for p in other)
        blk["other_gemm_kernel"] = {"kernel": "gemm_f32_kernel" if dom_s3 else "gemm_split3_kernel", "launches_per_step": len(other) // max(1, steps),
                                    "achieved": round(sum(p[0] for p in other) / (oms * 1e-3) / 1e12, 2) if oms > 0 else 0.0,
                                    "peak": PEAK_FP32_MFMA_TFLOPS if dom_s3 else PEAK_S3_TFLOPS, "share_of_gemm_time": round(oms / all_ms, 4) if all_ms else 0.0}
    return blk


# ------------------------------------------------------------------------------------------------ the bench
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--small", action="store_true", help="debug: SMALL/TINY configs (not a valid bench)")
    ap.add_argument("--mode", choices=["compress", "decompress"], default="compress",
                    help="compress = the headline metric (BASELINE.json configs[1]); decompress = configs[2]/[4] as the primary line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary decompress block (configs[2]) of the N=1 line")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--h2d", action="store_true", help="also copy the input batch host->device inside every step "
                    "(PCIe-inclusive rate for DESIGN.md; never the headline value)")
    ap.add_argument("--rehearse-cpu", action="store_true", help="launcher/rendezvous/all-gather protocol on gloo, no GPU work")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])     # before anything here touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
              f"mislabelled n_gpus", file=sys.stderr)
        return 2
    if args.rehearse_cpu:
        return rehearse_cpu(args, world, rank)

    import torch
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
    want_dec = args.mode == "decompress" or (world == 1 and not args.no_secondary)
    spec = W.full_spec(cfg) if want_dec else W.encoder_spec(cfg) + W.codec_misc_spec(cfg) + W.bottleneck_spec(cfg)
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

    gather_events = []

    def gather(unit):
        if world > 1:   # CLIP vectors for the FAISS index (RCCL over xGMI): the one collective of the path
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_gather_into_tensor(gathered, unit)
            e1.record()
            gather_events.append((e0, e1))

    pipe = CompressPipeline(codec, clipc, dev, on_unit=gather)
    x_host = x.cpu().pin_memory() if args.h2d else None

    def enqueue():
        xin = x
        if args.h2d:
            xin = torch.empty_like(x)
            xin.copy_(x_host, non_blocking=True)
        return pipe.submit(xin)

    def finalize(h):
        return [(d["z_bit_stream"], d["h_bit_stream"], d["clip_stream"]) for d in pipe.finish(h)]

    class _Pipe:
        prev, last = None, None

    def step():
        cur = enqueue()
        if _Pipe.prev is not None:
            _Pipe.last = finalize(_Pipe.prev)
        _Pipe.prev = cur
        return _Pipe.last

    def drain():
        if _Pipe.prev is not None:
            _Pipe.last = finalize(_Pipe.prev)
            _Pipe.prev = None
        return _Pipe.last

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def dec_step_factory():
        encs = codec.encode_batch(x)          # outside the timed region: the bitstreams to decode

        def dstep():   # one batch of .c2df payloads -> reconstructed pixels (device-resident fp32)
            x_hat = codec.decode_batch(encs)
            return [(b"", b"", b"")] * B if x_hat is not None else None
        return dstep

    if args.mode == "decompress":
        step = dec_step_factory()   # noqa: F811
        drain = lambda: None        # noqa: E731

    # one untimed priming step (independent of --warmup): first sight of every GEMM / attention shape consults the
    # per-shape tile cache or runs the autotuner, and lazily-built tables (Swin row maps, CLIP resize coefficients,
    # prior cache) fill
    step()
    drain()
    out = None
    for _ in range(args.warmup):
        out = step()
    out = drain() or out

    # profile window: closes pending in-context tile races (no tuning inside the timed region) and times every GEMM /
    # conv launch with its own dispatch-level event pair (hipExtLaunchKernel, no extra packets on the stream)
    gather_events.clear()
    ops.profile_begin()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    out = drain() or out      # the last step's host work is inside the timed region too
    sync()
    dt_local = dt = time.perf_counter() - t0
    prof = ops.profile_end()      # [(flops, ms, shape key)] per launch of the timed region
    if rank == 0 and os.environ.get("SGIC_BENCH_SHAPES"):   # launch-ordered shape list (tools/pmc_by_shape.py joins it with PMC rows)
        tiles = list(ops.PROFILE_TILES) if len(ops.PROFILE_TILES) == len(prof) else [ops.tile_of(k, dev) for _, _, k in prof]   # the mode each launch really took
        json.dump([[list(k) if isinstance(k, tuple) else k, fl, ms, t] for (fl, ms, k), t in zip(prof, tiles)], open(os.environ["SGIC_BENCH_SHAPES"], "w"))
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_events)
    per_rank = [(dt_local, gather_ms)]
    if world > 1:
        tt = torch.tensor([dt_local, gather_ms], device=dev, dtype=torch.float64)
        allt = torch.empty(world, 2, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allt, tt)
        per_rank = [(float(a), float(b)) for a, b in allt.cpu().tolist()]
        dt = max(p[0] for p in per_rank)     # MAX over ranks

    if rank == 0:
        total_bytes = sum(len(a) + len(b) + len(c) for a, b, c in out)
        dec_primary = args.mode == "decompress"
        gflop = (GFLOP_PER_IMAGE_DEC if dec_primary else GFLOP_PER_IMAGE) * (S / 256.0) ** 2
        res = {
            "metric": ("images/sec end-to-end compress (enc+entropy+CLIP) at 256x256" if not dec_primary else
                       f"images/sec decompress (entropy decode + hybrid decoder + generative decoder) at {S}x{S}"),
            "value": round(world * B * args.steps / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": ("f32 results; GEMMs as bf16x3 split on the bf16 matrix pipe (3 bf16 pieces per fp32 operand, 6 MFMAs per multiply-add, "
                           "fp32 accumulate; error vs fp64 <= the fp32 fmaf chain's, tests/test_gpu_split3.py) -- every large GEMM / 3x3 convolution and the two "
                           "products of attention (S = Q K^T, O = P V); softmax, norms, the bottleneck's small networks and the entropy stages in f32"
                           if ops.PRECISION == "split3" else "f32 throughout (fp32-input MFMA GEMMs)"),
            "config": {"workload": (f"configs[1]: batch={B} {S}x{S} encoder+entropy+CLIP compress per GPU, "
                                    f"{'SMALL debug model' if args.small else 'TiTok ViT-L hybrid encoder + ViT-B/32 CLIP'}, synthetic weights")
                       if not dec_primary else
                       (f"configs[2]: batch={B} {S}x{S} decompress per GPU (rANS decode chain + hybrid decoder + FeatMerge + "
                        f"taming VQGAN decoder), {'SMALL debug model' if args.small else 'production architecture'}, synthetic weights"),
                       "global_batch": world * B, "bytes_per_image": round(total_bytes / B, 1)},
            "per_rank_images_per_s": [round(B * args.steps / p[0], 2) for p in per_rank],
            "allgather_ms_per_step": [round(p[1] / args.steps, 4) for p in per_rank],
            "roofline": roofline_block(prof, ops, dev, dt, args.steps, gflop * 1e9 * B,
                                       "round3_pmc_decompress_gemm_summary.json" if dec_primary else "round3_pmc_gemm_summary.json"),
        }
        if world == 1 and not dec_primary and not args.no_secondary:
            # secondary line (configs[2]): the decompress path on the same process, a few steps, its own profile window
            dstep = dec_step_factory()
            dstep()
            dstep()
            ops.profile_begin()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.secondary_steps):
                dstep()
            torch.cuda.synchronize()
            ddt = time.perf_counter() - t0
            dprof = ops.profile_end()
            res["secondary"] = {"decompress": {
                "metric": f"images/sec decompress (rANS decode chain + hybrid decoder + FeatMerge + taming VQGAN decoder) at {S}x{S}",
                "workload": f"configs[2]: batch={B} {S}x{S} per GPU, production architecture, synthetic weights",
                "value": round(B * args.secondary_steps / ddt, 3), "unit": "images/s", "steps": args.secondary_steps,
                "ms_per_step": round(ddt / args.secondary_steps * 1e3, 3),
                "roofline": roofline_block(dprof, ops, dev, ddt, args.secondary_steps,
                                           GFLOP_PER_IMAGE_DEC * (S / 256.0) ** 2 * 1e9 * B,
                                           "round3_pmc_decompress_gemm_summary.json")}}
        if world == 1 and not dec_primary and not args.no_secondary and ops.PRECISION == "split3":
            # the same compress step with every GEMM on the exact-fp32 MFMA kernel (SGIC_GEMM=f32), for reference
            ops.set_precision("f32")
            try:
                for _ in range(2):
                    step()
                drain()
                ops.profile_begin()
                sync()
                t0 = time.perf_counter()
                for _ in range(args.secondary_steps):
                    step()
                drain()
                sync()
                fdt = time.perf_counter() - t0
                fprof = ops.profile_end()
            finally:
                ops.set_precision("split3")
            res["secondary"]["compress_f32_mfma"] = {
                "metric": "images/sec end-to-end compress at 256x256 with SGIC_GEMM=f32 (every GEMM on v_mfma_f32_32x32x2_f32)",
                "value": round(B * args.secondary_steps / fdt, 3), "unit": "images/s", "steps": args.secondary_steps,
                "ms_per_step": round(fdt / args.secondary_steps * 1e3, 3),
                "roofline": roofline_block(fprof, ops, dev, fdt, args.secondary_steps, gflop * 1e9 * B, "round3_pmc_f32_gemm_summary.json",
                                           "round3_pmc_f32_gemm_by_shape.json")}
        if world == 1 and not args.small:
            # live check of the claim behind `arithmetic`: rms error of a K = 4096 GEMM against an fp64 product, split GEMM vs the
            # exact-fp32 MFMA GEMM on the same operands (measurement only: the fp64 product is a torch matmul, not the product path)
            gg = torch.Generator(device=dev).manual_seed(7)
            ta = torch.randn(512, 4096, device=dev, generator=gg)
            tw = torch.randn(512, 4096, device=dev, generator=gg) * 0.03
            ref = ta.double() @ tw.double().T
            rms = float(ref.pow(2).mean().sqrt())
            e3 = float((ops.gemm(ta, tw, precision="split3").double() - ref).pow(2).mean().sqrt()) / rms
            e1 = float((ops.gemm(ta, tw, precision="f32").double() - ref).pow(2).mean().sqrt()) / rms
            res["arithmetic_check"] = {"what": "rms error / rms(C) of a 512x512x4096 GEMM vs an fp64 product, same operands",
                                       "split3_bf16x3": float(f"{e3:.3e}"), "fp32_mfma_chain": float(f"{e1:.3e}")}
        if world == 1 and not args.no_cpu_baseline and not args.small and not dec_primary:
            res["cpu_baseline"] = cpu_baseline(sd, clip_sd, cfg, clip_cfg)
        print(json.dumps(res), flush=True)
    ops.save_tile_cache()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
