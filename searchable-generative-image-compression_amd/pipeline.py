"""Two-deep software pipeline over batches for the compress hot path (what `compress.py`'s per-image loop,
compress.py:248-291, becomes when a batch of images is in flight on one MI355X).

`submit(x)` enqueues ALL device work of one batch -- hybrid encoder, VQ + 12-bit packing, analysis transform, 4-step
prior, rANS coding on a side HIP stream underneath the CLIP tower, CLIP preprocessing + tower + u8 codes -- and the
asynchronous device->host copies into pinned buffers, then returns at once.  `finish(handle)` waits for that batch's
copies and turns them into host byte strings (slice the end-aligned rANS streams, zstd-19 the 512-byte CLIP codes).
Calling `submit(batch i+1)` before `finish(batch i)` hides the host-side byte work under the GPU; `bench.py` times
exactly this, `compress.py` writes the .c2df files from it.
"""
import numpy as np
import torch


class CompressPipeline:
    def __init__(self, codec, clipc, device, want_unit=False, on_unit=None):
        """on_unit(unit (B,D) device tensor): optional hook run on the launch stream right after the CLIP tower
        (bench.py / compress.py use it for the RCCL all-gather of the CLIP vectors)."""
        self.codec, self.clipc, self.device = codec, clipc, torch.device(device)
        self.want_unit, self.on_unit = want_unit, on_unit
        self.side = torch.cuda.Stream(device=self.device)
        self._pinned = {}
        self._n = 0

    def _buffers(self, B, cap, nz, D):
        """pinned host buffers, two alternating slots per batch geometry"""
        slot = self._n & 1
        self._n += 1
        key = (slot, B, cap, nz, D)
        if key not in self._pinned:
            pin = lambda *s, dtype=torch.uint8: torch.empty(*s, dtype=dtype).pin_memory()
            self._pinned[key] = dict(hs=pin(B, cap), meta=pin(3, B, dtype=torch.int32), zs=pin(B, nz), q=pin(B, D),
                                     unit=pin(B, D, dtype=torch.float32) if self.want_unit else None, busy=False)
        p = self._pinned[key]
        if p["busy"]:
            raise RuntimeError("CompressPipeline is two deep: finish() the batch submitted two calls ago before submit()")
        p["busy"] = True
        return p

    def submit(self, x, clip_hw=None):
        """x (B,3,H,W) fp32 in [-1,1] on the device, H and W multiples of 256 (already padded, compress.py:258-261).
        clip_hw = (h, w): the top-left h x w region is the real image the CLIP tower must see (compress.py:266)."""
        B, _, H, W = x.shape
        r = self.codec.encode_device(x, side_stream=self.side)
        unit, q = self.clipc.batch_to_codes(x, *(clip_hw or (None, None)))
        if self.on_unit is not None:
            self.on_unit(unit)
        torch.cuda.current_stream().wait_stream(self.side)
        p = self._buffers(B, r["hs"].shape[1], r["zs"].shape[1], q.shape[1])
        p["hs"].copy_(r["hs"], non_blocking=True)
        p["meta"].copy_(r["hmeta"], non_blocking=True)
        p["zs"].copy_(r["zs"], non_blocking=True)
        p["q"].copy_(q, non_blocking=True)
        if self.want_unit:
            p["unit"].copy_(unit, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return dict(p=p, ev=ev, r=r, B=B, H=H, W=W)

    def finish(self, h):
        """-> list of B dicts {z_bit_stream, h_bit_stream, clip_stream[, clip_unit]} (host bytes)"""
        p, B = h["p"], h["B"]
        h["ev"].synchronize()
        p["busy"] = False
        from .bottleneck import slice_streams
        bn, r = self.codec.bottleneck, h["r"]
        # an image whose rANS slot overflowed is re-encoded on its own (its symbols are still on the device), not the batch
        h_streams = slice_streams(p["hs"].numpy(), p["meta"].numpy(), retry=(bn.tables.handles[bn.group], r["sym"], r["idx"], r["n"]))
        zs, qh = p["zs"].numpy(), p["q"].numpy()
        out = []
        for b in range(B):
            d = dict(z_bit_stream=zs[b].tobytes(), h_bit_stream=h_streams[b], clip_stream=self.clipc.compress_codes(qh[b]))
            if self.want_unit:
                d["clip_unit"] = p["unit"][b].numpy().copy()
            out.append(d)
        return out

    def enc_result(self, h, b, streams):
        """the reference's encode_only dict (codec_sq_fixbpp.py:870-878) for image b of a finished batch"""
        cfg, r = self.codec.cfg, h["r"]
        nH, nW = r["stack"]
        hh, ww = r["feat_hw"]
        return {"z_bit_stream": streams["z_bit_stream"], "h_bit_stream": streams["h_bit_stream"], "img_shape": (h["H"], h["W"]),
                "feat_shape": torch.Size([1, cfg.feat_dim, hh, ww]), "stack_shape": (nH, nW), "token_length": r["ntok"],
                "z_indices_shape": torch.Size([nH * nW, cfg.token_size, 1, cfg.num_latent_tokens])}
