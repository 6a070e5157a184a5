"""Codec facade: the drop-in for `models.codec_sq_fixbpp.Codec` on the compress path
(reference: models/codec_sq_fixbpp.py:442-508,841-878) plus `ClipCodec` (compress.py:58-86).

`Codec.encode_only(x)` keeps the reference contract (one padded image in, the enc_result dict out);
`Codec.encode_batch(x)` is the MI355X-native entry: B images per call, everything device-resident until
the finished byte strings are copied out once."""
import numpy as np
import torch

from . import ops
from .bottleneck import BottleneckHIP
from .clip import ClipHIP
from .config import CLIP_B32, LARGE, ClipConfig, CodecConfig
from .encoder import HybridEncoderHIP
from .zstd import Compressor


class _QuantizeFeatProxy:
    """`model.hybrid_codec.quantize_feat.force_zero_thres = 0.12; .update(force=True)` (compress.py:238-239)"""

    def __init__(self, bott):
        self._b = bott

    @property
    def force_zero_thres(self):
        return self._b.force_zero_thres

    @force_zero_thres.setter
    def force_zero_thres(self, v):
        self._b.force_zero_thres = v

    def update(self, force=False):
        self._b.update(force=force)


class _HybridCodecProxy:
    def __init__(self, bott):
        self.quantize_feat = _QuantizeFeatProxy(bott)


class Codec:
    def __init__(self, state_dict, cfg: CodecConfig = LARGE, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("sgic_amd.Codec needs an MI355X GPU (HIP-only hot path, no CPU fallback)")
        self.cfg = cfg
        self.device = torch.device(device)
        self.encoder = HybridEncoderHIP(state_dict, cfg, self.device)
        self.bottleneck = BottleneckHIP(state_dict, cfg, self.device)
        self.codebook = state_dict["hybrid_codec.quantize.embedding.weight"].to(self.device).float().contiguous()
        self.hybrid_codec = _HybridCodecProxy(self.bottleneck)
        self._sd = state_dict
        self._dec = None

    def to(self, device):
        return self

    def eval(self):
        return self

    def encode_device(self, x, side_stream=None):
        """x (B,3,H,W) on device, in [-1,1], H,W multiples of 256 -> device-side results.
        With `side_stream`, the rANS kernel (a serial recurrence: one lane per image, ~1 ms) is launched on that
        HIP stream so the caller's next kernels (the CLIP tower) overlap it; join with
        torch.cuda.current_stream().wait_stream(side_stream) before reading hs / hmeta."""
        cfg = self.cfg
        B, _, H, W = x.shape
        assert H % cfg.crop_size == 0 and W % cfg.crop_size == 0, "pad to a multiple of 256 first (compress.py:258)"
        z, h, (nH, nW) = self.encoder.forward(x)
        vq = ops.vq_argmin(z, self.codebook, l2norm=True)                  # [(B*nH*nW*T)] int32
        ntok = cfg.num_latent_tokens * nH * nW
        zs = ops.pack12_batch(vq, B, ntok)
        hh, ww = H // (2 * cfg.patch_size), W // (2 * cfg.patch_size)
        bn = self.bottleneck
        if bn.tables is None:
            raise RuntimeError("call hybrid_codec.quantize_feat.update(force=True) first (compress.py:239)")
        y = bn.analysis(h, B, hh, ww)
        sym, idx, _, _ = bn.quantise(y, B, hh, ww)
        n = 4 * (bn.Q // 4) * hh * ww
        if side_stream is not None:
            side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_stream):
                out, meta = ops.rans_encode_batch(bn.tables.handles[bn.group], sym, idx, B, n)
            for t in (sym, idx):
                t.record_stream(side_stream)
            for t in (out, meta):
                t.record_stream(torch.cuda.current_stream())
        else:
            out, meta = ops.rans_encode_batch(bn.tables.handles[bn.group], sym, idx, B, n)
        return dict(z=z, h=h, vq=vq, zs=zs, hs=out, hmeta=meta, sym=sym, idx=idx, n=n, stack=(nH, nW), feat_hw=(hh, ww), ntok=ntok)

    def encode_batch(self, x):
        """-> list of B enc_result dicts (the reference's encode_only contract, one per image)"""
        cfg = self.cfg
        B, _, H, W = x.shape
        r = self.encode_device(x)
        bn = self.bottleneck
        h_streams = BottleneckHIP.streams_to_host(r["hs"], r["hmeta"], retry=(bn.tables.handles[bn.group], r["sym"], r["idx"], r["n"]))
        zs = r["zs"].cpu().numpy()
        nH, nW = r["stack"]
        hh, ww = r["feat_hw"]
        res = []
        for b in range(B):
            res.append({
                "z_bit_stream": zs[b].tobytes(),
                "h_bit_stream": h_streams[b],
                "img_shape": (H, W),
                "feat_shape": torch.Size([1, cfg.feat_dim, hh, ww]),
                "stack_shape": (nH, nW),
                "token_length": r["ntok"],
                "z_indices_shape": torch.Size([nH * nW, cfg.token_size, 1, cfg.num_latent_tokens]),
            })
        return res

    # ------------------------------------------------------------------ decode side
    def _decoder(self):
        """decode-side weights are loaded on first use (a compress-only process never pays for them)"""
        if self._dec is None:
            from .decoder import FeatMergeHIP, HybridDecoderHIP, VqganDecoderHIP
            if "hybrid_codec.decoder.ln_pre.weight" not in self._sd:
                raise RuntimeError("state_dict has no decoder weights (hybrid_codec.decoder.*, prior_fusion.*, vqgan.*)")
            self._dec = (HybridDecoderHIP(self._sd, self.cfg, self.device), FeatMergeHIP(self._sd, self.cfg, self.device),
                         VqganDecoderHIP(self._sd, self.cfg, self.device))
        return self._dec

    def decode_batch(self, enc_results, taps=None):
        """B enc_result dicts with identical shapes (what encode_batch returned / unpack_c2df read) ->
        x_hat (B,3,H,W) on device, clamped to [-1,1]  (codec_sq_fixbpp.py:881-901)"""
        cfg = self.cfg
        B = len(enc_results)
        e0 = enc_results[0]
        H, W = (int(v) for v in e0["img_shape"])
        nH, nW = (int(v) for v in e0["stack_shape"])
        ntok = int(e0["token_length"])
        hh, ww = int(e0["feat_shape"][2]), int(e0["feat_shape"][3])
        hyb, fm, vq = self._decoder()
        # z branch: 12-bit unpack -> codebook rows -> l2 norm
        zb = np.stack([np.frombuffer(e["z_bit_stream"], dtype=np.uint8) for e in enc_results])
        zidx = ops.unpack12_batch(torch.from_numpy(zb).to(self.device), B, ntok)
        z_rows = ops.codebook_gather_norm(zidx.view(-1), self.codebook)
        # h branch
        h_hat = self.bottleneck.decompress([bytes(e["h_bit_stream"]) for e in enc_results], B, hh, ww)
        titok, feat = hyb.forward(z_rows, h_hat, B, (nH, nW))
        if taps is not None:
            taps.update(z_rows=z_rows.clone(), h_hat=h_hat.clone(), titok=titok.clone(), feat=feat.clone())
        Hf, Wf = nH * cfg.grid, nW * cfg.grid
        logits, latent = fm.forward(titok, feat, B, Hf, Wf)
        if taps is not None:
            taps.update(logits=logits.clone(), latent=latent.clone())
        return vq.forward(latent, B, Hf, Wf, tile16=True)

    @torch.no_grad()
    def decode_only(self, z_bit_stream, h_bit_stream, img_shape, feat_shape, stack_shape, token_length, z_indices_shape,
                    clip_stream=None, clip_meta=None):
        """reference signature (codec_sq_fixbpp.py:881)"""
        return self.decode_batch([dict(z_bit_stream=z_bit_stream, h_bit_stream=h_bit_stream, img_shape=img_shape,
                                       feat_shape=feat_shape, stack_shape=stack_shape, token_length=token_length,
                                       z_indices_shape=z_indices_shape)])

    @torch.no_grad()
    def encode_only(self, x):
        """reference signature: x (1,3,256a,256b) in [-1,1] -> dict (codec_sq_fixbpp.py:849-878)"""
        assert x.shape[0] == 1
        return self.encode_batch(x.to(self.device).float().contiguous())[0]


class ClipCodec:
    """compress.py:58-86"""

    def __init__(self, state_dict, cfg: ClipConfig = CLIP_B32, device="cuda:0", model_name="ViT-B-32",
                 pretrained="laion2b_s34b_b79k"):
        self.device = torch.device(device)
        self.model = ClipHIP(state_dict, cfg, self.device)
        self.model_name = f"{model_name}:{pretrained}"
        self.zctx = Compressor(level=19)

    def image_to_unit_vec(self, img_chw):
        unit, _ = self.model.encode(img_chw.to(self.device).float().contiguous()[None])
        return unit[0].cpu().numpy().astype("float32")

    def batch_to_codes(self, x, H=None, W=None):
        """(B,3,Hp,Wp) device batch -> (unit (B,D) device, u8 (B,D) device)"""
        return self.model.encode(x, H, W)

    def meta(self, dim):
        return {"model_id": self.model_name, "dim": int(dim), "quant": "u8_symmetric_-1_1", "codec": "zstd", "zstd_level": 19}

    def quantize_u8_and_compress(self, z_unit: np.ndarray):
        q = np.clip(np.round((z_unit * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
        return self.zctx.compress(q.tobytes()), self.meta(z_unit.shape[0])

    def compress_codes(self, q_u8_row: np.ndarray):
        return self.zctx.compress(q_u8_row.tobytes())
