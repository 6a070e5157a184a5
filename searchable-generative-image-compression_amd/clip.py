"""OpenCLIP ViT-B/32 image tower + PIL-exact preprocessing on MI355X (reference: ClipCodec in
compress.py:58-86; open_clip is a third-party dependency that is not vendored, so the architecture is
restated from its published definition: conv 32x32/s32 no bias, class token, learned positions, ln_pre,
pre-LN residual attention blocks with exact GELU, ln_post on the class token, linear projection)."""
import math

import numpy as np
import torch

from . import ops
from ._lib import call
from .config import ClipConfig
from .encoder import RabW, rab_forward

_PREC = 22  # Pillow PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter (double precision on the
    host, like Pillow).  -> (bounds int32 (out,2), kk int32 (out,ksize), ksize)"""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / fscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << _PREC)) if v < 0 else int(0.5 + v * (1 << _PREC))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def resize_geometry(H, W, S):
    """torchvision Resize(S) (shortest side -> S, other = int(S*long/short)) + CenterCrop(S)"""
    if H <= W:
        OH, OW = S, int(S * W / H)
    else:
        OH, OW = int(S * H / W), S
    top = int(round((OH - S) / 2.0))
    left = int(round((OW - S) / 2.0))
    return OH, OW, top, left


class ClipHIP:
    def __init__(self, sd, cfg: ClipConfig, device, p="clip.visual"):
        self.cfg, self.device = cfg, device
        g = lambda k: sd[f"{p}.{k}"].to(device=device, dtype=torch.float32).contiguous()
        Wd = cfg.width
        self.conv_w = sd[f"{p}.conv1.weight"].reshape(Wd, -1).to(device).contiguous()
        self.cls = g("class_embedding").reshape(1, Wd).contiguous()
        self.pos = g("positional_embedding")
        self.lnpre_w, self.lnpre_b = g("ln_pre.weight"), g("ln_pre.bias")
        self.blocks = [RabW(sd, f"{p}.transformer.resblocks.{i}", device) for i in range(cfg.layers)]
        self.lnpost_w, self.lnpost_b = g("ln_post.weight"), g("ln_post.bias")
        self.projT = sd[f"{p}.proj"].t().to(device).contiguous()          # (embed_dim, width)
        self.mean = np.asarray(cfg.mean, dtype=np.float32)
        self.std = np.asarray(cfg.std, dtype=np.float32)
        self._coef = {}

    def _coeffs(self, H, W):
        key = (H, W)
        if key not in self._coef:
            S = self.cfg.image_size
            OH, OW, top, left = resize_geometry(H, W, S)
            bh, kh, ksh = pil_coeffs(W, OW)
            bv, kv, ksv = pil_coeffs(H, OH)
            d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self._coef[key] = (OH, OW, top, left, d(bh), d(kh), ksh, d(bv), d(kv), ksv)
        return self._coef[key]

    def preprocess(self, x, H=None, W=None):
        """x (B,3,Hp,Wp) fp32 in [-1,1]; the top-left HxW region of every image is the real (unpadded)
        image (compress.py:266 feeds the UNPADDED image to CLIP).  -> (B,3,S,S) normalised."""
        B, _, Hp, Wp = x.shape
        H, W = H or Hp, W or Wp
        S = self.cfg.image_size
        OH, OW, top, left, bh, kh, ksh, bv, kv, ksv = self._coeffs(H, W)
        u8 = torch.empty(B * 3 * H * W, dtype=torch.uint8, device=x.device)
        th = torch.empty(B * 3 * H * OW, dtype=torch.uint8, device=x.device)
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=x.device)
        import ctypes
        call("sgic_clip_preprocess", ops._p(x), ctypes.c_long(x.stride(0)), ctypes.c_long(x.stride(1)), int(x.stride(2)),
             B, H, W, OH, OW, S, top, left, ops._p(bh), ops._p(kh), ksh, ops._p(bv), ops._p(kv), ksv,
             self.mean.ctypes.data_as(ctypes.c_void_p), self.std.ctypes.data_as(ctypes.c_void_p), ops._p(u8), ops._p(th),
             ops._p(out))
        return out

    def tower(self, pre):
        """(B,3,S,S) normalised -> (unit fp32 (B,D), u8 codes (B,D))"""
        cfg = self.cfg
        B = pre.shape[0]
        g = cfg.image_size // cfg.patch
        L = g * g + 1
        A = ops.im2col_patch(pre, cfg.patch, 1.0, 0.0, tile16=False)
        emb = ops.gemm(A, self.conv_w)
        X = ops.assemble_tokens(emb, self.cls, self.pos, None, None, B, g * g, 0, cfg.width)
        ops.layernorm(X, self.lnpre_w, self.lnpre_b, out=X)
        for w in self.blocks:
            rab_forward(X, w, L, B, cfg.heads)
        pooled = ops.layernorm(X, self.lnpost_w, self.lnpost_b, M=B, x_seg=(1, L))
        z = ops.gemm(pooled, self.projT)
        return ops.l2norm_u8(z)

    def encode(self, x, H=None, W=None):
        return self.tower(self.preprocess(x, H, W))


class ClipTextHIP:
    """OpenCLIP text tower (search.py:54-55,93-97 -> open_clip CLIP.encode_text; third-party, architecture restated
    from its published definition): token embedding + learned positions, `t_layers` pre-LN residual attention blocks
    with a causal additive mask, ln_final, the row of the EOT token (argmax of the ids) times text_projection.
    Tokenisation stays with the caller (open_clip's BPE vocabulary is not part of the reference tree)."""

    def __init__(self, sd, cfg: ClipConfig, device, p="clip"):
        self.cfg, self.device = cfg, device
        g = lambda k: sd[f"{p}.{k}"].to(device=device, dtype=torch.float32).contiguous()
        self.table, self.pos = g("token_embedding.weight"), g("positional_embedding")
        self.blocks = [RabW(sd, f"{p}.transformer.resblocks.{i}", device) for i in range(cfg.t_layers)]
        self.lnf_w, self.lnf_b = g("ln_final.weight"), g("ln_final.bias")
        self.projT = sd[f"{p}.text_projection"].t().to(device=device, dtype=torch.float32).contiguous()   # (embed_dim, width)
        # the attention kernel reads its bias in float4s (L % 4 == 0): run on ctx rounded up to a multiple of 4; under
        # the causal mask the extra trailing positions (id 0, zero position row) cannot reach the real ones.
        L = self.L = (cfg.ctx + 3) & ~3
        if L != cfg.ctx:
            self.pos = torch.cat([self.pos, torch.zeros(L - cfg.ctx, cfg.t_width, device=device)]).contiguous()
        self.causal = torch.full((L, L), float("-inf")).triu_(1).reshape(1, L, L).to(device).contiguous()

    def encode_text(self, tokens):
        """tokens: int tensor/array (B, ctx) -> (unit fp32 (B, embed_dim) on the device)"""
        cfg = self.cfg
        ids = torch.as_tensor(np.asarray(tokens.cpu() if torch.is_tensor(tokens) else tokens)).to(torch.int32)
        if ids.dim() != 2 or ids.shape[1] != cfg.ctx:
            raise ValueError(f"tokens must be (B, {cfg.ctx}), got {tuple(ids.shape)}")
        if self.L != cfg.ctx:
            ids = torch.nn.functional.pad(ids, (0, self.L - cfg.ctx))
        ids = ids.to(self.device).contiguous()
        B, L, D = ids.shape[0], self.L, cfg.t_width
        X = torch.empty(B * L, D, dtype=torch.float32, device=self.device)
        call("sgic_embed_tokens", ops._p(ids), ops._p(self.table), ops._p(self.pos), ops._p(X), B, L, D, cfg.vocab)
        for w in self.blocks:
            rab_forward(X, w, L, B, cfg.t_heads, bias=self.causal)
        pooled = torch.empty(B, D, dtype=torch.float32, device=self.device)
        call("sgic_gather_eot_rows", ops._p(ids), ops._p(X), D, ops._p(pooled), B, L, D)
        ops.layernorm(pooled, self.lnf_w, self.lnf_b, out=pooled)
        z = ops.gemm(pooled, self.projT)
        return ops.l2norm_u8(z)[0]
