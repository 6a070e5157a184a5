"""HybridEncoder forward on MI355X (reference: models/codec_sq_fixbpp.py:48-183, titok/blocks.py:26-115,
blocks/swin_transformer.py:64-156, blocks/conv_blocks.py:48-81, models/cross_blocks.py:39-98).

Host side only sequences kernel launches; every FLOP runs in libsgic (csrc/*.hip).  Data layout in HBM:
  * ViT tokens      X  [(n, l), W]  row = n*L + l, n = (b*nH + ty)*nW + tx   (batch-major, not LND)
  * detail features F  [(b, y, x), Fd] in 16x16-tile-major order (TM16) so tile n's 256 positions are the
                       contiguous rows n*256 .. n*256+255 -> the reference's tile<->stack rearranges vanish
  * joint buffer    J  [(n, 545), Fd]  rows 0..288 = projected ViT tokens, 289..544 = tile features
Residual adds, biases and activations are fused into the GEMM epilogues; token slices are addressed
through the GEMM/LayerNorm row maps instead of being copied.
"""
import numpy as np
import torch

from . import ops
from .config import CodecConfig


def _dev(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


class RabW:
    """ResidualAttentionBlock weights (titok/blocks.py:26-64)"""

    def __init__(self, sd, p, device):
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        self.ln1w, self.ln1b = g("ln_1.weight"), g("ln_1.bias")
        self.inw, self.inb = g("attn.in_proj_weight"), g("attn.in_proj_bias")
        self.ow, self.ob = g("attn.out_proj.weight"), g("attn.out_proj.bias")
        self.ln2w, self.ln2b = g("ln_2.weight"), g("ln_2.bias")
        self.fcw, self.fcb = g("mlp.c_fc.weight"), g("mlp.c_fc.bias")
        self.pjw, self.pjb = g("mlp.c_proj.weight"), g("mlp.c_proj.bias")


def rab_forward(X, w: RabW, L, nseq, heads, bias=None):
    """in place on X [(nseq*L), D]; bias: optional (1,L,L) additive attention mask (CLIP text tower: causal)"""
    D = X.shape[1]
    # to_gemm: an activation whose only consumer is the next GEMM goes out as that GEMM's operand (bf16x3 planes under the
    # split arithmetic, the plain fp32 tensor otherwise)
    h = ops.layernorm(X, w.ln1w, w.ln1b, to_gemm=True)
    qkv = ops.gemm(h, w.inw, w.inb)
    # the attention output reuses the LN buffer (fp32 tensor or planes)
    att = ops.attention(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], h, L, nseq, heads, bias=bias, to_gemm=True)
    ops.gemm(att, w.ow, w.ob, residual=X, out=X)
    h = ops.layernorm(X, w.ln2w, w.ln2b, out=h, to_gemm=True)
    f = ops.gemm(h, w.fcw, w.fcb, act=ops.ACT_GELU, to_gemm=True)
    ops.gemm(f, w.pjw, w.pjb, residual=X, out=X)
    return X


def rel_indices(win):
    idx = torch.tensor([[x, y] for x in range(win) for y in range(win)])
    return idx[None, :, :] - idx[:, None, :] + win - 1


class SwinW:
    """SwinBlock weights + the dense additive bias variants [plain | +UL | +LR | +UL+LR]
    (blocks/swin_transformer.py:64-92,110-117)"""

    def __init__(self, sd, p, shifted, rel, win, device):
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        self.shifted = shifted
        self.n1w, self.n1b = g("norm_attn.weight"), g("norm_attn.bias")
        self.qkvw = g("attention_block.to_qkv.weight")
        self.ow, self.ob = g("attention_block.to_out.weight"), g("attention_block.to_out.bias")
        self.n2w, self.n2b = g("norm_mlp.weight"), g("norm_mlp.bias")
        self.w0, self.b0 = g("mlp_block.net.0.weight"), g("mlp_block.net.0.bias")
        self.w2, self.b2 = g("mlp_block.net.2.weight"), g("mlp_block.net.2.bias")
        pos = sd[f"{p}.attention_block.pos_embedding"].float()
        if rel:
            ri = rel_indices(win)
            pos = pos[ri[:, :, 0], ri[:, :, 1]]
        if shifted:
            ul = sd[f"{p}.attention_block.upper_lower_mask"].float()
            lr = sd[f"{p}.attention_block.left_right_mask"].float()
            bias = torch.stack([pos, pos + ul, pos + lr, (pos + ul) + lr])
        else:
            bias = pos[None]
        self.bias = _dev(bias, device)


_ROWMAP_CACHE = {}


def swin_rowmap(B, H, W, win, shifted, device):
    """row (TM16 order) of window-token (seq=(b,wy,wx), t=(i,j)) after the cyclic shift, and the bias
    variant of each window (bit0: last window row -> upper_lower mask, bit1: last window col -> left_right)"""
    key = (B, H, W, win, shifted, str(device))
    if key not in _ROWMAP_CACHE:
        nh, nw = H // win, W // win
        s = win // 2 if shifted else 0
        b, wy, wx, i, j = np.meshgrid(np.arange(B), np.arange(nh), np.arange(nw), np.arange(win), np.arange(win), indexing="ij")
        y = (wy * win + i + s) % H
        x = (wx * win + j + s) % W
        nH, nW = H // 16, W // 16
        row = (((b * nH + y // 16) * nW + x // 16) << 8) + ((y % 16) << 4) + (x % 16)
        var = np.zeros((B, nh, nw), dtype=np.int32)
        if shifted:
            var[:, nh - 1, :] |= 1
            var[:, :, nw - 1] |= 2
        _ROWMAP_CACHE[key] = (torch.from_numpy(row.reshape(-1).astype(np.int32)).to(device),
                              torch.from_numpy(var.reshape(-1)).to(device))
    return _ROWMAP_CACHE[key]


def swin_forward(Fm, w: SwinW, B, H, W, win=16):
    """in place on the TM16 feature map Fm [(B*H*W), C]"""
    C = Fm.shape[1]
    rowmap, var = swin_rowmap(B, H, W, win, w.shifted, Fm.device)
    h = ops.layernorm(Fm, w.n1w, w.n1b, to_gemm=True)
    qkv = ops.gemm(h, w.qkvw)
    nseq = B * (H // win) * (W // win)
    att = ops.attention(qkv[:, 0:C], qkv[:, C:2 * C], qkv[:, 2 * C:3 * C], h, win * win, nseq, C // 64, rowmap=rowmap,
                        bias=w.bias, biasvar=var if w.shifted else None, to_gemm=True)
    ops.gemm(att, w.ow, w.ob, residual=Fm, out=Fm)
    h = ops.layernorm(Fm, w.n2w, w.n2b, out=h, to_gemm=True)
    f = ops.gemm(h, w.w0, w.b0, act=ops.ACT_GELU, to_gemm=True)
    ops.gemm(f, w.w2, w.b2, residual=Fm, out=Fm)
    return Fm


def swin_stack_weights(sd, p, n, win, device, first_index=1):
    return [SwinW(sd, f"{p}.{first_index + i}", bool(i % 2), i == 0, win, device) for i in range(n)]


class ConvNextW:
    def __init__(self, sd, p, device):
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        C = sd[f"{p}.conv.weight"].shape[0]
        self.ls = _dev(sd[f"{p}.layer_scale"].reshape(C), device)
        self.dw = _dev(sd[f"{p}.conv.weight"].reshape(C, 25).t(), device)  # [k*k][C]
        self.db = g("conv.bias")
        self.nw, self.nb = g("norm.weight"), g("norm.bias")
        self.w0, self.b0 = g("mlp.0.weight"), g("mlp.0.bias")
        self.w2, self.b2 = g("mlp.2.weight"), g("mlp.2.bias")


def convnext_forward(Fm, w: ConvNextW, B, H, W):
    t = ops.dwconv(Fm, w.dw, w.db, w.ls, B, H, W, 5, tile16=True)
    t = ops.layernorm(t, w.nw, w.nb, out=t, to_gemm=True)
    f = ops.gemm(t, w.w0, w.b0, act=ops.ACT_GELU, to_gemm=True)
    ops.gemm(f, w.w2, w.b2, residual=Fm, out=Fm)
    return Fm


class CrossW:
    def __init__(self, sd, p, n_attn, device):
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        self.tpos = _dev(sd[f"{p}.titok_pos_emb"].squeeze(1), device)   # (289, W)
        self.fpos = _dev(sd[f"{p}.feat_pos_emb"].squeeze(1), device)    # (256, F)
        self.cw, self.cb = g("titok_compress_proj.weight"), g("titok_compress_proj.bias")
        self.attn = [RabW(sd, f"{p}.attn.{j}", device) for j in range(n_attn)]
        self.dw, self.db = g("titok_decompress_proj.0.weight"), g("titok_decompress_proj.0.bias")
        self.dnw, self.dnb = g("titok_decompress_proj.1.weight"), g("titok_decompress_proj.1.bias")
        self.fnw, self.fnb = g("feat_add.0.weight"), g("feat_add.0.bias")
        self.fw, self.fb = g("feat_add.1.weight"), g("feat_add.1.bias")
        self.zw, self.zb = g("zero_add.weight"), g("zero_add.bias")


def cross_forward(Fm, X, w: CrossW, N, Lt, P2):
    """Interactive_crossAttn_type4 (models/cross_blocks.py:75-98); Fm [(N*P2), F] TM16, X [(N*Lt), W]"""
    Fd = Fm.shape[1]
    LJ = Lt + P2
    J = torch.empty(N * LJ, Fd, device=X.device, dtype=torch.float32)
    tmp = torch.empty_like(X)
    ops.add_rows_bcast(X, Lt, w.tpos, tmp, Lt, N, Lt)
    ops.gemm(tmp, w.cw, w.cb, out=J, M=N * Lt, c_seg=(Lt, LJ))
    ops.add_rows_bcast(Fm, P2, w.fpos, J[Lt:], LJ, N, P2)
    for rw in w.attn:
        rab_forward(J, rw, LJ, N, Fd // 64)
    t = ops.layernorm(J[Lt:], w.fnw, w.fnb, M=N * P2, x_seg=(P2, LJ), to_gemm=True)
    ops.gemm(t, w.fw, w.fb, residual=Fm, out=Fm)
    t = ops.gemm(J, w.dw, w.db, M=N * Lt, a_seg=(Lt, LJ))
    t = ops.layernorm(t, w.dnw, w.dnb, out=t, act=ops.ACT_SILU, to_gemm=True)
    ops.gemm(t, w.zw, w.zb, residual=X, out=X)
    return Fm, X


class HybridEncoderHIP:
    def __init__(self, sd, cfg: CodecConfig, device, p="hybrid_codec.encoder"):
        self.cfg, self.device = cfg, device
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        Wd, Fd = cfg.width, cfg.feat_dim
        self.pe_w = _dev(sd[f"{p}.patch_embed.weight"].reshape(Wd, -1), device)
        self.pe_b = g("patch_embed.bias")
        self.cls, self.pos, self.latpos = g("class_embedding"), g("positional_embedding"), g("latent_token_positional_embedding")
        self.lat = _dev(sd["hybrid_codec.latent_tokens"], device)
        self.lnpre_w, self.lnpre_b = g("ln_pre.weight"), g("ln_pre.bias")
        self.layers = [RabW(sd, f"{p}.transformer.{i}", device) for i in range(cfg.layers)]
        self.lnpost_w, self.lnpost_b = g("ln_post.weight"), g("ln_post.bias")
        self.co_w = _dev(sd[f"{p}.conv_out.weight"].reshape(cfg.token_size, Wd), device)
        self.co_b = g("conv_out.bias")
        self.pix_w = _dev(sd[f"{p}.pix_emb_proj.weight"].reshape(Fd, Wd), device)
        self.pix_b = g("pix_emb_proj.bias")
        self.cross, self.fblocks = {}, {}
        for i in cfg.in_pos_enc:
            self.cross[i] = CrossW(sd, f"{p}.inter_blocks.{i}", cfg.n_attn, device)
            self.fblocks[i] = (swin_stack_weights(sd, f"{p}.feat_blocks.{i}.0", 2, cfg.window, device),
                               ConvNextW(sd, f"{p}.feat_blocks.{i}.1", device), ConvNextW(sd, f"{p}.feat_blocks.{i}.2", device))
        self.feat_in = swin_stack_weights(sd, f"{p}.feat_in", 4, cfg.window, device)
        self.feat_out_swin = swin_stack_weights(sd, f"{p}.feat_out.0", 2, cfg.window, device)
        # 2x2/s2 conv as GEMM over K = (ky, kx, cin), matching sgic_im2col_2x2
        self.fo_cw = _dev(sd[f"{p}.feat_out.1.weight"].permute(0, 2, 3, 1).reshape(Fd, 4 * Fd), device)
        self.fo_cb = g("feat_out.1.bias")
        self.fo_nw, self.fo_nb = g("feat_out.3.weight"), g("feat_out.3.bias")
        self.fo_lw, self.fo_lb = g("feat_out.4.weight"), g("feat_out.4.bias")

    def forward(self, x, taps=None):
        """x (B,3,H,W) in [-1,1], H and W multiples of 256.  Returns
        z [(N*T), token_size]  (row n*T + t  <->  reference z[n, :, 0, t]),
        h [(B*(Hf/2)*(Wf/2)), Fd] plain NHWC  (<-> reference h[b, :, y, x]),  stack_shape (nH, nW)."""
        cfg = self.cfg
        B, _, H, W = x.shape
        P, g, T, Wd = cfg.patch_size, cfg.grid, cfg.num_latent_tokens, cfg.width
        Hf, Wf = H // P, W // P
        nH, nW = Hf // g, Wf // g
        N, P2 = B * nH * nW, g * g
        L = 1 + P2 + T
        A = ops.im2col_patch(x, P, 0.5, 0.5, tile16=True)              # x*0.5+0.5 fused (codec_sq_fixbpp.py:855)
        emb = ops.gemm(A, self.pe_w, self.pe_b)                        # [(N*256), W] TM16
        Fm = ops.gemm(emb, self.pix_w, self.pix_b)                     # [(N*256), F] TM16
        X = ops.assemble_tokens(emb, self.cls, self.pos, self.lat, self.latpos, N, P2, T, Wd)
        for w in self.feat_in:
            swin_forward(Fm, w, B, Hf, Wf, cfg.window)
        ops.layernorm(X, self.lnpre_w, self.lnpre_b, out=X)
        if taps is not None:
            taps["feat_in"] = Fm.clone()
            taps["x_ln_pre"] = X.clone()
        for i in range(cfg.layers):
            rab_forward(X, self.layers[i], L, N, cfg.heads)
            if i in self.cross:
                cross_forward(Fm, X, self.cross[i], N, 1 + P2 + T, P2)
                sw, c1, c2 = self.fblocks[i]
                for w in sw:
                    swin_forward(Fm, w, B, Hf, Wf, cfg.window)
                convnext_forward(Fm, c1, B, Hf, Wf)
                convnext_forward(Fm, c2, B, Hf, Wf)
            if taps is not None and i == 0:
                taps["x_layer0"] = X.clone()
        lat = ops.layernorm(X[1 + P2:], self.lnpost_w, self.lnpost_b, M=N * T, x_seg=(T, L))   # [(N*T), W]
        latT = ops.fake2d_transpose(lat, T * Wd, N, T, Wd)            # out[n][t][c] = flat_n[c*T + t]
        z = ops.gemm(latT, self.co_w, self.co_b)                       # [(N*T), token_size]
        for w in self.feat_out_swin:
            swin_forward(Fm, w, B, Hf, Wf, cfg.window)
        A2 = ops.im2col_2x2(Fm, B, Hf, Wf, tile16=True)
        h = ops.gemm(A2, self.fo_cw, self.fo_cb)
        ops.layernorm(h, self.fo_nw, self.fo_nb, out=h)
        h = ops.gemm(h, self.fo_lw, self.fo_lb)
        return z, h, (nH, nW)
