"""Decode-side transforms on MI355X: HybridDecoder, FeatMerge + soft codebook lookup, taming VQGAN decoder
(reference: models/codec_sq_fixbpp.py:186-300,395-439,658-669; taming/modules/diffusionmodules/model.py:
78-192,436-537).  Same layouts as encoder.py; the VQGAN part runs on plain NHWC maps, its 3x3 convolutions are
implicit GEMMs on the matrix cores over zero-halo buffers that the GroupNorm+swish kernel writes directly."""
import torch

from . import ops
from .config import CodecConfig
from .encoder import (ConvNextW, CrossW, RabW, convnext_forward, cross_forward, rab_forward, swin_forward,
                      swin_stack_weights, _dev)
from .weights import vqgan_plan


class HybridDecoderHIP:
    def __init__(self, sd, cfg: CodecConfig, device, p="hybrid_codec.decoder"):
        self.cfg, self.device = cfg, device
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        Wd, Fd = cfg.width, cfg.feat_dim
        self.de_w, self.de_b = g("decoder_embed.weight"), g("decoder_embed.bias")
        self.cls, self.pos, self.latpos = g("class_embedding"), g("positional_embedding"), g("latent_token_positional_embedding")
        self.mask = _dev(sd[f"{p}.mask_token"].reshape(1, Wd), device)
        self.lnpre_w, self.lnpre_b = g("ln_pre.weight"), g("ln_pre.bias")
        self.layers = [RabW(sd, f"{p}.transformer.{i}", device) for i in range(cfg.layers)]
        self.lnpost_w, self.lnpost_b = g("ln_post.weight"), g("ln_post.bias")
        self.up_w = _dev(sd[f"{p}.init_feat_up.0.weight"].reshape(4 * Fd, Fd), device)
        self.up_b = g("init_feat_up.0.bias")
        self.up_swin = swin_stack_weights(sd, f"{p}.init_feat_up.2", 4, cfg.window, device)
        self.cross, self.fblocks = {}, {}
        for i in cfg.in_pos_dec:
            self.cross[i] = CrossW(sd, f"{p}.inter_blocks.{i}", cfg.n_attn, device)
            self.fblocks[i] = (swin_stack_weights(sd, f"{p}.feat_blocks.{i}.0", 2, cfg.window, device),
                               ConvNextW(sd, f"{p}.feat_blocks.{i}.1", device), ConvNextW(sd, f"{p}.feat_blocks.{i}.2", device))

    def forward(self, z_rows, h, B, stack):
        """z_rows [(N*T), token_size] l2-normalised code vectors, h [(B*hh*ww), Fd] plain NHWC.
        Returns titok_hat [(N*256), W] and feat_hat [(N*256), Fd], both tile-major over (B, 16nH, 16nW)."""
        cfg = self.cfg
        nH, nW = stack
        g, T, Wd, Fd = cfg.grid, cfg.num_latent_tokens, cfg.width, cfg.feat_dim
        N, P2 = B * nH * nW, g * g
        L = 1 + P2 + T
        hh, ww = nH * g // 2, nW * g // 2
        emb = ops.gemm(z_rows, self.de_w, self.de_b)                                  # decoder_embed
        X = ops.assemble_dec_tokens(emb, self.cls, self.mask, self.pos, self.latpos, N, P2, T, Wd)
        up = ops.gemm(h, self.up_w, self.up_b)                                        # 1x1 conv F -> 4F
        Fm = ops.pixel_shuffle2_tm16(up, B, hh, ww, Fd)
        Hf, Wf = 2 * hh, 2 * ww
        for w in self.up_swin:
            swin_forward(Fm, w, B, Hf, Wf, cfg.window)
        ops.layernorm(X, self.lnpre_w, self.lnpre_b, out=X)
        for i in range(cfg.layers):
            rab_forward(X, self.layers[i], L, N, cfg.heads)
            if i in self.cross:
                cross_forward(Fm, X, self.cross[i], N, L, P2)
                sw, c1, c2 = self.fblocks[i]
                for w in sw:
                    swin_forward(Fm, w, B, Hf, Wf, cfg.window)
                convnext_forward(Fm, c1, B, Hf, Wf)
                convnext_forward(Fm, c2, B, Hf, Wf)
        titok = ops.layernorm(X[1:], self.lnpost_w, self.lnpost_b, M=N * P2, x_seg=(P2, L))   # drop cls, keep 256 patches
        return titok, Fm


class FeatMergeHIP:
    """FeatMerge + softmax(logits) @ codebook (codec_sq_fixbpp.py:395-439,658-663)"""

    def __init__(self, sd, cfg: CodecConfig, device, p="prior_fusion"):
        self.cfg = cfg
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        self.feat_in = swin_stack_weights(sd, f"{p}.feat_in.1", 2, cfg.window, device, first_index=0)
        self.titok_in = swin_stack_weights(sd, f"{p}.titok_in.1", 2, cfg.window, device, first_index=0)
        self.m0w, self.m0b = g("merge.0.weight"), g("merge.0.bias")
        self.m1w, self.m1b = g("merge.1.weight"), g("merge.1.bias")
        self.m3w, self.m3b = g("merge.3.weight"), g("merge.3.bias")
        self.merge_swin = swin_stack_weights(sd, f"{p}.merge.4", 4, cfg.window, device, first_index=0)
        self.f0w, self.f0b = g("ffn.0.weight"), g("ffn.0.bias")
        self.f1w, self.f1b = g("ffn.1.weight"), g("ffn.1.bias")
        self.f3w, self.f3b = g("ffn.3.weight"), g("ffn.3.bias")
        # soft lookup as a GEMM: latent[m, d] = sum_c P[m, c] * E[c, d]  ->  W = E^T  (n_embed x n_embed here)
        self.embT = _dev(sd["vqgan.quantize.embedding.weight"].t(), device)

    def forward(self, titok, feat, B, Hf, Wf):
        """titok [(M), W], feat [(M), F] tile-major rows -> (logits [(M), n_embed], latent [(M), embed_dim])"""
        cfg = self.cfg
        Wd, Fd = titok.shape[1], feat.shape[1]
        for w in self.titok_in:
            swin_forward(titok, w, B, Hf, Wf, cfg.window)
        for w in self.feat_in:
            swin_forward(feat, w, B, Hf, Wf, cfg.window)
        M = titok.shape[0]
        cat = torch.empty(M, Wd + Fd, device=titok.device)
        ops.add_rows_bcast(titok, M, None, cat[:, :Wd], M, 1, M)      # strided copies into the concat buffer
        ops.add_rows_bcast(feat, M, None, cat[:, Wd:], M, 1, M)
        h = ops.gemm(cat, self.m0w, self.m0b)
        ops.layernorm(h, self.m1w, self.m1b, out=h, act=ops.ACT_SILU)
        h = ops.gemm(h, self.m3w, self.m3b)
        for w in self.merge_swin:
            swin_forward(h, w, B, Hf, Wf, cfg.window)
        t = ops.layernorm(h, self.f0w, self.f0b)
        t = ops.gemm(t, self.f1w, self.f1b, act=ops.ACT_TANH)
        logits = ops.gemm(t, self.f3w, self.f3b)
        probs = ops.softmax_rows(logits, logits.shape[1])
        latent = ops.gemm(probs, self.embT)
        return logits, latent


class _ResW:
    def __init__(self, sd, p, device):
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        cw = lambda k: _dev(sd[f"{p}.{k}"].permute(0, 2, 3, 1).reshape(sd[f"{p}.{k}"].shape[0], -1), device)  # (co,(ky,kx,ci))
        self.n1w, self.n1b, self.n2w, self.n2b = g("norm1.weight"), g("norm1.bias"), g("norm2.weight"), g("norm2.bias")
        self.c1w, self.c1b, self.c2w, self.c2b = cw("conv1.weight"), g("conv1.bias"), cw("conv2.weight"), g("conv2.bias")
        self.cin, self.cout = sd[f"{p}.conv1.weight"].shape[1], sd[f"{p}.conv1.weight"].shape[0]
        self.sw = self.sb = None
        if f"{p}.nin_shortcut.weight" in sd:
            self.sw = _dev(sd[f"{p}.nin_shortcut.weight"].reshape(self.cout, self.cin), device)
            self.sb = g("nin_shortcut.bias")


class _AttnW:
    def __init__(self, sd, p, device):
        g = lambda k: _dev(sd[f"{p}.{k}"], device)
        c = sd[f"{p}.q.weight"].shape[0]
        m = lambda k: _dev(sd[f"{p}.{k}"].reshape(c, c), device)
        self.c = c
        self.nw, self.nb = g("norm.weight"), g("norm.bias")
        self.qw, self.qb, self.kw, self.kb = m("q.weight"), g("q.bias"), m("k.weight"), g("k.bias")
        self.vw, self.vb, self.pw, self.pb = m("v.weight"), g("v.bias"), m("proj_out.weight"), g("proj_out.bias")


class VqganDecoderHIP:
    """post_quant_conv + taming Decoder (codec_sq_fixbpp.py:666-669, model.py:436-537)"""

    def __init__(self, sd, cfg: CodecConfig, device, p="vqgan"):
        self.cfg = cfg
        d = f"{p}.decoder"
        g = lambda k: _dev(sd[k], device)
        cw = lambda k: _dev(sd[k].permute(0, 2, 3, 1).reshape(sd[k].shape[0], -1), device)
        self.pq_w = _dev(sd[f"{p}.post_quant_conv.weight"].reshape(cfg.vq_z_channels, cfg.vq_embed_dim), device)
        self.pq_b = g(f"{p}.post_quant_conv.bias")
        self.ci_w, self.ci_b = cw(f"{d}.conv_in.weight"), g(f"{d}.conv_in.bias")
        self.mid1, self.mida, self.mid2 = _ResW(sd, f"{d}.mid.block_1", device), _AttnW(sd, f"{d}.mid.attn_1", device), _ResW(sd, f"{d}.mid.block_2", device)
        self.plan, self.c0, self.c_last = vqgan_plan(cfg)
        self.levels = []
        for lvl, blocks, up in self.plan:
            res = [_ResW(sd, f"{d}.up.{lvl}.block.{i}", device) for i in range(len(blocks))]
            att = [_AttnW(sd, f"{d}.up.{lvl}.attn.{i}", device) if blocks[i][2] else None for i in range(len(blocks))]
            upw = (cw(f"{d}.up.{lvl}.upsample.conv.weight"), g(f"{d}.up.{lvl}.upsample.conv.bias")) if up else None
            self.levels.append((res, att, upw))
        self.no_w, self.no_b = g(f"{d}.norm_out.weight"), g(f"{d}.norm_out.bias")
        self.co_w, self.co_b = cw(f"{d}.conv_out.weight"), g(f"{d}.conv_out.bias")

    @staticmethod
    def _res(x, w: _ResW, B, H, W):
        # to_conv = (Cout, has_residual) of the consuming 3x3 conv: under the split arithmetic the halo buffer is written directly
        # as that conv's operand planes
        h = ops.groupnorm(x, w.n1w, w.n1b, B, H, W, swish=True, halo=True, to_conv=(w.cout, False))
        h = ops.conv3x3(h, w.c1w, w.c1b, B, H, W, w.cin, w.cout)
        h = ops.groupnorm(h, w.n2w, w.n2b, B, H, W, swish=True, halo=True, to_conv=(w.cout, True))
        sc = ops.gemm(x, w.sw, w.sb) if w.sw is not None else x
        return ops.conv3x3(h, w.c2w, w.c2b, B, H, W, w.cout, w.cout, residual=sc)

    @staticmethod
    def _attn(x, w: _AttnW, B, H, W):
        """single-head attention over the H*W positions of each image (model.py:168-192)"""
        L, c = H * W, w.c
        h = ops.groupnorm(x, w.nw, w.nb, B, H, W, swish=False, halo=False)
        q = ops.gemm(h, w.qw, w.qb)
        k = ops.gemm(h, w.kw, w.kb)
        S = torch.empty(B, L, L, device=x.device)
        ops.gemm_batched(q, c, L * c, k, c, L * c, S, L, L * L, L, L, c, B)          # S_b = Q_b K_b^T
        Pm = ops.softmax_rows(S, L, scale=float(int(c) ** (-0.5)), out=S)
        vT = torch.empty(B, c, L, device=x.device)                                    # V_b^T = Wv h_b^T (bias added below)
        ops.gemm_batched(w.vw, c, 0, h, c, L * c, vT, L, c * L, c, L, c, B)
        o = torch.empty(B * L, c, device=x.device)
        ops.gemm_batched(Pm, L, L * L, vT, L, c * L, o, c, L * c, L, c, L, B, bias=w.vb)  # rows of P sum to 1 -> + bv
        return ops.gemm(o, w.pw, w.pb, residual=x)

    def forward(self, latent, B, H, W, tile16=True):
        """latent [(B*H*W), embed_dim] (tile-major rows if tile16) -> x_hat (B,3,16H,16W) clamped"""
        cfg = self.cfg
        z = ops.gemm(latent, self.pq_w, self.pq_b)                                    # post_quant_conv 1x1
        zh = ops.halo_copy(z, B, H, W, cfg.vq_z_channels, upsample=False, tile16=tile16, to_conv=(self.c0, False))
        h = ops.conv3x3(zh, self.ci_w, self.ci_b, B, H, W, cfg.vq_z_channels, self.c0)
        h = self._res(h, self.mid1, B, H, W)
        h = self._attn(h, self.mida, B, H, W)
        h = self._res(h, self.mid2, B, H, W)
        for res, att, upw in self.levels:
            for r, a in zip(res, att):
                h = self._res(h, r, B, H, W)
                if a is not None:
                    h = self._attn(h, a, B, H, W)
            if upw is not None:
                c = h.shape[1]
                hu = ops.halo_copy(h, B, H, W, c, upsample=True, tile16=False, to_conv=(c, False))
                H, W = 2 * H, 2 * W
                h = ops.conv3x3(hu, upw[0], upw[1], B, H, W, c, c)
        hn = ops.groupnorm(h, self.no_w, self.no_b, B, H, W, swish=True, halo=True)
        o = torch.empty(B * H * W, 4, device=latent.device)
        ops.conv3x3(hn, self.co_w, self.co_b, B, H, W, self.c_last, 3, out=o[:, :3])
        return ops.nhwc3_to_nchw_clamp(o, 4, B, H, W)
