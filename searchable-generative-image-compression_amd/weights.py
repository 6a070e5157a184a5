"""Parameter inventory (names = the reference checkpoint's state_dict keys, SURVEY.md §8a "weights
note") and a deterministic synthetic-weight generator.  The real checkpoint is not available offline
(checkpoints/ is empty upstream), so parity and benchmarks use synthetic weights of the exact
architecture; `load_state_dict`-style dicts with the same keys drop in unchanged."""
import math
import zlib

import numpy as np
import torch

from .config import ClipConfig, CodecConfig


def _rab(prefix, d, out):
    out += [(f"{prefix}.ln_1.weight", (d,), "ln_w"), (f"{prefix}.ln_1.bias", (d,), "ln_b"),
            (f"{prefix}.attn.in_proj_weight", (3 * d, d), "lin"), (f"{prefix}.attn.in_proj_bias", (3 * d,), "bias"),
            (f"{prefix}.attn.out_proj.weight", (d, d), "lin"), (f"{prefix}.attn.out_proj.bias", (d,), "bias"),
            (f"{prefix}.ln_2.weight", (d,), "ln_w"), (f"{prefix}.ln_2.bias", (d,), "ln_b"),
            (f"{prefix}.mlp.c_fc.weight", (4 * d, d), "lin"), (f"{prefix}.mlp.c_fc.bias", (4 * d,), "bias"),
            (f"{prefix}.mlp.c_proj.weight", (d, 4 * d), "lin"), (f"{prefix}.mlp.c_proj.bias", (d,), "bias")]


def _swin(prefix, d, shifted, rel, win, out):
    w2 = win * win
    out += [(f"{prefix}.norm_attn.weight", (d,), "ln_w"), (f"{prefix}.norm_attn.bias", (d,), "ln_b")]
    if shifted:
        out += [(f"{prefix}.attention_block.upper_lower_mask", (w2, w2), "mask_ul"),
                (f"{prefix}.attention_block.left_right_mask", (w2, w2), "mask_lr")]
    out += [(f"{prefix}.attention_block.to_qkv.weight", (3 * d, d), "lin"),
            (f"{prefix}.attention_block.pos_embedding", (2 * win - 1, 2 * win - 1) if rel else (w2, w2), "posbias"),
            (f"{prefix}.attention_block.to_out.weight", (d, d), "lin"), (f"{prefix}.attention_block.to_out.bias", (d,), "bias"),
            (f"{prefix}.norm_mlp.weight", (d,), "ln_w"), (f"{prefix}.norm_mlp.bias", (d,), "ln_b"),
            (f"{prefix}.mlp_block.net.0.weight", (4 * d, d), "lin"), (f"{prefix}.mlp_block.net.0.bias", (4 * d,), "bias"),
            (f"{prefix}.mlp_block.net.2.weight", (d, 4 * d), "lin"), (f"{prefix}.mlp_block.net.2.bias", (d,), "bias")]


def _swin_stack(prefix, d, n, win, out, first_index=1):
    """get_swin (codec_sq_fixbpp.py:33-45): Sequential[Rearrange?, SwinBlock x n, Rearrange?]"""
    for i in range(n):
        _swin(f"{prefix}.{first_index + i}", d, shifted=bool(i % 2), rel=(i == 0), win=win, out=out)


def _convnext(prefix, d, out):
    out += [(f"{prefix}.layer_scale", (1, d, 1, 1), "ls"), (f"{prefix}.conv.weight", (d, 1, 5, 5), "dw"),
            (f"{prefix}.conv.bias", (d,), "bias"), (f"{prefix}.norm.weight", (d,), "ln_w"), (f"{prefix}.norm.bias", (d,), "ln_b"),
            (f"{prefix}.mlp.0.weight", (2 * d, d), "lin"), (f"{prefix}.mlp.0.bias", (2 * d,), "bias"),
            (f"{prefix}.mlp.2.weight", (d, 2 * d), "lin"), (f"{prefix}.mlp.2.bias", (d,), "bias")]


def _dcb4(prefix, cin, cout, out):
    """DepthConvBlock4 (blocks/dcvc.py:57-66)"""
    p = f"{prefix}.block.0"
    out += [(f"{p}.conv1.0.weight", (cin, cin, 1, 1), "lin"), (f"{p}.conv1.0.bias", (cin,), "bias"),
            (f"{p}.depth_conv.weight", (cin, 1, 3, 3), "dw"), (f"{p}.depth_conv.bias", (cin,), "bias"),
            (f"{p}.conv2.weight", (cout, cin, 1, 1), "lin"), (f"{p}.conv2.bias", (cout,), "bias")]
    if cin != cout:
        out += [(f"{p}.adaptor.weight", (cout, cin, 1, 1), "lin"), (f"{p}.adaptor.bias", (cout,), "bias")]
    p = f"{prefix}.block.1"
    out += [(f"{p}.conv.weight", (4 * cout, cout, 1, 1), "lin"), (f"{p}.conv.bias", (4 * cout,), "bias"),
            (f"{p}.conv_out.weight", (cout, 2 * cout, 1, 1), "lin"), (f"{p}.conv_out.bias", (cout,), "bias")]


def _cross(prefix, W, F, n_attn, P, T, out):
    out += [(f"{prefix}.titok_pos_emb", (P * P + T + 1, 1, W), "emb"), (f"{prefix}.feat_pos_emb", (P * P, 1, F), "emb"),
            (f"{prefix}.titok_compress_proj.weight", (F, W), "lin"), (f"{prefix}.titok_compress_proj.bias", (F,), "bias")]
    for j in range(n_attn):
        _rab(f"{prefix}.attn.{j}", F, out)
    out += [(f"{prefix}.titok_decompress_proj.0.weight", (2 * F, F), "lin"), (f"{prefix}.titok_decompress_proj.0.bias", (2 * F,), "bias"),
            (f"{prefix}.titok_decompress_proj.1.weight", (2 * F,), "ln_w"), (f"{prefix}.titok_decompress_proj.1.bias", (2 * F,), "ln_b"),
            (f"{prefix}.feat_add.0.weight", (F,), "ln_w"), (f"{prefix}.feat_add.0.bias", (F,), "ln_b"),
            (f"{prefix}.feat_add.1.weight", (F, F), "lin"), (f"{prefix}.feat_add.1.bias", (F,), "bias"),
            (f"{prefix}.zero_add.weight", (W, 2 * F), "lin"), (f"{prefix}.zero_add.bias", (W,), "bias")]


def encoder_spec(cfg: CodecConfig, prefix="hybrid_codec.encoder"):
    """HybridEncoder (codec_sq_fixbpp.py:48-96 + titok/blocks.py:72-115)"""
    W, F, P, T, g = cfg.width, cfg.feat_dim, cfg.patch_size, cfg.num_latent_tokens, cfg.grid
    o = [(f"{prefix}.patch_embed.weight", (W, 3, P, P), "lin"), (f"{prefix}.patch_embed.bias", (W,), "bias"),
         (f"{prefix}.class_embedding", (1, W), "emb"), (f"{prefix}.positional_embedding", (g * g + 1, W), "emb"),
         (f"{prefix}.latent_token_positional_embedding", (T, W), "emb"),
         (f"{prefix}.ln_pre.weight", (W,), "ln_w"), (f"{prefix}.ln_pre.bias", (W,), "ln_b")]
    for i in range(cfg.layers):
        _rab(f"{prefix}.transformer.{i}", W, o)
    o += [(f"{prefix}.ln_post.weight", (W,), "ln_w"), (f"{prefix}.ln_post.bias", (W,), "ln_b"),
          (f"{prefix}.conv_out.weight", (cfg.token_size, W, 1, 1), "lin"), (f"{prefix}.conv_out.bias", (cfg.token_size,), "bias"),
          (f"{prefix}.pix_emb_proj.weight", (F, W, 1, 1), "lin"), (f"{prefix}.pix_emb_proj.bias", (F,), "bias")]
    for i in cfg.in_pos_enc:
        _cross(f"{prefix}.inter_blocks.{i}", W, F, cfg.n_attn, g, T, o)
        _swin_stack(f"{prefix}.feat_blocks.{i}.0", F, 2, cfg.window, o)
        _convnext(f"{prefix}.feat_blocks.{i}.1", F, o)
        _convnext(f"{prefix}.feat_blocks.{i}.2", F, o)
    _swin_stack(f"{prefix}.feat_in", F, 4, cfg.window, o)
    _swin_stack(f"{prefix}.feat_out.0", F, 2, cfg.window, o)
    o += [(f"{prefix}.feat_out.1.weight", (F, F, 2, 2), "lin"), (f"{prefix}.feat_out.1.bias", (F,), "bias"),
          (f"{prefix}.feat_out.3.weight", (F,), "ln_w"), (f"{prefix}.feat_out.3.bias", (F,), "ln_b"),
          (f"{prefix}.feat_out.4.weight", (F, F), "lin"), (f"{prefix}.feat_out.4.bias", (F,), "bias")]
    return o


def decoder_spec(cfg: CodecConfig, prefix="hybrid_codec.decoder"):
    """HybridDecoder (codec_sq_fixbpp.py:186-246 + titok/blocks.py:147-190; TiTok's ffn is removed, :195)"""
    W, F, P, T, g = cfg.width, cfg.feat_dim, cfg.patch_size, cfg.num_latent_tokens, cfg.grid
    o = [(f"{prefix}.decoder_embed.weight", (W, cfg.token_size), "lin"), (f"{prefix}.decoder_embed.bias", (W,), "bias"),
         (f"{prefix}.class_embedding", (1, W), "emb"), (f"{prefix}.positional_embedding", (g * g + 1, W), "emb"),
         (f"{prefix}.mask_token", (1, 1, W), "emb"), (f"{prefix}.latent_token_positional_embedding", (T, W), "emb"),
         (f"{prefix}.ln_pre.weight", (W,), "ln_w"), (f"{prefix}.ln_pre.bias", (W,), "ln_b")]
    for i in range(cfg.layers):
        _rab(f"{prefix}.transformer.{i}", W, o)
    o += [(f"{prefix}.ln_post.weight", (W,), "ln_w"), (f"{prefix}.ln_post.bias", (W,), "ln_b"),
          (f"{prefix}.init_feat_up.0.weight", (4 * F, F, 1, 1), "lin"), (f"{prefix}.init_feat_up.0.bias", (4 * F,), "bias")]
    _swin_stack(f"{prefix}.init_feat_up.2", F, 4, cfg.window, o)
    for i in cfg.in_pos_dec:
        _cross(f"{prefix}.inter_blocks.{i}", W, F, cfg.n_attn, g, T, o)
        _swin_stack(f"{prefix}.feat_blocks.{i}.0", F, 2, cfg.window, o)
        _convnext(f"{prefix}.feat_blocks.{i}.1", F, o)
        _convnext(f"{prefix}.feat_blocks.{i}.2", F, o)
    return o


def featmerge_spec(cfg: CodecConfig, prefix="prior_fusion"):
    """FeatMerge (codec_sq_fixbpp.py:395-425)"""
    W, F, I = cfg.width, cfg.feat_dim, cfg.fm_inner
    o = []
    _swin_stack(f"{prefix}.feat_in.1", F, 2, cfg.window, o, first_index=0)
    _swin_stack(f"{prefix}.titok_in.1", W, 2, cfg.window, o, first_index=0)
    o += [(f"{prefix}.merge.0.weight", (2 * W, W + F), "lin"), (f"{prefix}.merge.0.bias", (2 * W,), "bias"),
          (f"{prefix}.merge.1.weight", (2 * W,), "ln_w"), (f"{prefix}.merge.1.bias", (2 * W,), "ln_b"),
          (f"{prefix}.merge.3.weight", (I, 2 * W), "lin"), (f"{prefix}.merge.3.bias", (I,), "bias")]
    _swin_stack(f"{prefix}.merge.4", I, 4, cfg.window, o, first_index=0)
    o += [(f"{prefix}.ffn.0.weight", (I,), "ln_w"), (f"{prefix}.ffn.0.bias", (I,), "ln_b"),
          (f"{prefix}.ffn.1.weight", (2 * I, I), "lin"), (f"{prefix}.ffn.1.bias", (2 * I,), "bias"),
          (f"{prefix}.ffn.3.weight", (cfg.vq_n_embed, 2 * I), "lin"), (f"{prefix}.ffn.3.bias", (cfg.vq_n_embed,), "bias")]
    return o


def _vq_res(prefix, cin, cout, o):
    o += [(f"{prefix}.norm1.weight", (cin,), "ln_w"), (f"{prefix}.norm1.bias", (cin,), "ln_b"),
          (f"{prefix}.conv1.weight", (cout, cin, 3, 3), "lin"), (f"{prefix}.conv1.bias", (cout,), "bias"),
          (f"{prefix}.norm2.weight", (cout,), "ln_w"), (f"{prefix}.norm2.bias", (cout,), "ln_b"),
          (f"{prefix}.conv2.weight", (cout, cout, 3, 3), "lin"), (f"{prefix}.conv2.bias", (cout,), "bias")]
    if cin != cout:
        o += [(f"{prefix}.nin_shortcut.weight", (cout, cin, 1, 1), "lin"), (f"{prefix}.nin_shortcut.bias", (cout,), "bias")]


def _vq_attn(prefix, c, o):
    o += [(f"{prefix}.norm.weight", (c,), "ln_w"), (f"{prefix}.norm.bias", (c,), "ln_b")]
    for n in ("q", "k", "v", "proj_out"):
        o += [(f"{prefix}.{n}.weight", (c, c, 1, 1), "lin"), (f"{prefix}.{n}.bias", (c,), "bias")]


def vqgan_plan(cfg: CodecConfig):
    """level-by-level plan of the taming Decoder (model.py:436-537): list of (i_level, [(cin, cout, attn)], upsample)"""
    nres = len(cfg.vq_ch_mult)
    block_in = cfg.vq_ch * cfg.vq_ch_mult[-1]
    res = cfg.crop_size // 2 ** (nres - 1)
    plan = []
    for lvl in reversed(range(nres)):
        out = cfg.vq_ch * cfg.vq_ch_mult[lvl]
        blocks = []
        for _ in range(cfg.vq_num_res_blocks + 1):
            blocks.append((block_in, out, res in cfg.vq_attn_resolutions))
            block_in = out
        plan.append((lvl, blocks, lvl != 0))
        if lvl != 0:
            res *= 2
    return plan, cfg.vq_ch * cfg.vq_ch_mult[-1], block_in


def vqgan_spec(cfg: CodecConfig, prefix="vqgan"):
    """the parts of the taming VQGAN the decoder path uses: codebook, post_quant_conv, Decoder"""
    plan, c0, c_last = vqgan_plan(cfg)
    o = [(f"{prefix}.quantize.embedding.weight", (cfg.vq_n_embed, cfg.vq_embed_dim), "codebook"),
         (f"{prefix}.post_quant_conv.weight", (cfg.vq_z_channels, cfg.vq_embed_dim, 1, 1), "lin"),
         (f"{prefix}.post_quant_conv.bias", (cfg.vq_z_channels,), "bias"),
         (f"{prefix}.decoder.conv_in.weight", (c0, cfg.vq_z_channels, 3, 3), "lin"), (f"{prefix}.decoder.conv_in.bias", (c0,), "bias")]
    _vq_res(f"{prefix}.decoder.mid.block_1", c0, c0, o)
    _vq_attn(f"{prefix}.decoder.mid.attn_1", c0, o)
    _vq_res(f"{prefix}.decoder.mid.block_2", c0, c0, o)
    for lvl, blocks, up in plan:
        for i, (cin, cout, attn) in enumerate(blocks):
            _vq_res(f"{prefix}.decoder.up.{lvl}.block.{i}", cin, cout, o)
            if attn:
                _vq_attn(f"{prefix}.decoder.up.{lvl}.attn.{i}", cout, o)
        if up:
            c = blocks[-1][1]
            o += [(f"{prefix}.decoder.up.{lvl}.upsample.conv.weight", (c, c, 3, 3), "lin"),
                  (f"{prefix}.decoder.up.{lvl}.upsample.conv.bias", (c,), "bias")]
    o += [(f"{prefix}.decoder.norm_out.weight", (c_last,), "ln_w"), (f"{prefix}.decoder.norm_out.bias", (c_last,), "ln_b"),
          (f"{prefix}.decoder.conv_out.weight", (3, c_last, 3, 3), "lin"), (f"{prefix}.decoder.conv_out.bias", (3,), "bias")]
    return o


def full_spec(cfg: CodecConfig):
    return (encoder_spec(cfg) + codec_misc_spec(cfg) + bottleneck_spec(cfg) + decoder_spec(cfg) + featmerge_spec(cfg) +
            vqgan_spec(cfg))


def codec_misc_spec(cfg: CodecConfig, prefix="hybrid_codec"):
    """latent tokens + TiTok codebook (codec_sq_fixbpp.py:311-323)"""
    return [(f"{prefix}.latent_tokens", (cfg.num_latent_tokens, cfg.width), "emb"),
            (f"{prefix}.quantize.embedding.weight", (cfg.codebook_size, cfg.token_size), "codebook")]


def bottleneck_spec(cfg: CodecConfig, prefix="hybrid_codec.quantize_feat"):
    """Compressive_bottleneck_varbpp_type2 (models/sq_bottleneck.py:55-100), bpp_num = 1"""
    F, Q = cfg.feat_dim, cfg.embed_dim
    o = [(f"{prefix}.enc_q", (1, F, 1, 1), "qscale"), (f"{prefix}.dec_q", (1, F, 1, 1), "qscale")]
    _dcb4(f"{prefix}.enc_trans_0.0", F, F, o)
    _dcb4(f"{prefix}.enc_trans_0.1", F, F, o)
    _dcb4(f"{prefix}.enc_trans_1.0", F, F, o)
    _dcb4(f"{prefix}.enc_trans_1.1", F, Q, o)
    _dcb4(f"{prefix}.dec_trans_0.0", Q, F, o)
    _dcb4(f"{prefix}.dec_trans_0.1", F, F, o)
    _dcb4(f"{prefix}.dec_trans_1.0", F, F, o)
    _dcb4(f"{prefix}.dec_trans_1.1", F, F, o)
    o += [(f"{prefix}.factorized_prior_vec", (1, Q, 1, 1), "prior_vec")]
    _dcb4(f"{prefix}.y_prior_fusion.0", Q, 2 * Q, o)
    _dcb4(f"{prefix}.y_prior_fusion.1", 2 * Q, 3 * Q, o)
    o += [(f"{prefix}.y_spatial_prior_reduction.weight", (Q, 3 * Q, 1, 1), "lin"),
          (f"{prefix}.y_spatial_prior_reduction.bias", (Q,), "bias")]
    for k in (1, 2, 3):
        _dcb4(f"{prefix}.y_spatial_prior_adaptor_{k}", 2 * Q, 2 * Q, o)
    for k in range(3):
        _dcb4(f"{prefix}.y_spatial_prior.{k}", 2 * Q, 2 * Q, o)
    return o


def clip_spec(cfg: ClipConfig, prefix="clip.visual"):
    """open_clip VisionTransformer state_dict names (image tower only)"""
    W, g = cfg.width, cfg.image_size // cfg.patch
    o = [(f"{prefix}.conv1.weight", (W, 3, cfg.patch, cfg.patch), "lin"), (f"{prefix}.class_embedding", (W,), "emb"),
         (f"{prefix}.positional_embedding", (g * g + 1, W), "emb"),
         (f"{prefix}.ln_pre.weight", (W,), "ln_w"), (f"{prefix}.ln_pre.bias", (W,), "ln_b")]
    for i in range(cfg.layers):
        _rab(f"{prefix}.transformer.resblocks.{i}", W, o)
    o += [(f"{prefix}.ln_post.weight", (W,), "ln_w"), (f"{prefix}.ln_post.bias", (W,), "ln_b"),
          (f"{prefix}.proj", (W, cfg.embed_dim), "proj")]
    return o


def clip_text_spec(cfg: ClipConfig, prefix="clip"):
    """open_clip CLIP text-tower state_dict names (token_embedding, positional_embedding, transformer,
    ln_final, text_projection)"""
    W = cfg.t_width
    o = [(f"{prefix}.token_embedding.weight", (cfg.vocab, W), "emb"), (f"{prefix}.positional_embedding", (cfg.ctx, W), "emb")]
    for i in range(cfg.t_layers):
        _rab(f"{prefix}.transformer.resblocks.{i}", W, o)
    o += [(f"{prefix}.ln_final.weight", (W,), "ln_w"), (f"{prefix}.ln_final.bias", (W,), "ln_b"),
          (f"{prefix}.text_projection", (W, cfg.embed_dim), "proj")]
    return o


def create_mask(window_size, displacement, upper_lower, left_right):
    """the constant -inf shift masks the reference stores as Parameters (swin_transformer.py:42-55)"""
    w = window_size
    mask = torch.zeros(w * w, w * w)
    if upper_lower:
        mask[-displacement * w:, :-displacement * w] = float("-inf")
        mask[:-displacement * w, -displacement * w:] = float("-inf")
    if left_right:
        m = mask.view(w, w, w, w)
        m[:, -displacement:, :, :-displacement] = float("-inf")
        m[:, :-displacement, :, -displacement:] = float("-inf")
    return mask


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()), seed]))


def synth_weights(spec, seed=1234, window=16):
    """name -> fp32 CPU tensor; every tensor depends only on (name, seed) via numpy's Philox stream."""
    sd = {}
    for name, shape, kind in spec:
        r = _rng(name, seed)
        n = int(np.prod(shape))

        def normal(std):
            return torch.from_numpy(r.standard_normal(n, dtype=np.float32).reshape(shape)) * std

        if kind == "lin":
            fan_in = int(np.prod(shape[1:]))
            # the DepthConvBlock4 stacks of the bottleneck are residual-on-residual: damp them so the
            # synthetic prior produces sigma in roughly 0.05..20 and symbols of a few tens (some bypass)
            gain = 0.75 if "quantize_feat" in name else 1.0
            t = normal(gain / math.sqrt(fan_in))
        elif kind == "proj":
            t = normal(1.0 / math.sqrt(shape[0]))
        elif kind == "bias":
            t = normal(0.02)
        elif kind == "ln_w":
            t = 1.0 + normal(0.1)
        elif kind == "ln_b":
            t = normal(0.05)
        elif kind == "emb":
            t = normal(shape[-1] ** -0.5)
        elif kind == "posbias":
            t = normal(0.5)
        elif kind == "ls":
            t = 1.0 + normal(0.1)
        elif kind == "dw":
            t = normal(1.0 / math.sqrt(shape[-1] * shape[-2]))
        elif kind == "codebook":
            t = normal(1.0)
        elif kind == "qscale":
            t = 1.0 + normal(0.1)
        elif kind == "prior_vec":
            t = 1.0 + normal(0.3)
        elif kind == "mask_ul":
            t = create_mask(window, window // 2, True, False)
        elif kind == "mask_lr":
            t = create_mask(window, window // 2, False, True)
        else:
            raise ValueError(kind)
        sd[name] = t.contiguous().float()
    return sd
