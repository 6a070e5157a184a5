"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the floating-point transforms on the
compress path (the "plain PyTorch fp32 reference" of the HIP kernels and the `port` CPU baseline of
bench.py).  Functional style over a {state_dict key: tensor} dict with the reference's key names.

Parity status: PINNED -- oracle/gen_golden_nn.py runs these functions side by side with the REAL
reference modules (imported from /root/reference under stubs, synthetic weights) and commits the
reference outputs as tests/golden/nn_small_*.npz; tests/test_oracle_nn.py re-checks this file against
those fixtures on every run.

Each function cites the reference lines it restates (paths relative to /root/reference/src).
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------
def mha(x, sd, p, heads, mask=None):
    """nn.MultiheadAttention self-attention, (L, N, D) layout (titok/blocks.py:50-54); mask: additive (L, L)"""
    L, N, D = x.shape
    hd = D // heads
    qkv = F.linear(x, sd[f"{p}.in_proj_weight"], sd[f"{p}.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(L, N * heads, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.reshape(L, N * heads, hd).transpose(0, 1)
    v = v.reshape(L, N * heads, hd).transpose(0, 1)
    s = q @ k.transpose(1, 2)
    if mask is not None:
        s = s + mask
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(0, 1).reshape(L, N, D)
    return F.linear(a, sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])


def rab(x, sd, p, heads, mask=None):
    """ResidualAttentionBlock (titok/blocks.py:56-64)"""
    D = x.shape[-1]
    x = x + mha(F.layer_norm(x, (D,), sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"]), sd, f"{p}.attn", heads, mask)
    h = F.layer_norm(x, (D,), sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"])
    h = F.linear(F.gelu(F.linear(h, sd[f"{p}.mlp.c_fc.weight"], sd[f"{p}.mlp.c_fc.bias"])),
                 sd[f"{p}.mlp.c_proj.weight"], sd[f"{p}.mlp.c_proj.bias"])
    return x + h


def rel_indices(win):
    idx = torch.tensor([[x, y] for x in range(win) for y in range(win)])
    return idx[None, :, :] - idx[:, None, :] + win - 1  # (w2, w2, 2)


def swin_block(x, sd, p, shifted, rel, win=16):
    """SwinBlock on (B, H, W, C) (blocks/swin_transformer.py:94-156)"""
    B, H, W, C = x.shape
    heads = C // 64
    idn = x
    h = F.layer_norm(x, (C,), sd[f"{p}.norm_attn.weight"], sd[f"{p}.norm_attn.bias"])
    a = f"{p}.attention_block"
    if shifted:
        h = torch.roll(h, shifts=(-(win // 2), -(win // 2)), dims=(1, 2))
    qkv = F.linear(h, sd[f"{a}.to_qkv.weight"]).chunk(3, dim=-1)
    nh, nw = H // win, W // win

    def split(t):  # b (nh wh) (nw ww) (h d) -> b h (nh nw) (wh ww) d
        return t.reshape(B, nh, win, nw, win, heads, 64).permute(0, 5, 1, 3, 2, 4, 6).reshape(B, heads, nh * nw, win * win, 64)

    q, k, v = map(split, qkv)
    dots = torch.einsum("bhwid,bhwjd->bhwij", q, k) * (64 ** -0.5)
    if rel:
        ri = rel_indices(win)
        dots = dots + sd[f"{a}.pos_embedding"][ri[:, :, 0], ri[:, :, 1]]
    else:
        dots = dots + sd[f"{a}.pos_embedding"]
    if shifted:
        dots[:, :, -nw:] += sd[f"{a}.upper_lower_mask"]
        dots[:, :, nw - 1::nw] += sd[f"{a}.left_right_mask"]
    out = torch.einsum("bhwij,bhwjd->bhwid", dots.softmax(dim=-1), v)
    out = out.reshape(B, heads, nh, nw, win, win, 64).permute(0, 2, 4, 3, 5, 1, 6).reshape(B, H, W, C)
    out = F.linear(out, sd[f"{a}.to_out.weight"], sd[f"{a}.to_out.bias"])
    if shifted:
        out = torch.roll(out, shifts=(win // 2, win // 2), dims=(1, 2))
    x = out + idn
    h = F.layer_norm(x, (C,), sd[f"{p}.norm_mlp.weight"], sd[f"{p}.norm_mlp.bias"])
    h = F.linear(F.gelu(F.linear(h, sd[f"{p}.mlp_block.net.0.weight"], sd[f"{p}.mlp_block.net.0.bias"])),
                 sd[f"{p}.mlp_block.net.2.weight"], sd[f"{p}.mlp_block.net.2.bias"])
    return x + h


def swin_stack(x_bchw, sd, p, n, win=16, first_index=1, bchw=True):
    """get_swin (models/codec_sq_fixbpp.py:33-45)"""
    x = x_bchw.permute(0, 2, 3, 1).contiguous() if bchw else x_bchw
    for i in range(n):
        x = swin_block(x, sd, f"{p}.{first_index + i}", shifted=bool(i % 2), rel=(i == 0), win=win)
    return x.permute(0, 3, 1, 2).contiguous() if bchw else x


def convnext(x, sd, p):
    """ConvNeXtBlock, (B,C,H,W) (blocks/conv_blocks.py:71-81)"""
    B, C, H, W = x.shape
    h = x * sd[f"{p}.layer_scale"]
    h = F.conv2d(h, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], padding=2, groups=C)
    h = h.permute(0, 2, 3, 1)
    h = F.layer_norm(h, (C,), sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"])
    h = F.linear(F.gelu(F.linear(h, sd[f"{p}.mlp.0.weight"], sd[f"{p}.mlp.0.bias"])), sd[f"{p}.mlp.2.weight"], sd[f"{p}.mlp.2.bias"])
    return h.permute(0, 3, 1, 2) + x


def dcb4(x, sd, p):
    """DepthConvBlock4 = DepthConv + ConvFFN3, (B,C,H,W) (blocks/dcvc.py:14-66)"""
    q = f"{p}.block.0"
    idn = x
    if f"{q}.adaptor.weight" in sd:
        idn = F.conv2d(x, sd[f"{q}.adaptor.weight"], sd[f"{q}.adaptor.bias"])
    h = F.leaky_relu(F.conv2d(x, sd[f"{q}.conv1.0.weight"], sd[f"{q}.conv1.0.bias"]), 0.01)
    h = F.conv2d(h, sd[f"{q}.depth_conv.weight"], sd[f"{q}.depth_conv.bias"], padding=1, groups=h.shape[1])
    x = F.conv2d(h, sd[f"{q}.conv2.weight"], sd[f"{q}.conv2.bias"]) + idn
    q = f"{p}.block.1"
    x1, x2 = F.conv2d(x, sd[f"{q}.conv.weight"], sd[f"{q}.conv.bias"]).chunk(2, 1)
    return x + F.conv2d(F.leaky_relu(x1, 0.1) + F.leaky_relu(x2, 0.01), sd[f"{q}.conv_out.weight"], sd[f"{q}.conv_out.bias"])


def cross_block(feat, x, stack_shape, sd, p, cfg):
    """Interactive_crossAttn_type4.forward (models/cross_blocks.py:75-98); feat (B,F,H,W), x (L,N,W)"""
    nH, nW = stack_shape
    B, Fd, H, W = feat.shape
    P = cfg.grid
    fs = feat.reshape(B, Fd, nH, P, nW, P).permute(3, 5, 0, 2, 4, 1).reshape(P * P, B * nH * nW, Fd)
    fpos = sd[f"{p}.feat_pos_emb"] + fs
    tpos = F.linear(sd[f"{p}.titok_pos_emb"] + x, sd[f"{p}.titok_compress_proj.weight"], sd[f"{p}.titok_compress_proj.bias"])
    f = torch.cat([tpos, fpos], dim=0)
    for j in range(cfg.n_attn):
        f = rab(f, sd, f"{p}.attn.{j}", Fd // 64)
    f_feat_new, f_titok_new = f[-(P * P):], f[:-(P * P)]
    fs = fs + F.linear(F.layer_norm(f_feat_new, (Fd,), sd[f"{p}.feat_add.0.weight"], sd[f"{p}.feat_add.0.bias"]),
                       sd[f"{p}.feat_add.1.weight"], sd[f"{p}.feat_add.1.bias"])
    t = F.linear(f_titok_new, sd[f"{p}.titok_decompress_proj.0.weight"], sd[f"{p}.titok_decompress_proj.0.bias"])
    t = F.silu(F.layer_norm(t, (2 * Fd,), sd[f"{p}.titok_decompress_proj.1.weight"], sd[f"{p}.titok_decompress_proj.1.bias"]))
    x = x + F.linear(t, sd[f"{p}.zero_add.weight"], sd[f"{p}.zero_add.bias"])
    feat = fs.reshape(P, P, B, nH, nW, Fd).permute(2, 5, 3, 0, 4, 1).reshape(B, Fd, H, W)
    return feat, x


# ---------------------------------------------------------------------------------------------
# hybrid encoder + VQ
# ---------------------------------------------------------------------------------------------
def encoder_forward(x01, sd, cfg, p="hybrid_codec.encoder", taps=None):
    """HybridEncoder.forward (models/codec_sq_fixbpp.py:117-183).  x01 in [0,1], (B,3,256a,256b)."""
    Wd, Fd, P, T = cfg.width, cfg.feat_dim, cfg.grid, cfg.num_latent_tokens
    x_emb = F.conv2d(x01, sd[f"{p}.patch_embed.weight"], sd[f"{p}.patch_embed.bias"], stride=cfg.patch_size)
    feat = F.conv2d(x_emb, sd[f"{p}.pix_emb_proj.weight"], sd[f"{p}.pix_emb_proj.bias"])
    B0 = x_emb.shape[0]
    nH, nW = x_emb.shape[2] // P, x_emb.shape[3] // P
    x = x_emb.reshape(B0, Wd, nH, P, nW, P).permute(0, 2, 4, 1, 3, 5).reshape(B0 * nH * nW, Wd, P * P).permute(0, 2, 1)
    N = x.shape[0]
    x = torch.cat([sd[f"{p}.class_embedding"].unsqueeze(0).expand(N, -1, -1), x], dim=1) + sd[f"{p}.positional_embedding"]
    lat = sd["hybrid_codec.latent_tokens"].unsqueeze(0).expand(N, -1, -1) + sd[f"{p}.latent_token_positional_embedding"]
    x = torch.cat([x, lat], dim=1)
    feat = swin_stack(feat, sd, f"{p}.feat_in", 4, cfg.window)
    x = F.layer_norm(x, (Wd,), sd[f"{p}.ln_pre.weight"], sd[f"{p}.ln_pre.bias"]).permute(1, 0, 2)
    if taps is not None:
        taps["feat_in"] = feat.clone()
        taps["x_ln_pre"] = x.permute(1, 0, 2).clone()
    for i in range(cfg.layers):
        x = rab(x, sd, f"{p}.transformer.{i}", cfg.heads)
        if i in cfg.in_pos_enc:
            feat, x = cross_block(feat, x, (nH, nW), sd, f"{p}.inter_blocks.{i}", cfg)
            feat = swin_stack(feat, sd, f"{p}.feat_blocks.{i}.0", 2, cfg.window)
            feat = convnext(feat, sd, f"{p}.feat_blocks.{i}.1")
            feat = convnext(feat, sd, f"{p}.feat_blocks.{i}.2")
        if taps is not None and i == 0:
            taps["x_layer0"] = x.permute(1, 0, 2).clone()
    x = x.permute(1, 0, 2)
    lat = F.layer_norm(x[:, 1 + P * P:], (Wd,), sd[f"{p}.ln_post.weight"], sd[f"{p}.ln_post.bias"])
    lat = lat.reshape(N, Wd, T, 1)  # the reference's "fake 2-D" reinterpretation
    z = F.conv2d(lat, sd[f"{p}.conv_out.weight"], sd[f"{p}.conv_out.bias"]).reshape(N, cfg.token_size, 1, T)
    h = swin_stack(feat, sd, f"{p}.feat_out.0", 2, cfg.window)
    h = F.conv2d(h, sd[f"{p}.feat_out.1.weight"], sd[f"{p}.feat_out.1.bias"], stride=2).permute(0, 2, 3, 1)
    h = F.linear(F.layer_norm(h, (Fd,), sd[f"{p}.feat_out.3.weight"], sd[f"{p}.feat_out.3.bias"]),
                 sd[f"{p}.feat_out.4.weight"], sd[f"{p}.feat_out.4.bias"]).permute(0, 3, 1, 2).contiguous()
    return z, h, (nH, nW)


def vq_indices(z, sd, p="hybrid_codec.quantize"):
    """VectorQuantizer.forward -> min_encoding_indices (titok/quantizer.py:46-61), l2-norm variant"""
    zf = z.permute(0, 2, 3, 1).reshape(-1, z.shape[1])
    zf = F.normalize(zf, dim=-1)
    e = F.normalize(sd[f"{p}.embedding.weight"], dim=-1)
    d = (zf ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * zf @ e.t()
    return d.argmin(dim=1)


# ---------------------------------------------------------------------------------------------
# bottleneck: analysis transform + 4-step prior (write path)
# ---------------------------------------------------------------------------------------------
def four_part_masks(B, C, H, W):
    """get_mask_four_parts (entropy/compression_model.py:241-283)"""
    def micro(a, b):
        m = torch.zeros(H, W)
        m[a::2, b::2] = 1
        return m[None, None]
    m0, m1, m2, m3 = micro(0, 0), micro(0, 1), micro(1, 0), micro(1, 1)
    o = torch.ones(B, C // 4, H, W)
    return (torch.cat((o * m0, o * m1, o * m2, o * m3), 1), torch.cat((o * m3, o * m2, o * m1, o * m0), 1),
            torch.cat((o * m2, o * m3, o * m0, o * m1), 1), torch.cat((o * m1, o * m0, o * m3, o * m2), 1))


def bottleneck_analysis(h, sd, p="hybrid_codec.quantize_feat"):
    """get_qp + encode (models/sq_bottleneck.py:102-113)"""
    y = dcb4(dcb4(h, sd, f"{p}.enc_trans_0.0"), sd, f"{p}.enc_trans_0.1")
    y = y * sd[f"{p}.enc_q"][0:1]
    return dcb4(dcb4(y, sd, f"{p}.enc_trans_1.0"), sd, f"{p}.enc_trans_1.1")


def prior_params(B, H, W, sd, p="hybrid_codec.quantize_feat"):
    q = sd[f"{p}.factorized_prior_vec"][0:1].repeat(B, 1, H, W)
    return dcb4(dcb4(q, sd, f"{p}.y_prior_fusion.0"), sd, f"{p}.y_prior_fusion.1")


def four_part_prior_write(y, sd, thr, p="hybrid_codec.quantize_feat"):
    """forward_four_part_prior(write=True) (entropy/compression_model.py:303-366) followed by
    GaussianEncoder.build_indexes + symbol clamp (entropy_models.py:355-362,66-69).
    Returns (symbols (B,4,16,H,W) int16, indexes int16, y_hat, per-step (scales, means) list)."""
    B, C, H, W = y.shape
    params = prior_params(B, H, W, sd, p)
    qs, scales, means = params.chunk(3, 1)
    common = F.conv2d(params, sd[f"{p}.y_spatial_prior_reduction.weight"], sd[f"{p}.y_spatial_prior_reduction.bias"])
    masks = four_part_masks(B, C, H, W)
    qs = torch.clamp_min(qs, 0.5)
    y = y / qs
    log_min, log_step = math.log(0.11), (math.log(64.0) - math.log(0.11)) / 255
    syms, idxs, sms = [], [], []
    y_hat_so_far = None
    for k in range(4):
        if k > 0:
            t = dcb4(torch.cat((y_hat_so_far, common), 1), sd, f"{p}.y_spatial_prior_adaptor_{k}")
            for j in range(3):
                t = dcb4(t, sd, f"{p}.y_spatial_prior.{j}")
            scales, means = t.chunk(2, 1)
        sms.append((scales.clone(), means.clone()))
        m = masks[k]
        s_hat, m_hat = scales * m, means * m
        y_q = torch.round((y - m_hat) * m)
        cond = s_hat < thr
        y_q = torch.where(cond, torch.zeros_like(y_q), y_q)
        s_hat = torch.where(cond, torch.zeros_like(s_hat), s_hat)
        y_hat = y_q + m_hat
        y_hat_so_far = y_hat if y_hat_so_far is None else y_hat_so_far + y_hat
        x0, x1, x2, x3 = y_q.chunk(4, 1)
        yq_w = (x0 + x1) + (x2 + x3)
        x0, x1, x2, x3 = s_hat.chunk(4, 1)
        sc_w = (x0 + x1) + (x2 + x3)
        sc = torch.maximum(sc_w, torch.zeros_like(sc_w) + 1e-5)
        ind = ((torch.log(sc) - log_min) / log_step).clamp_(0, 255)
        ind = torch.where(sc_w < thr, torch.zeros_like(ind) - 1, ind).int()
        syms.append(yq_w.clamp(-30000, 30000).to(torch.int16))
        idxs.append(ind.to(torch.int16))
    return torch.stack(syms, 1), torch.stack(idxs, 1), y_hat_so_far * qs, sms


# ---------------------------------------------------------------------------------------------
# CLIP ViT image tower (open_clip VisionTransformer; compress.py:69-74)
# ---------------------------------------------------------------------------------------------
def clip_tower(x, sd, cfg, p="clip.visual"):
    """x: (B,3,224,224) normalised -> unit-norm (B, embed_dim)"""
    Wd = cfg.width
    x = F.conv2d(x, sd[f"{p}.conv1.weight"], None, stride=cfg.patch)
    B = x.shape[0]
    x = x.reshape(B, Wd, -1).permute(0, 2, 1)
    x = torch.cat([sd[f"{p}.class_embedding"].reshape(1, 1, Wd).expand(B, -1, -1), x], dim=1) + sd[f"{p}.positional_embedding"]
    x = F.layer_norm(x, (Wd,), sd[f"{p}.ln_pre.weight"], sd[f"{p}.ln_pre.bias"]).permute(1, 0, 2)
    for i in range(cfg.layers):
        x = rab(x, sd, f"{p}.transformer.resblocks.{i}", cfg.heads)
    x = x.permute(1, 0, 2)
    pooled = F.layer_norm(x[:, 0], (Wd,), sd[f"{p}.ln_post.weight"], sd[f"{p}.ln_post.bias"])
    z = pooled @ sd[f"{p}.proj"]
    return z / z.norm(dim=-1, keepdim=True)


def clip_text_tower(tokens, sd, cfg, p="clip"):
    """open_clip CLIP.encode_text as search.py:93-97 calls it (third-party, restated from its published definition):
    tokens (B, ctx) int -> unit-norm (B, embed_dim).  Pinned against the independent HuggingFace `transformers` CLIP
    implementation (same synthetic weights) by tests/test_oracle_clip_hf.py."""
    Wd, L = cfg.t_width, cfg.ctx
    tokens = torch.as_tensor(tokens).long()
    x = sd[f"{p}.token_embedding.weight"][tokens] + sd[f"{p}.positional_embedding"]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    x = x.permute(1, 0, 2)
    for i in range(cfg.t_layers):
        x = rab(x, sd, f"{p}.transformer.resblocks.{i}", cfg.t_heads, mask)
    x = F.layer_norm(x.permute(1, 0, 2), (Wd,), sd[f"{p}.ln_final.weight"], sd[f"{p}.ln_final.bias"])
    z = x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd[f"{p}.text_projection"]
    return z / z.norm(dim=-1, keepdim=True)


# ---------------------------------------------------------------------------------------------
# decode side (D1-D5)
# ---------------------------------------------------------------------------------------------
def four_part_prior_decode(sym, sd, thr, B, H, W, p="hybrid_codec.quantize_feat"):
    """decompress_four_part_prior (entropy/compression_model.py:377-418) given the already entropy-decoded
    symbols sym (B,4,16,H,W) (0 where skipped).  Also returns the per-step indexes it would request."""
    C = sd[f"{p}.factorized_prior_vec"].shape[1]
    params = prior_params(B, H, W, sd, p)
    qs, scales, means = params.chunk(3, 1)
    common = F.conv2d(params, sd[f"{p}.y_spatial_prior_reduction.weight"], sd[f"{p}.y_spatial_prior_reduction.bias"])
    masks = four_part_masks(B, C, H, W)
    qs = torch.clamp_min(qs, 0.5)
    log_min, log_step = math.log(0.11), (math.log(64.0) - math.log(0.11)) / 255
    y_hat_so_far, idxs = None, []
    for k in range(4):
        if k > 0:
            t = dcb4(torch.cat((y_hat_so_far, common), 1), sd, f"{p}.y_spatial_prior_adaptor_{k}")
            for j in range(3):
                t = dcb4(t, sd, f"{p}.y_spatial_prior.{j}")
            scales, means = t.chunk(2, 1)
        m = masks[k]
        x0, x1, x2, x3 = (scales * m).chunk(4, 1)
        sc_r = (x0 + x1) + (x2 + x3)
        sc = torch.maximum(sc_r, torch.zeros_like(sc_r) + 1e-5)
        ind = ((torch.log(sc) - log_min) / log_step).clamp_(0, 255)
        idxs.append(torch.where(sc_r < thr, torch.zeros_like(ind) - 1, ind).to(torch.int16))
        yq = sym[:, k].float()
        cur = (torch.cat((yq, yq, yq, yq), 1) + means) * m
        y_hat_so_far = cur if y_hat_so_far is None else y_hat_so_far + cur
    return y_hat_so_far * qs, torch.stack(idxs, 1)


def bottleneck_synthesis(y_hat, sd, p="hybrid_codec.quantize_feat"):
    """decode (models/sq_bottleneck.py:115-119)"""
    h = dcb4(dcb4(y_hat, sd, f"{p}.dec_trans_0.0"), sd, f"{p}.dec_trans_0.1")
    h = h * sd[f"{p}.dec_q"][0:1]
    return dcb4(dcb4(h, sd, f"{p}.dec_trans_1.0"), sd, f"{p}.dec_trans_1.1")


def z_from_indices(idx, n_tiles, sd, cfg, p="hybrid_codec.quantize"):
    """decode_only's z branch (codec_sq_fixbpp.py:889-892): codebook rows -> (N,12,1,32), l2-normalised over c"""
    e = sd[f"{p}.embedding.weight"][idx.long()]                      # (N*T, c)
    z = e.reshape(n_tiles, cfg.num_latent_tokens, -1).permute(0, 2, 1).unsqueeze(2).contiguous()
    return F.normalize(z, dim=1)


def decoder_forward(z_hat, h_hat, stack_shape, sd, cfg, p="hybrid_codec.decoder"):
    """HybridDecoder.forward (models/codec_sq_fixbpp.py:248-300)"""
    nH, nW = stack_shape
    Wd, Fd, P, T = cfg.width, cfg.feat_dim, cfg.grid, cfg.num_latent_tokens
    N = z_hat.shape[0]
    x = z_hat.reshape(N, -1, T).permute(0, 2, 1)
    x = F.linear(x, sd[f"{p}.decoder_embed.weight"], sd[f"{p}.decoder_embed.bias"])
    mt = sd[f"{p}.mask_token"].repeat(N, P * P, 1)
    mt = torch.cat([sd[f"{p}.class_embedding"].unsqueeze(0).expand(N, -1, -1), mt], dim=1) + sd[f"{p}.positional_embedding"]
    x = torch.cat([mt, x + sd[f"{p}.latent_token_positional_embedding"][:T]], dim=1)
    feat = F.conv2d(h_hat, sd[f"{p}.init_feat_up.0.weight"], sd[f"{p}.init_feat_up.0.bias"])
    feat = F.pixel_shuffle(feat, 2)
    feat = swin_stack(feat, sd, f"{p}.init_feat_up.2", 4, cfg.window)
    x = F.layer_norm(x, (Wd,), sd[f"{p}.ln_pre.weight"], sd[f"{p}.ln_pre.bias"]).permute(1, 0, 2)
    for i in range(cfg.layers):
        x = rab(x, sd, f"{p}.transformer.{i}", cfg.heads)
        if i in cfg.in_pos_dec:
            feat, x = cross_block(feat, x, (nH, nW), sd, f"{p}.inter_blocks.{i}", cfg)
            feat = swin_stack(feat, sd, f"{p}.feat_blocks.{i}.0", 2, cfg.window)
            feat = convnext(feat, sd, f"{p}.feat_blocks.{i}.1")
            feat = convnext(feat, sd, f"{p}.feat_blocks.{i}.2")
    x = x.permute(1, 0, 2)[:, 1:1 + P * P]
    x = F.layer_norm(x, (Wd,), sd[f"{p}.ln_post.weight"], sd[f"{p}.ln_post.bias"])
    x = x.permute(0, 2, 1).reshape(N, Wd, P, P)
    B = N // (nH * nW)
    x = x.reshape(B, nH, nW, Wd, P, P).permute(0, 3, 1, 4, 2, 5).reshape(B, Wd, nH * P, nW * P)
    return x, feat


def featmerge_forward(titok, feat, sd, cfg, p="prior_fusion"):
    """FeatMerge.forward -> logits (B, n_embed, H, W) (models/codec_sq_fixbpp.py:427-439)"""
    t = swin_stack(titok.permute(0, 2, 3, 1).contiguous(), sd, f"{p}.titok_in.1", 2, cfg.window, first_index=0, bchw=False)
    f = swin_stack(feat.permute(0, 2, 3, 1).contiguous(), sd, f"{p}.feat_in.1", 2, cfg.window, first_index=0, bchw=False)
    h = F.linear(torch.cat([t, f], dim=-1), sd[f"{p}.merge.0.weight"], sd[f"{p}.merge.0.bias"])
    h = F.silu(F.layer_norm(h, (h.shape[-1],), sd[f"{p}.merge.1.weight"], sd[f"{p}.merge.1.bias"]))
    h = F.linear(h, sd[f"{p}.merge.3.weight"], sd[f"{p}.merge.3.bias"])
    h = swin_stack(h, sd, f"{p}.merge.4", 4, cfg.window, first_index=0, bchw=False)
    h = F.layer_norm(h, (h.shape[-1],), sd[f"{p}.ffn.0.weight"], sd[f"{p}.ffn.0.bias"])
    h = F.linear(torch.tanh(F.linear(h, sd[f"{p}.ffn.1.weight"], sd[f"{p}.ffn.1.bias"])), sd[f"{p}.ffn.3.weight"], sd[f"{p}.ffn.3.bias"])
    return h.permute(0, 3, 1, 2).contiguous()


def soft_lookup(logits, sd, p="vqgan"):
    """decode_to_latent's soft codebook lookup (codec_sq_fixbpp.py:660-662)"""
    return torch.einsum("nchw,cd->ndhw", logits.softmax(1), sd[f"{p}.quantize.embedding.weight"])


def _gn(x, sd, p):
    return F.group_norm(x, 32, sd[f"{p}.weight"], sd[f"{p}.bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def vq_resblock(x, sd, p):
    """taming ResnetBlock, temb = None, dropout 0 (model.py:117-137)"""
    h = F.conv2d(_swish(_gn(x, sd, f"{p}.norm1")), sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(h, sd, f"{p}.norm2")), sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.nin_shortcut.weight"], sd[f"{p}.nin_shortcut.bias"])
    return x + h


def vq_attnblock(x, sd, p):
    """taming AttnBlock (model.py:168-192)"""
    h = _gn(x, sd, f"{p}.norm")
    q = F.conv2d(h, sd[f"{p}.q.weight"], sd[f"{p}.q.bias"])
    k = F.conv2d(h, sd[f"{p}.k.weight"], sd[f"{p}.k.bias"])
    v = F.conv2d(h, sd[f"{p}.v.weight"], sd[f"{p}.v.bias"])
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, -1).permute(0, 2, 1), k.reshape(b, c, -1)) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2).permute(0, 2, 1)
    h = torch.bmm(v.reshape(b, c, -1), w_).reshape(b, c, hh, ww)
    return x + F.conv2d(h, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])


def vqgan_decode(latent, sd, cfg, p="vqgan"):
    """decode_to_image: post_quant_conv + taming Decoder.forward (codec_sq_fixbpp.py:666-669, model.py:506-537)"""
    import sgic_amd.weights as W  # plan helper only (pure python)
    plan, _, _ = W.vqgan_plan(cfg)
    d = f"{p}.decoder"
    h = F.conv2d(latent, sd[f"{p}.post_quant_conv.weight"], sd[f"{p}.post_quant_conv.bias"])
    h = F.conv2d(h, sd[f"{d}.conv_in.weight"], sd[f"{d}.conv_in.bias"], padding=1)
    h = vq_resblock(h, sd, f"{d}.mid.block_1")
    h = vq_attnblock(h, sd, f"{d}.mid.attn_1")
    h = vq_resblock(h, sd, f"{d}.mid.block_2")
    for lvl, blocks, up in plan:
        for i, (cin, cout, attn) in enumerate(blocks):
            h = vq_resblock(h, sd, f"{d}.up.{lvl}.block.{i}")
            if attn:
                h = vq_attnblock(h, sd, f"{d}.up.{lvl}.attn.{i}")
        if up:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"{d}.up.{lvl}.upsample.conv.weight"], sd[f"{d}.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _swish(_gn(h, sd, f"{d}.norm_out"))
    return F.conv2d(h, sd[f"{d}.conv_out.weight"], sd[f"{d}.conv_out.bias"], padding=1)
