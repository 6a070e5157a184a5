"""Per-block parity against the REAL reference: tests/golden/blocks.npz holds, for every block class of the
hot path instantiated on its own from the imported reference (oracle/gen_golden_blocks.py), the reference's output for seeded
weights and inputs (tests/blockgen.py regenerates both bit for bit; large outputs are kept as a seeded random quarter of
their elements).  Each HIP block is run alone on the same data, under both GEMM arithmetics.

  swin_*     blocks/swin_transformer.py:94-156   one window / 2 x 2 windows, plain / cyclic-shifted with both masks,
                                                 31 x 31 relative table / dense 256 x 256 bias
  cross      models/cross_blocks.py:75-98        Interactive_crossAttn_type4 over two tiles
  convnext   blocks/conv_blocks.py:71-81         depthwise 5 x 5 across a tile seam + LN + MLP
  dcb4_*     blocks/dcvc.py:28-66                DepthConvBlock4 with / without the channel adaptor
  res_* / attn / upsample   taming/modules/diffusionmodules/model.py:38-53,117-137,168-192

Tolerance: max |ours - ref| <= 2e-5 * max |ref| (fp32 summation-order noise through one block measures ~1e-6)."""
import os

import numpy as np
import pytest
import torch

import blockgen

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def G(golden_dir):
    import sgic_amd  # noqa: F401
    g = np.load(os.path.join(golden_dir, "blocks.npz"))
    return g, blockgen.load_meta(g), {"ul": g["mask_ul"], "lr": g["mask_lr"]}


@pytest.fixture(params=["split3", "f32"])
def precision(request):
    from sgic_amd import ops
    old = ops.PRECISION
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(old)


def _check(g, name, what, ours, shape):
    ours = ours.detach().cpu().numpy().astype(np.float32)
    assert tuple(ours.shape) == tuple(shape), (name, what, ours.shape, shape)
    ref = g[f"{name}.{what}"]
    got = ours.reshape(-1)[blockgen.sample_index(name, what, ours.size)]
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    print(f"[block parity] {name}.{what}: rel err {err:.2e} over {ref.size} of {ours.size} elements")
    assert err <= TOL, (name, what, err)


def _to_tm16(x_nhwc):
    """(B, H, W, C) -> rows in 16 x 16-tile-major order [(b, ty, tx, y % 16, x % 16), C]"""
    B, H, W, C = x_nhwc.shape
    return x_nhwc.reshape(B, H // 16, 16, W // 16, 16, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C).contiguous()


def _from_tm16(rows, B, H, W):
    C = rows.shape[1]
    return rows.reshape(B, H // 16, W // 16, 16, 16, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


def _x(name, what, meta):
    return torch.from_numpy(blockgen.input_for(name, what, meta[name]["inputs"][what]))


@pytest.mark.parametrize("name", ["swin_1x1_plain", "swin_1x1_shift", "swin_2x2_plain", "swin_2x2_shift"])
def test_swin_block_vs_reference(G, precision, name):
    from sgic_amd.encoder import SwinW, swin_forward
    g, meta, masks = G
    m = meta[name]
    sd = blockgen.state_dict_for(name, m, masks)
    w = SwinW(sd, "w", m["shifted"], m["rel"], 16, "cuda:0")
    x = _x(name, "x", meta)
    B, H, W, C = x.shape
    Fm = _to_tm16(x).cuda()
    swin_forward(Fm, w, B, H, W, 16)
    _check(g, name, "out", _from_tm16(Fm.cpu(), B, H, W), m["outputs"]["out"])


def test_cross_block_vs_reference(G, precision):
    from sgic_amd.encoder import CrossW, cross_forward
    g, meta, masks = G
    name = "cross"
    sd = blockgen.state_dict_for(name, meta[name], masks)
    w = CrossW(sd, "w", 2, "cuda:0")
    feat, tok = _x(name, "feat", meta), _x(name, "tokens", meta)          # (1, C, 16, 32), (289, N, 512)
    Lt, N, Wd = tok.shape
    Fm = _to_tm16(feat.permute(0, 2, 3, 1).contiguous()).cuda()
    X = tok.permute(1, 0, 2).reshape(N * Lt, Wd).contiguous().cuda()
    cross_forward(Fm, X, w, N, Lt, 256)
    _check(g, name, "feat", _from_tm16(Fm.cpu(), 1, 16, 32).permute(0, 3, 1, 2), meta[name]["outputs"]["feat"])
    _check(g, name, "tokens", X.cpu().reshape(N, Lt, Wd).permute(1, 0, 2), meta[name]["outputs"]["tokens"])


def test_convnext_block_vs_reference(G, precision):
    from sgic_amd.encoder import ConvNextW, convnext_forward
    g, meta, masks = G
    name = "convnext"
    w = ConvNextW(blockgen.state_dict_for(name, meta[name]), "w", "cuda:0")
    x = _x(name, "x", meta)                                                 # (1, C, 16, 32)
    Fm = _to_tm16(x.permute(0, 2, 3, 1).contiguous()).cuda()
    convnext_forward(Fm, w, 1, 16, 32)
    _check(g, name, "out", _from_tm16(Fm.cpu(), 1, 16, 32).permute(0, 3, 1, 2), meta[name]["outputs"]["out"])


@pytest.mark.parametrize("name", ["dcb4_same", "dcb4_adapt"])
def test_dcb4_block_vs_reference(G, precision, name):
    from sgic_amd.bottleneck import Dcb4W, dcb4_forward
    g, meta, masks = G
    w = Dcb4W(blockgen.state_dict_for(name, meta[name]), "w", "cuda:0")
    x = _x(name, "x", meta)
    B, C, H, W = x.shape
    y = dcb4_forward(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda(), w, B, H, W)
    _check(g, name, "out", y.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2), meta[name]["outputs"]["out"])


@pytest.mark.parametrize("name", ["res_same", "res_short"])
def test_taming_resnet_block_vs_reference(G, precision, name):
    from sgic_amd.decoder import VqganDecoderHIP, _ResW
    g, meta, masks = G
    w = _ResW(blockgen.state_dict_for(name, meta[name]), "w", "cuda:0")
    x = _x(name, "x", meta)
    B, C, H, W = x.shape
    y = VqganDecoderHIP._res(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda(), w, B, H, W)
    _check(g, name, "out", y.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2), meta[name]["outputs"]["out"])


def test_taming_attn_block_vs_reference(G, precision):
    from sgic_amd.decoder import VqganDecoderHIP, _AttnW
    g, meta, masks = G
    name = "attn"
    w = _AttnW(blockgen.state_dict_for(name, meta[name]), "w", "cuda:0")
    x = _x(name, "x", meta)
    B, C, H, W = x.shape
    y = VqganDecoderHIP._attn(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda(), w, B, H, W)
    _check(g, name, "out", y.cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2), meta[name]["outputs"]["out"])


def test_taming_upsample_vs_reference(G, precision):
    """Upsample(with_conv): nearest x2 then 3x3 conv (model.py:49-53) = halo_copy(upsample) + conv3x3 here"""
    from sgic_amd import ops
    g, meta, masks = G
    name = "upsample"
    sd = blockgen.state_dict_for(name, meta[name])
    cw = sd["w.conv.weight"]
    wt = cw.permute(0, 2, 3, 1).reshape(cw.shape[0], -1).contiguous().cuda()
    bias = sd["w.conv.bias"].cuda()
    x = _x(name, "x", meta)
    B, C, H, W = x.shape
    h = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().cuda()
    hu = ops.halo_copy(h, B, H, W, C, upsample=True, tile16=False, to_conv=(C, False))
    y = ops.conv3x3(hu, wt, bias, B, 2 * H, 2 * W, C, C)
    _check(g, name, "out", y.cpu().reshape(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2), meta[name]["outputs"]["out"])
