"""GPU unit tests of the secondary kernels against plain PyTorch (fp64) references of the same op -- LayerNorm
(titok/blocks.py:36,42), GroupNorm + swish into a zero-halo buffer (taming model.py:38-48), depthwise convolutions in
both feature-map layouts (blocks/conv_blocks.py:63-67, blocks/dcvc.py:21,35), row softmax, exact top-k.  The model-level
tests exercise them only at the shapes the codec uses; these cover ragged shapes and every layout flag."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tm16(x_nhwc):
    """(B,H,W,C) -> rows in 16x16-tile-major order"""
    B, H, W, C = x_nhwc.shape
    return x_nhwc.reshape(B, H // 16, 16, W // 16, 16, C).permute(0, 1, 3, 2, 4, 5).reshape(B * H * W, C)


def _from_tm16(rows, B, H, W):
    C = rows.shape[1]
    return rows.reshape(B, H // 16, W // 16, 16, 16, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


@pytest.mark.parametrize("M,C,act", [(9248, 1024, 0), (1000, 768, 0), (77, 512, 0), (2048, 128, 2), (5, 64, 0)])
def test_layernorm_vs_torch(M, C, act):
    import sgic_amd  # noqa
    from sgic_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 3 + 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (C,), w.double(), b.double(), eps=1e-5)
    if act == 2:
        ref = torch.nn.functional.silu(ref)
    got = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), act=act).cpu().double()
    assert float((got - ref).abs().max()) < 2e-5
    # segment maps: normalise the first 3 rows of every 5-row segment into the first 3 rows of every 4-row segment
    if M >= 20:
        n = M // 5
        out = torch.zeros(n * 4, C, device=DEV)
        ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), out=out, M=n * 3, x_seg=(3, 5), y_seg=(3, 4))
        o = out.cpu().double().view(n, 4, C)
        assert float((o[:, :3] - ref.view(-1, C)[:n * 5].view(n, 5, C)[:, :3]).abs().max()) < 2e-5 if act == 0 else True
        assert float(o[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,C,swish,halo", [(2, 16, 16, 128, True, True), (3, 8, 24, 256, False, False), (1, 32, 32, 64, True, False)])
def test_groupnorm_vs_torch(B, H, W, C, swish, halo):
    import sgic_amd  # noqa
    from sgic_amd import ops
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = torch.nn.functional.group_norm(x.double(), 32, w.double(), b.double(), eps=1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
    out = ops.groupnorm(rows, w.to(DEV), b.to(DEV), B, H, W, swish=swish, halo=halo).cpu().double()
    if halo:
        o = out.view(B, H + 2, W + 2, C)
        assert float(o[:, 0].abs().max()) == 0.0 and float(o[:, :, 0].abs().max()) == 0.0 and float(o[:, -1].abs().max()) == 0.0
        out = o[:, 1:-1, 1:-1]
    got = out.reshape(B, H, W, C).permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) < 3e-5


@pytest.mark.parametrize("B,H,W,C,k,tile16,pre", [(2, 16, 16, 768, 5, True, True), (3, 8, 8, 64, 3, False, False), (1, 32, 16, 128, 5, True, False),
                                                   (2, 6, 10, 32, 3, False, True), (1, 16, 16, 64, 7, False, False)])
def test_dwconv_vs_torch(B, H, W, C, k, tile16, pre):
    """both kernels (4-pixels-per-thread for k in {3,5} and W % 4 == 0, generic otherwise), both layouts"""
    import sgic_amd  # noqa
    from sgic_amd import ops
    g = torch.Generator().manual_seed(H * W + k)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g)
    b = torch.randn(C, generator=g)
    ps = torch.rand(C, generator=g) + 0.5 if pre else None
    xin = x * ps.view(1, C, 1, 1) if pre else x
    ref = torch.nn.functional.conv2d(xin.double(), w.double(), b.double(), padding=k // 2, groups=C).permute(0, 2, 3, 1)
    nhwc = x.permute(0, 2, 3, 1).contiguous()
    rows = (_tm16(nhwc) if tile16 else nhwc.reshape(B * H * W, C)).contiguous().to(DEV)
    wk = w.reshape(C, k * k).t().contiguous().to(DEV)
    out = ops.dwconv(rows, wk, b.to(DEV), ps.to(DEV) if pre else None, B, H, W, k, tile16=tile16).cpu()
    got = (_from_tm16(out, B, H, W) if tile16 else out.reshape(B, H, W, C)).double()
    assert float((got - ref).abs().max()) < 2e-5


def test_softmax_rows_and_topk_vs_torch():
    import sgic_amd  # noqa
    from sgic_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(300, 256, generator=g) * 4
    got = ops.softmax_rows(x.to(DEV).contiguous(), 256, scale=0.5).cpu().double()
    assert float((got - torch.softmax(x.double() * 0.5, dim=-1)).abs().max()) < 1e-6
    s = torch.randn(9, 5000, generator=g)
    s[3, 100] = s[3, 7] = 9.0                                  # a tie: the lower index must come first
    val, idx = ops.topk_rows(s.to(DEV).clone(), 20)
    ref = torch.sort(s, dim=1, descending=True, stable=True)
    assert torch.equal(idx.cpu().long(), ref.indices[:, :20]) and torch.equal(val.cpu(), ref.values[:, :20])
    assert idx[3, 0].item() == 7 and idx[3, 1].item() == 100


def test_pad_replicate_vs_torch():
    """compress.py:258-261 (get_padding_size + F.pad replicate): bit-exact data movement"""
    import sgic_amd  # noqa
    from sgic_amd import ops
    g = torch.Generator().manual_seed(1)
    for (B, H, W, pad) in [(2, 200, 300, (0, 212, 0, 56)), (1, 256, 256, (0, 0, 0, 0)), (3, 17, 9, (2, 3, 1, 4))]:
        x = torch.randn(B, 3, H, W, generator=g)
        ref = torch.nn.functional.pad(x, pad, mode="replicate")
        got = ops.pad_replicate(x.to(DEV), *pad).cpu()
        assert got.shape == ref.shape and torch.equal(got, ref)
