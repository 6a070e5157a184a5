"""Per-block goldens from the REAL reference modules: every block class of the hot path is
instantiated ON ITS OWN from the imported reference (read-only, under the stubs of ref_harness.py), filled with the deterministic
weights of tests/blockgen.py, run on a seeded input, and only its OUTPUT (plus names / shapes) is committed to
tests/golden/blocks.npz.  tests/test_gpu_blocks.py rebuilds the same weights and inputs and compares the HIP blocks one by one.

Blocks (reference file:line):
  swin_1x1_plain / swin_1x1_shift / swin_2x2_plain / swin_2x2_shift   blocks/swin_transformer.py:64-156 (window 16; the relative
        31x31 table and the dense 256x256 bias; cyclic shift + upper_lower / left_right masks; one window and 2 x 2 windows)
  cross           models/cross_blocks.py:39-98  Interactive_crossAttn_type4 (two tiles, joint 545-token transformer, zero_add filled)
  convnext        blocks/conv_blocks.py:48-81   ConvNeXtBlock(k = 5, ratio 2) across a tile seam
  dcb4_same / dcb4_adapt   blocks/dcvc.py:13-66 DepthConvBlock4 with and without the channel adaptor
  res_same / res_short / attn / upsample   taming/modules/diffusionmodules/model.py:38-53,76-192 ResnetBlock (+ nin_shortcut),
        AttnBlock, Upsample(with_conv)

Run:  make -C oracle && python oracle/gen_golden_blocks.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness  # noqa: E402

ref_harness.setup()
import torch  # noqa: E402

import blockgen  # noqa: E402
from blocks.conv_blocks import ConvNeXtBlock  # noqa: E402  (reference code)
from blocks.dcvc import DepthConvBlock4  # noqa: E402
from blocks.swin_transformer import SwinBlock  # noqa: E402
from models.cross_blocks import Interactive_crossAttn_type4  # noqa: E402
from taming.modules.diffusionmodules.model import AttnBlock, ResnetBlock, Upsample  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)
out, meta = {}, {}
masks = {}


def fill(name, mod):
    sd = mod.state_dict()
    keys = []
    for k, t in sd.items():
        keys.append([k, list(t.shape)])
        if k.endswith("upper_lower_mask") or k.endswith("left_right_mask"):
            m = torch.isinf(t).numpy()
            assert set(np.unique(t.numpy()[~m])) <= {0.0}
            which = "ul" if k.endswith("upper_lower_mask") else "lr"
            assert which not in masks or np.array_equal(masks[which], m)
            masks[which] = m
        else:
            sd[k] = torch.from_numpy(blockgen.tensor_for(name, k, t.shape))
    mod.load_state_dict(sd)
    meta[name] = {"keys": keys}
    return mod.eval()


def keep(name, what, t):
    """the committed part of an output: all of a small one, a seeded random quarter of a large one (tests/blockgen.sample_index)"""
    flat = t.contiguous().numpy().reshape(-1)
    out[f"{name}.{what}"] = flat[blockgen.sample_index(name, what, flat.size)]
    meta[name].setdefault("outputs", {})[what] = list(t.shape)


def inp(name, what, shape):
    meta[name].setdefault("inputs", {})[what] = list(shape)
    return torch.from_numpy(blockgen.input_for(name, what, shape))


C = 256
for name, shifted, rel, (B, H, W) in [("swin_1x1_plain", False, True, (2, 16, 16)), ("swin_1x1_shift", True, False, (2, 16, 16)),
                                      ("swin_2x2_plain", False, False, (1, 32, 32)), ("swin_2x2_shift", True, True, (1, 32, 32))]:
    m = fill(name, SwinBlock(dim=C, heads=C // 64, head_dim=64, mlp_dim=4 * C, shifted=shifted, window_size=16, relative_pos_embedding=rel))
    x = inp(name, "x", (B, H, W, C))
    y = m(x)
    meta[name].update(shifted=shifted, rel=rel)
    keep(name, "out", y)
    print(name, tuple(y.shape), float(y.abs().max()))

name = "cross"
m = fill(name, Interactive_crossAttn_type4(titok_width=512, feat_width=C, num_attns=2, feat_patch_size=16, titok_patch_size=16,
                                           extra_titok_tokens=33))
feat = inp(name, "feat", (1, C, 16, 32))                 # two tiles side by side
tok = inp(name, "tokens", (289, 2, 512))
f2, t2 = m(feat, tok, (1, 2))
keep(name, "feat", f2)
keep(name, "tokens", t2)
print(name, tuple(f2.shape), tuple(t2.shape), float(f2.abs().max()), float(t2.abs().max()))

name = "convnext"
m = fill(name, ConvNeXtBlock(C, mlp_ratio=2, kernel_size=5))
x = inp(name, "x", (1, C, 16, 32))
y = m(x)
keep(name, "out", y)
print(name, tuple(y.shape), float(y.abs().max()))

for name, cin, cout in [("dcb4_same", C, C), ("dcb4_adapt", C, 64)]:
    m = fill(name, DepthConvBlock4(cin, cout))
    x = inp(name, "x", (2, cin, 8, 8))
    y = m(x)
    keep(name, "out", y)
    print(name, tuple(y.shape), float(y.abs().max()))

for name, cin, cout in [("res_same", 128, 128), ("res_short", 256, 128)]:
    m = fill(name, ResnetBlock(in_channels=cin, out_channels=cout, temb_channels=0, dropout=0.0))
    x = inp(name, "x", (2, cin, 16, 16))
    y = m(x, None)
    keep(name, "out", y)
    print(name, tuple(y.shape), float(y.abs().max()))

name = "attn"
m = fill(name, AttnBlock(128))
x = inp(name, "x", (2, 128, 16, 16))
y = m(x)
keep(name, "out", y)
print(name, tuple(y.shape), float(y.abs().max()))

name = "upsample"
m = fill(name, Upsample(128, with_conv=True))
x = inp(name, "x", (2, 128, 8, 8))
y = m(x)
keep(name, "out", y)
print(name, tuple(y.shape), float(y.abs().max()))

out["mask_ul"], out["mask_lr"] = masks["ul"], masks["lr"]
out["meta"] = np.array(json.dumps(meta))
dst = os.path.join(ROOT, "tests", "golden", "blocks.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
