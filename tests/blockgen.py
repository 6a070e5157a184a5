"""Deterministic weights / inputs for the per-block goldens (tests/golden/blocks.npz).  numpy only: the SAME function fills the
reference module's state_dict in oracle/gen_golden_blocks.py (build container) and the HIP block's weights in
tests/test_gpu_blocks.py (GPU box), so the fixture carries only names, shapes and the reference's OUTPUTS.
PCG64 + standard_normal(float64) -> float32 is bit-reproducible across machines."""
import json
import zlib

import numpy as np


def tensor_for(block, key, shape, seed=0):
    """value of state_dict entry `key` of golden block `block`.  Scales keep activations O(1) through a block:
    matrices ~ N(0, 1/fan_in), norm gains ~ 1 +- 0.1, biases / positional tables small, layer_scale around 0.5."""
    rng = np.random.default_rng([zlib.crc32(f"{block}/{key}".encode()), seed])
    shape = tuple(int(s) for s in shape)
    x = rng.standard_normal(shape)
    leaf = key.rsplit(".", 1)[-1]
    if "layer_scale" in key:
        x = 0.5 + 0.1 * x
    elif "pos_emb" in key:                      # pos_embedding (Swin), titok_pos_emb / feat_pos_emb (cross block)
        x = 0.2 * x
    elif leaf == "bias":
        x = 0.05 * x
    elif len(shape) == 1:                       # LayerNorm / GroupNorm gains
        x = 1.0 + 0.1 * x
    else:                                       # Linear (out, in) / Conv (out, in/groups, kh, kw)
        fan_in = int(np.prod(shape[1:]))
        x = x / np.sqrt(fan_in)
    return x.astype(np.float32)


def input_for(block, name, shape, seed=0):
    rng = np.random.default_rng([zlib.crc32(f"{block}/input/{name}".encode()), seed])
    return rng.standard_normal(tuple(shape)).astype(np.float32)


def state_dict_for(block, meta, masks=None, prefix="w"):
    """{f"{prefix}.{key}": torch tensor} for every entry the fixture lists for `block`; the Swin shift masks (constants of the
    window size, -inf patterns) come from the fixture's boolean images"""
    import torch
    sd = {}
    for key, shape in meta["keys"]:
        if key.endswith("upper_lower_mask") or key.endswith("left_right_mask"):
            m = masks["ul" if key.endswith("upper_lower_mask") else "lr"]
            t = torch.zeros(m.shape, dtype=torch.float32)
            t[torch.from_numpy(m)] = float("-inf")
        else:
            t = torch.from_numpy(tensor_for(block, key, shape))
        sd[f"{prefix}.{key}"] = t
    return sd


def load_meta(g):
    return json.loads(str(g["meta"]))


def sample_index(block, what, n, keep=0.25, small=40000):
    """flat indices of the elements of an n-element output the fixture keeps: everything for small outputs, else a seeded random
    quarter (every channel, head, window and seam position is hit many times; the file stays small)"""
    if n <= small:
        return np.arange(n)
    rng = np.random.default_rng([zlib.crc32(f"{block}/sample/{what}".encode()), 1])
    return np.nonzero(rng.random(n) < keep)[0]
