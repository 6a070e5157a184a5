"""CPU: the torch restatement (oracle/torch_ref.py) against the golden outputs of the REAL reference modules
(tests/golden/nn_small_*.npz, oracle/gen_golden_nn.py), and host logic of the product (CDF table builder)."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as TR


@pytest.fixture(scope="module")
def small():
    import sgic_amd  # noqa
    from sgic_amd import weights as W
    from sgic_amd.config import SMALL
    spec = W.encoder_spec(SMALL) + W.codec_misc_spec(SMALL) + W.bottleneck_spec(SMALL)
    return SMALL, W.synth_weights(spec, seed=1234)


@pytest.mark.parametrize("case", ["a", "c"])
def test_torch_ref_vs_reference_golden(case, small, golden_dir):
    from sgic_amd.data import synth_images
    cfg, sd = small
    g = np.load(os.path.join(golden_dir, f"nn_small_{case}.npz"))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    x = synth_images(B, H, W, int(g["seed"]))
    with torch.no_grad():
        z, h, stack = TR.encoder_forward(x * 0.5 + 0.5, sd, cfg)
        assert float((z - torch.from_numpy(g["z"])).abs().max()) < 2e-4
        assert float((h - torch.from_numpy(g["h"])).abs().max()) < 2e-4
        assert np.array_equal(TR.vq_indices(torch.from_numpy(g["z"]), sd).numpy(), g["vq_idx"])
        y = torch.from_numpy(g["y"])
        for b in range(B):
            s, i, _, _ = TR.four_part_prior_write(y[b:b + 1], sd, cfg.force_zero_thres)
            assert np.array_equal(s[0].numpy(), g["sym"][b]) and np.array_equal(i[0].numpy(), g["idx"][b])


def test_golden_streams_decode_with_oracle(golden_dir):
    """the reference's h_bit_stream fixtures decode (C oracle) back to the reference's symbols"""
    from oracle import orc
    t = np.load(os.path.join(golden_dir, "cdf_table.npz"))
    tab = orc.Table(t["cdf"], t["cdf_length"], t["offset"])
    for case in "abc":
        g = np.load(os.path.join(golden_dir, f"nn_small_{case}.npz"))
        for b in range(int(g["B"])):
            sym, idx = g["sym"][b].reshape(-1), g["idx"][b].reshape(-1)
            assert orc.rans_encode(sym, idx, tab) == g[f"stream_{b}"].tobytes()
            d = orc.Decoder(g[f"stream_{b}"].tobytes(), tab)
            q = len(sym) // 4
            got = np.concatenate([d.decode(idx[k * q:(k + 1) * q]) for k in range(4)])
            assert np.array_equal(got, np.where(idx < 0, 0, sym))


def test_product_cdf_table_builder_matches_reference(golden_dir):
    import sgic_amd  # noqa
    from sgic_amd.bottleneck import gaussian_cdf_table
    cdf, ln, off = gaussian_cdf_table()
    t = np.load(os.path.join(golden_dir, "cdf_table.npz"))
    assert np.array_equal(cdf, t["cdf"]) and np.array_equal(ln, t["cdf_length"]) and np.array_equal(off, t["offset"])


def test_quant_step_oracle_matches_torch_restatement(small):
    """C oracle of the fused quantiser (what the HIP kernel is checked against) == torch restatement"""
    from oracle import orc
    cfg, sd = small
    torch.manual_seed(1)
    y = torch.randn(1, 64, 6, 10) * 3
    s_t, i_t, yhat_t, sms = TR.four_part_prior_write(y, sd, 0.12)
    qs = torch.clamp_min(TR.prior_params(1, 6, 10, sd)[:, :64], 0.5)
    yq = (y / qs)[0].numpy()
    hat = np.zeros((64, 6, 10), np.float32)
    for k in range(4):
        s, i = orc.quant_step(yq, sms[k][0][0].numpy(), sms[k][1][0].numpy(), k, 0.12, hat)
        assert np.array_equal(s, s_t[0, k].numpy()) and np.array_equal(i, i_t[0, k].numpy())
