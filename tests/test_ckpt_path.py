"""`ckpt_path` round trip (codec_sq_fixbpp.py:494-507 init_from_ckpt): a checkpoint file with the reference's key layout --
a {"state_dict": ...} wrapper, the three scalars compress.py asks to ignore (compress.py:219), training-only tensors
the inference path never reads -- loads through the plug-in class and gives the same bitstreams as the in-memory dict."""
import os

import pytest
import torch


def _small_kwargs():
    import sgic_amd  # noqa
    from sgic_amd.config import SMALL
    return dict(embed_dim=SMALL.embed_dim, feat_dim=SMALL.feat_dim, in_pos_enc=list(SMALL.in_pos_enc), in_pos_dec=list(SMALL.in_pos_dec),
                n_attn=SMALL.n_attn, config=SMALL.titok_dict(),
                vqganconfig={"embed_dim": SMALL.vq_embed_dim, "n_embed": SMALL.vq_n_embed, "ddconfig": SMALL.vqgan_ddconfig()},
                imglossconfig=None, featlossconfig=None, training_strategy=None, monitor="val/loss",
                ignore_keys=["epoch_for_strategy", "lmbda_idx", "lmbda_list"])


def _write_ckpt(path):
    import sgic_amd  # noqa
    from sgic_amd import weights as W
    from sgic_amd.config import SMALL
    sd = W.synth_weights(W.full_spec(SMALL), seed=77)
    extra = {"epoch_for_strategy": torch.tensor(3), "lmbda_idx": torch.tensor(1), "lmbda_list": torch.tensor([0.1, 0.2]),
             "vqgan.encoder.conv_in.weight": torch.zeros(4, 3, 3, 3), "img_loss.logvar": torch.zeros(())}
    torch.save({"state_dict": {**sd, **extra}, "epoch": 12, "global_step": 3400}, path)
    return sd


def test_load_checkpoint_filters_ignored_keys(tmp_path):
    """CPU: the loader half of the plug-in (no GPU needed)"""
    import sgic_amd  # noqa
    from sgic_amd.models.codec_sq_fixbpp import load_checkpoint
    p = str(tmp_path / "model.ckpt")
    sd = _write_ckpt(p)
    got = load_checkpoint(p, ignore_keys=["epoch_for_strategy", "lmbda_idx", "lmbda_list"])
    assert not any(k.startswith(("epoch_for_strategy", "lmbda_")) for k in got)
    assert set(sd) <= set(got) and "vqgan.encoder.conv_in.weight" in got            # unexpected keys are tolerated (strict=False)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    torch.save(sd, p)                                                                 # a bare state_dict loads too
    assert set(load_checkpoint(p)) == set(sd)
    with pytest.raises(Exception):                                                    # a pickle with code in it is refused, not executed
        torch.save({"state_dict": {"x": os.getcwd}}, p)
        load_checkpoint(p)


@pytest.mark.gpu
def test_ckpt_path_through_the_plugin_class(tmp_path):
    import sgic_amd  # noqa
    from sgic_amd.codec import Codec as Engine
    from sgic_amd.config import SMALL
    from sgic_amd.data import synth_images
    from sgic_amd.models.codec_sq_fixbpp import Codec
    p = str(tmp_path / "model.ckpt")
    sd = _write_ckpt(p)
    m = Codec(ckpt_path=p, **_small_kwargs()).to("cuda:0").eval()
    m.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    m.hybrid_codec.quantize_feat.update(force=True)
    ref = Engine(sd, SMALL, "cuda:0")
    ref.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    ref.hybrid_codec.quantize_feat.update(force=True)
    x = synth_images(1, 256, 512, 5).cuda()
    a, b = m.encode_only(x), ref.encode_only(x)
    assert a["h_bit_stream"] == b["h_bit_stream"] and a["z_bit_stream"] == b["z_bit_stream"]
    assert torch.equal(m.decode_only(**a), ref.decode_only(**b))
    with pytest.raises(KeyError):                                   # a checkpoint that misses weights the path needs fails loudly
        torch.save({k: v for k, v in sd.items() if "hybrid_codec.encoder.transformer.0." not in k}, p)
        Codec(ckpt_path=p, **_small_kwargs())
