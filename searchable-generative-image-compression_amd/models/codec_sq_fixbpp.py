"""Drop-in for the reference plug-in class `models.codec_sq_fixbpp.Codec` (src/config/config_test.yaml:2,
src/models/codec_sq_fixbpp.py:442-508): constructed as `Codec(**cfg.model.params)` with the reference's kwargs.
Training-only arguments (losses, training_strategy, monitor, tune_titok, save_mem) are accepted and ignored; the
architecture kwargs are checked against what the HIP path implements; `ckpt_path` is loaded with
`torch.load(weights_only=True)` (state_dict keys as in the reference checkpoint) and, when it is None, the
deterministic synthetic weights are used (the upstream checkpoint is not available offline)."""
import torch

from .. import weights as W
from ..codec import Codec as _Codec
from ..config import CodecConfig


def _get(d, *path, default=None):
    for p in path:
        if d is None:
            return default
        d = d.get(p) if isinstance(d, dict) else getattr(d, p, None)
    return default if d is None else d


def load_checkpoint(path, ignore_keys=()):
    """init_from_ckpt (codec_sq_fixbpp.py:494-507) without the unpickler: `weights_only=True` never executes code from
    the file; a {"state_dict": ...} wrapper is unwrapped; keys starting with an ignore_keys entry are dropped; keys the
    inference path does not know (vqgan.encoder.*, img_loss.*) stay in the dict and are simply never read
    (the reference loads with strict=False)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}


class Codec(_Codec):
    def __init__(self, embed_dim, feat_dim, in_pos_enc, in_pos_dec, n_attn, config, vqganconfig, imglossconfig=None,
                 featlossconfig=None, training_strategy=None, monitor="", ckpt_path=None, ignore_keys=(),
                 titok_pretrain_path=None, tune_titok=False, no_attn_vqgan=False, save_mem=False, device="cuda:0"):
        if no_attn_vqgan:
            raise NotImplementedError("no_attn_vqgan=True (VQGAN_wo_Attn) is not part of the shipped configuration")
        vq = _get(config, "model", "vq_model")
        dd = _get(vqganconfig, "ddconfig")
        cfg = CodecConfig(
            model_size=_get(vq, "vit_enc_model_size", default="large"), feat_dim=int(feat_dim), embed_dim=int(embed_dim),
            in_pos_enc=tuple(in_pos_enc), in_pos_dec=tuple(in_pos_dec), n_attn=int(n_attn),
            patch_size=int(_get(vq, "vit_enc_patch_size", default=16)),
            crop_size=int(_get(config, "dataset", "preprocessing", "crop_size", default=256)),
            num_latent_tokens=int(_get(vq, "num_latent_tokens", default=32)), token_size=int(_get(vq, "token_size", default=12)),
            codebook_size=int(_get(vq, "codebook_size", default=4096)),
            vq_ch=int(_get(dd, "ch", default=128)), vq_ch_mult=tuple(_get(dd, "ch_mult", default=(1, 1, 2, 2, 4))),
            vq_num_res_blocks=int(_get(dd, "num_res_blocks", default=2)),
            vq_attn_resolutions=tuple(_get(dd, "attn_resolutions", default=(16,))),
            vq_embed_dim=int(_get(vqganconfig, "embed_dim", default=256)), vq_n_embed=int(_get(vqganconfig, "n_embed", default=256)),
            vq_z_channels=int(_get(dd, "z_channels", default=256)))
        if _get(vq, "vit_dec_model_size", default=cfg.model_size) != cfg.model_size or not _get(vq, "use_l2_norm", default=True):
            raise NotImplementedError("encoder/decoder ViT sizes must match and use_l2_norm must be True")
        if ckpt_path is not None:
            sd = load_checkpoint(ckpt_path, ignore_keys)
        else:
            print("[Warning] ckpt_path is None: using deterministic synthetic weights")
            sd = W.synth_weights(W.full_spec(cfg), seed=1234)
        super().__init__(sd, cfg, device)
