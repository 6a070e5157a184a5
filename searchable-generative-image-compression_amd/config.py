"""Static configuration of the codec (mirrors src/config/config_test.yaml:3-54 of the reference)."""
from dataclasses import dataclass, field
from typing import Tuple

_SIZES = {"small": (512, 8, 8), "base": (768, 12, 12), "large": (1024, 24, 16)}  # titok/blocks.py:82-96


@dataclass(frozen=True)
class CodecConfig:
    model_size: str = "large"                      # vit_enc_model_size == vit_dec_model_size
    feat_dim: int = 768                            # detail-branch width (params.feat_dim)
    embed_dim: int = 64                            # bottleneck latent channels (params.embed_dim)
    in_pos_enc: Tuple[int, ...] = (3, 7, 11, 15, 19)
    in_pos_dec: Tuple[int, ...] = (3, 7, 11, 15, 19)
    n_attn: int = 2
    patch_size: int = 16
    crop_size: int = 256
    num_latent_tokens: int = 32
    token_size: int = 12
    codebook_size: int = 4096
    window: int = 16
    force_zero_thres: float = 0.12                 # compress.py:238
    # generative decoder: FeatMerge + taming VQGAN (config_test.yaml:37-54, codec_sq_fixbpp.py:395-439)
    fm_inner: int = 1024
    vq_ch: int = 128
    vq_ch_mult: Tuple[int, ...] = (1, 1, 2, 2, 4)
    vq_num_res_blocks: int = 2
    vq_attn_resolutions: Tuple[int, ...] = (16,)
    vq_embed_dim: int = 256
    vq_n_embed: int = 256
    vq_z_channels: int = 256

    @property
    def width(self):
        return _SIZES[self.model_size][0]

    @property
    def layers(self):
        return _SIZES[self.model_size][1]

    @property
    def heads(self):
        return _SIZES[self.model_size][2]

    @property
    def grid(self):
        return self.crop_size // self.patch_size

    def titok_dict(self):
        """the `config` kwarg the reference's Codec/Hybrid_Codec expects (config_test.yaml:19-35)"""
        return {"model": {"vq_model": {"codebook_size": self.codebook_size, "token_size": self.token_size,
                                       "use_l2_norm": True, "commitment_cost": 0.25,
                                       "vit_enc_model_size": self.model_size, "vit_dec_model_size": self.model_size,
                                       "vit_enc_patch_size": self.patch_size, "vit_dec_patch_size": self.patch_size,
                                       "num_latent_tokens": self.num_latent_tokens}},
                "dataset": {"preprocessing": {"crop_size": self.crop_size}}}

    def vqgan_ddconfig(self):
        """`vqganconfig.ddconfig` of the reference (config_test.yaml:44-54)"""
        return dict(double_z=False, z_channels=self.vq_z_channels, resolution=self.crop_size, in_channels=3, out_ch=3,
                    ch=self.vq_ch, ch_mult=list(self.vq_ch_mult), num_res_blocks=self.vq_num_res_blocks,
                    attn_resolutions=list(self.vq_attn_resolutions), dropout=0.0)


LARGE = CodecConfig()
# small configuration used by fast parity tests / golden fixtures (same topology, 8 layers, width 512,
# detail width 256, cross stages after layers 1 and 5)
SMALL = CodecConfig(model_size="small", feat_dim=256, in_pos_enc=(1, 5), in_pos_dec=(1, 5),
                    vq_ch=32, vq_embed_dim=64, vq_n_embed=64, vq_z_channels=64)


@dataclass(frozen=True)
class ClipConfig:
    """OpenCLIP ViT-B-32: image tower (compress.py:59-63) and text tower (search.py:54-55,93-97)."""
    image_size: int = 224
    patch: int = 32
    width: int = 768
    layers: int = 12
    heads: int = 12
    embed_dim: int = 512
    vocab: int = 49408          # text tower: BPE vocabulary, context length, transformer width/heads/layers
    ctx: int = 77
    t_width: int = 512
    t_heads: int = 8
    t_layers: int = 12
    mean: Tuple[float, float, float] = (0.48145466, 0.4578275, 0.40821073)
    std: Tuple[float, float, float] = (0.26862954, 0.26130258, 0.27577711)


CLIP_B32 = ClipConfig()
CLIP_TINY = ClipConfig(width=128, layers=2, heads=2, embed_dim=64, vocab=512, ctx=16, t_width=128, t_heads=2, t_layers=2)
