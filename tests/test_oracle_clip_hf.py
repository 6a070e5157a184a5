"""CPU: pin the oracle's CLIP restatement (oracle/torch_ref.py clip_tower / clip_text_tower) against an
independent third-party implementation of the same published architecture: HuggingFace `transformers`
CLIPVisionModelWithProjection / CLIPTextModelWithProjection, built from a config (no download) and loaded with
the same synthetic weights.  open_clip itself (compress.py:61-63, search.py:54-55) is not installed and its
laion2b weights are a network fetch, so this is the strongest pin available offline: architecture parity on
synthetic weights (numeric parity with the real checkpoint stays unpinned)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as TR

transformers = pytest.importorskip("transformers")


def _map_layers(sd, p, n, width, hf, hp):
    for i in range(n):
        s, d = f"{p}.transformer.resblocks.{i}", f"{hp}.encoder.layers.{i}"
        q, k, v = sd[f"{s}.attn.in_proj_weight"].split(width, 0)
        qb, kb, vb = sd[f"{s}.attn.in_proj_bias"].split(width, 0)
        hf.update({f"{d}.self_attn.q_proj.weight": q, f"{d}.self_attn.k_proj.weight": k, f"{d}.self_attn.v_proj.weight": v,
                   f"{d}.self_attn.q_proj.bias": qb, f"{d}.self_attn.k_proj.bias": kb, f"{d}.self_attn.v_proj.bias": vb,
                   f"{d}.self_attn.out_proj.weight": sd[f"{s}.attn.out_proj.weight"],
                   f"{d}.self_attn.out_proj.bias": sd[f"{s}.attn.out_proj.bias"],
                   f"{d}.layer_norm1.weight": sd[f"{s}.ln_1.weight"], f"{d}.layer_norm1.bias": sd[f"{s}.ln_1.bias"],
                   f"{d}.layer_norm2.weight": sd[f"{s}.ln_2.weight"], f"{d}.layer_norm2.bias": sd[f"{s}.ln_2.bias"],
                   f"{d}.mlp.fc1.weight": sd[f"{s}.mlp.c_fc.weight"], f"{d}.mlp.fc1.bias": sd[f"{s}.mlp.c_fc.bias"],
                   f"{d}.mlp.fc2.weight": sd[f"{s}.mlp.c_proj.weight"], f"{d}.mlp.fc2.bias": sd[f"{s}.mlp.c_proj.bias"]})


@pytest.fixture(scope="module")
def tiny():
    import sgic_amd  # noqa
    from sgic_amd import weights as W
    from sgic_amd.config import CLIP_TINY
    return CLIP_TINY, W.synth_weights(W.clip_spec(CLIP_TINY) + W.clip_text_spec(CLIP_TINY), seed=4321)


def test_image_tower_vs_hf(tiny):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    cfg, sd = tiny
    hc = CLIPVisionConfig(hidden_size=cfg.width, intermediate_size=4 * cfg.width, projection_dim=cfg.embed_dim,
                          num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, image_size=cfg.image_size,
                          patch_size=cfg.patch, hidden_act="gelu", layer_norm_eps=1e-5, attn_implementation="eager")
    m = CLIPVisionModelWithProjection(hc).eval()
    p = "clip.visual"
    hf = {"vision_model.embeddings.class_embedding": sd[f"{p}.class_embedding"],
          "vision_model.embeddings.patch_embedding.weight": sd[f"{p}.conv1.weight"],
          "vision_model.embeddings.position_embedding.weight": sd[f"{p}.positional_embedding"],
          "vision_model.pre_layrnorm.weight": sd[f"{p}.ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd[f"{p}.ln_pre.bias"],
          "vision_model.post_layernorm.weight": sd[f"{p}.ln_post.weight"],
          "vision_model.post_layernorm.bias": sd[f"{p}.ln_post.bias"],
          "visual_projection.weight": sd[f"{p}.proj"].t().contiguous()}
    _map_layers(sd, p, cfg.layers, cfg.width, hf, "vision_model")
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((3, 3, cfg.image_size, cfg.image_size), dtype=np.float32))
    with torch.no_grad():
        ref = m(pixel_values=x).image_embeds
        ref = ref / ref.norm(dim=-1, keepdim=True)
        got = TR.clip_tower(x, sd, cfg)
    assert float((got - ref).abs().max()) < 2e-5


def test_text_tower_vs_hf(tiny):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    cfg, sd = tiny
    eot = cfg.vocab - 1
    hc = CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.t_width, intermediate_size=4 * cfg.t_width,
                        projection_dim=cfg.embed_dim, num_hidden_layers=cfg.t_layers, num_attention_heads=cfg.t_heads,
                        max_position_embeddings=cfg.ctx, hidden_act="gelu", layer_norm_eps=1e-5, bos_token_id=cfg.vocab - 2,
                        eos_token_id=eot, pad_token_id=0, attn_implementation="eager")
    m = CLIPTextModelWithProjection(hc).eval()
    p = "clip"
    hf = {"text_model.embeddings.token_embedding.weight": sd[f"{p}.token_embedding.weight"],
          "text_model.embeddings.position_embedding.weight": sd[f"{p}.positional_embedding"],
          "text_model.final_layer_norm.weight": sd[f"{p}.ln_final.weight"],
          "text_model.final_layer_norm.bias": sd[f"{p}.ln_final.bias"],
          "text_projection.weight": sd[f"{p}.text_projection"].t().contiguous()}
    _map_layers(sd, p, cfg.t_layers, cfg.t_width, hf, "text_model")
    missing, unexpected = m.load_state_dict(hf, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    # open_clip token layout: <start> words... <end> then zero padding; <end> is the largest id (argmax pooling)
    rng = np.random.default_rng(1)
    toks = np.zeros((4, cfg.ctx), dtype=np.int64)
    for b, n in enumerate((1, 5, cfg.ctx - 2, 9)):
        toks[b, 0] = cfg.vocab - 2
        toks[b, 1:1 + n] = rng.integers(1, cfg.vocab - 2, n)
        toks[b, 1 + n] = eot
    t = torch.from_numpy(toks)
    with torch.no_grad():
        ref = m(input_ids=t).text_embeds
        ref = ref / ref.norm(dim=-1, keepdim=True)
        got = TR.clip_text_tower(t, sd, cfg)
    assert float((got - ref).abs().max()) < 2e-5
