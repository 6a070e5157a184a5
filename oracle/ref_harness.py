"""Build-container-only harness that makes the REFERENCE importable (read-only) so golden vectors can
be generated from it.  TEST INFRASTRUCTURE: never imported by the product package, never runs on the
GPU box (there is no /root/reference there).

What it does:
  * stubs third-party modules the image lacks (torchac, pytorch_lightning, pytorch_msssim, lpips,
    torchvision, omegaconf) with empty shells -- none of them contributes arithmetic to the hot path;
  * creates a scratch shadow package  <tmp>/entropy/  that symlinks the reference's
    entropy_models.py / compression_model.py next to the py3.10 builds of the reference C++ coder
    (oracle/_ref/*.so, built by oracle/Makefile from /root/reference/src/cpp);
  * puts /root/reference/src on sys.path.
Nothing is copied into the repository.
"""
import os
import sys
import tempfile
import types

REF = os.environ.get("SGIC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def setup():
    import torch
    from torch import nn

    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present; goldens can only be generated in the build container")
    refdir = os.path.join(HERE, "_ref")
    if not os.path.isdir(refdir):
        raise RuntimeError("run `make -C oracle` first")

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _stub("torchac")
    pl = _stub("pytorch_lightning", LightningModule=nn.Module)
    pl.utilities = _stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    _stub("pytorch_msssim", MS_SSIM=_Dummy)
    _stub("lpips", LPIPS=_Dummy)
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils")

    class _AD(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    def _wrap(o):
        if isinstance(o, dict):
            return _AD({k: _wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [_wrap(v) for v in o]
        return o

    class OmegaConf:
        @staticmethod
        def create(o):
            return _wrap(o)

        @staticmethod
        def load(p):
            import yaml
            return _wrap(yaml.safe_load(open(p)))

    _stub("omegaconf", OmegaConf=OmegaConf)

    shadow = tempfile.mkdtemp(prefix="sgic_ref_shadow_")
    ent = os.path.join(shadow, "entropy")
    os.makedirs(ent)
    open(os.path.join(ent, "__init__.py"), "w").close()
    for f in ("entropy_models.py", "compression_model.py"):
        os.symlink(os.path.join(REF, "src", "entropy", f), os.path.join(ent, f))
    for f in os.listdir(refdir):
        if f.endswith(".so"):
            os.symlink(os.path.join(refdir, f), os.path.join(ent, f))
    sys.path.insert(0, shadow)
    sys.path.insert(1, os.path.join(REF, "src"))
    torch.set_grad_enabled(False)
    return _wrap
