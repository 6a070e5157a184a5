"""Drop-in for the two `torchac` functions the reference calls (models/codec_sq_fixbpp.py:864,887), so that
`import sgic_amd.torchac as torchac` is the whole integration of seam B2.

The reference only ever passes the UNIFORM cdf i/4096 over the 4096 VQ codes (`set_torchac`, :841-846).  torchac
(third-party, 0.9.3, not vendored) normalises that to the 16-bit cdf 16*i, so every symbol costs exactly 12 bits and
its range coder's output is the big-endian 12-bit packing of the indices followed by the terminator byte 0x40 --
verified on the reference's own `IO/bitstreams/apple.c2df` (oracle/sgic_oracle.c restates the coder).  That is what
the device kernels `sgic_pack12_batch` / `sgic_unpack12_batch` compute; any other cdf is rejected loudly."""
import numpy as np
import torch

from . import ops

_LP = 4097


def _check_uniform(cdf_float):
    c = torch.as_tensor(cdf_float)
    if c.shape[-1] != _LP:
        raise NotImplementedError(f"only the uniform {_LP - 1}-symbol cdf of the codec is supported (got Lp = {c.shape[-1]})")
    row = c.reshape(-1, _LP)[0].double().cpu()
    if float((row - torch.arange(_LP, dtype=torch.float64) / (_LP - 1)).abs().max()) > 1e-7:
        raise NotImplementedError("only the uniform cdf i/4096 (Codec.set_torchac) is supported")


def encode_float_cdf(cdf_float, sym, needs_normalization=True, check_input_bounds=False):
    """cdf_float (N, 4097) uniform, sym int16 (N,) -> bytes"""
    _check_uniform(cdf_float)
    s = torch.as_tensor(sym).reshape(-1)
    if check_input_bounds and (int(s.min()) < 0 or int(s.max()) >= _LP - 1):
        raise ValueError("symbol out of range")
    ops.require_gpu()
    d = s.to(device="cuda", dtype=torch.int32).contiguous()
    return ops.pack12_batch(d, 1, d.numel())[0].cpu().numpy().tobytes()


def decode_float_cdf(cdf_float, byte_stream, needs_normalization=True):
    """-> int16 tensor (N,) on the CPU; N is the number of rows of cdf_float, as in torchac"""
    _check_uniform(cdf_float)
    n = int(torch.as_tensor(cdf_float).reshape(-1, _LP).shape[0])
    raw = np.frombuffer(bytes(byte_stream), dtype=np.uint8)
    need = ops.lib.sgic_pack12_size(n)
    if raw.size != need:
        raise ValueError(f"stream of {raw.size} bytes does not hold {n} 12-bit symbols ({need} bytes)")
    ops.require_gpu()
    d = torch.from_numpy(raw.copy()).cuda().reshape(1, -1)
    return ops.unpack12_batch(d, 1, n)[0].cpu().to(torch.int16)
