"""Host mirror of the reference's entropy/entropy_models.py:32-94,252-374 -- `EntropyCoder` and `GaussianEncoder`
with the same methods, on top of the HIP coder (MLCodec_rans facade) and the HIP index kernel.  The codec's batched
path (sgic_amd.bottleneck) does not go through these per-tensor wrappers; they exist so that code written against
the reference's L3 API (tensor in, bytes out) keeps working."""
import math

import numpy as np
import torch


class EntropyCoder:
    def __init__(self, ec_thread=False, stream_part=1):
        from .MLCodec_rans import RansDecoder, RansEncoder
        self.encoder = RansEncoder(ec_thread, stream_part)
        self.decoder = RansDecoder(stream_part)

    @staticmethod
    def pmf_to_quantized_cdf(pmf, precision=16):
        from .MLCodec_CXX import pmf_to_quantized_cdf as _f
        return torch.IntTensor(_f(pmf.tolist(), precision))

    def reset(self):
        self.encoder.reset()

    def add_cdf(self, cdf, cdf_length, offset):
        e = self.encoder.add_cdf(cdf, cdf_length, offset)
        d = self.decoder.add_cdf(cdf, cdf_length, offset)
        assert e == d
        return e

    def encode_with_indexes(self, symbols, indexes, cdf_group_index):
        self.encoder.encode_with_indexes(symbols.clamp(-30000, 30000).to(torch.int16).cpu().numpy(),
                                         indexes.to(torch.int16).cpu().numpy(), cdf_group_index)

    def flush(self):
        self.encoder.flush()

    def get_encoded_stream(self):
        return self.encoder.get_encoded_stream().tobytes()

    def set_stream(self, stream):
        self.decoder.set_stream(np.frombuffer(stream, dtype=np.uint8))

    def decode_stream(self, indexes, cdf_group_index):
        return torch.Tensor(self.decoder.decode_stream(indexes.to(torch.int16).cpu().numpy(), cdf_group_index))


class GaussianEncoder:
    """distribution='gaussian' only (what the codec uses, models/sq_bottleneck.py:57)"""

    def __init__(self, distribution="gaussian"):
        assert distribution == "gaussian"
        self.scale_min, self.scale_max, self.scale_level = 0.11, 64.0, 256
        self.scale_table = torch.exp(torch.linspace(math.log(self.scale_min), math.log(self.scale_max), self.scale_level))
        self.log_scale_min = math.log(self.scale_min)
        self.log_scale_step = (math.log(self.scale_max) - self.log_scale_min) / (self.scale_level - 1)
        self.entropy_coder = None
        self.cdf_group_index = None
        self._cdf = None

    def update(self, force=False, entropy_coder=None):
        assert entropy_coder is not None
        self.entropy_coder = entropy_coder
        if not force and self._cdf is not None:
            return
        from ..bottleneck import gaussian_cdf_table
        self._cdf = gaussian_cdf_table()
        self.cdf_group_index = entropy_coder.add_cdf(*self._cdf)

    def get_cdf_info(self):
        return self._cdf

    def build_indexes(self, scales, skip_thres=None):
        """entropy_models.py:355-362, evaluated by the HIP index kernel (same fp32 formula)"""
        from .. import ops
        s = scales.detach().to(torch.float32).contiguous()
        if not s.is_cuda:
            s = s.cuda()
        idx = torch.empty(s.numel(), dtype=torch.int16, device=s.device)
        ops.scale_indexes(s.view(-1), idx, skip_thres)
        return idx.view(scales.shape).int()

    def encode(self, x, scales, skip_thres=None):
        idx = self.build_indexes(scales, skip_thres)
        return self.entropy_coder.encode_with_indexes(x.reshape(-1), idx.reshape(-1), self.cdf_group_index)

    def decode_stream(self, scales, dtype, device, skip_thres=None):
        idx = self.build_indexes(scales, skip_thres)
        val = self.entropy_coder.decode_stream(idx.reshape(-1), self.cdf_group_index)
        return val.reshape(scales.shape).to(device).to(dtype)
