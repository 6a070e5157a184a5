"""MI355X-native compress/decompress hot path of the searchable generative image codec.

Host side is Python (like the reference); all arithmetic on the hot path runs in hand-written HIP
kernels for gfx950 behind the C ABI declared in include/sgic.h (csrc/ -> libsgic.so).  PyTorch is used
only for device memory, streams and torch.distributed.  There is NO CPU fallback: importing
`sgic_amd._lib` fails loudly when libsgic.so is missing, and every op raises without a GPU.
"""
__version__ = "0.1.0"
