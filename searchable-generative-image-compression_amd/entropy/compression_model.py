"""Mirror of the names compress.py imports from the reference's entropy/compression_model.py."""


def get_padding_size(height, width, p=64):
    """(left, right, top, bottom) padding that brings (height, width) up to multiples of p -- all of it on the right /
    bottom edge, as the reference pads (entropy/compression_model.py:13-22; consumed by F.pad(..., mode="replicate"),
    compress.py:258-261)"""
    return 0, -width % p, 0, -height % p
