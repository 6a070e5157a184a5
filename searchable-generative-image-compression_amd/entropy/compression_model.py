"""Mirror of the names compress.py imports from the reference's entropy/compression_model.py."""


def get_padding_size(height, width, p=64):
    """entropy/compression_model.py:13-22 -- right/bottom padding to a multiple of p"""
    new_h = (height + p - 1) // p * p
    new_w = (width + p - 1) // p * p
    padding_left = 0
    padding_right = new_w - width - padding_left
    padding_top = 0
    padding_bottom = new_h - height - padding_top
    return padding_left, padding_right, padding_top, padding_bottom
