"""Mirror of modules/vocoder_blocks/gan_utils.py:12-28 (the two helpers on the generator path)."""


def get_padding(kernel_size, dilation=1):
    """gan_utils.py:12-13"""
    return int((kernel_size * dilation - dilation) / 2)


def init_weights(m, mean=0.0, std=0.01):
    """gan_utils.py:25-28 -- normal init of ``m.weight`` for Conv-like modules."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 and getattr(m, "weight", None) is not None:
        m.weight.data.normal_(mean, std)
