from .base_module import LayerNorm  # noqa: F401
