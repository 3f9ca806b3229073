from .attentions import FFN, Encoder, MultiHeadAttention  # noqa: F401
