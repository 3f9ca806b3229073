from .snake import Snake, SnakeBeta  # noqa: F401
