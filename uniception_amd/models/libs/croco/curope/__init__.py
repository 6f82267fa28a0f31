from .curope2d import cuRoPE2D, cuRoPE2D_func, rope_2d  # noqa: F401
