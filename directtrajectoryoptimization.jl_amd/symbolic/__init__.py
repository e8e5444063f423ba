"""Construction-time symbolic layer (tracing, differentiation, sparsity, code emission)."""
from . import expr, diff, codegen  # noqa: F401
