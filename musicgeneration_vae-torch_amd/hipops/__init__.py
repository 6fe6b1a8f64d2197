"""HIP hot path of the bar VAE: ctypes binding (_native), autograd ops (functional),
flat parameter/Adam buffers (flat) and the RCCL data-parallel exchange (dist)."""
from . import functional  # noqa: F401
from .flat import FlatParams  # noqa: F401
