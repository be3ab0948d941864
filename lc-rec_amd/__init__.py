"""lcrec_amd -- MI355X-native implementation of LC-Rec's item-indexing hot path.

Host side mirrors the reference's module API (index/models/*.py, index/trainer.py,
index/main.py, index/generate_indices.py); the arithmetic lives in
csrc/liblcrec_hip.so behind the C-ABI of include/lcrec.h.  There is no CPU path:
modules can be constructed and (de)serialised anywhere, but every forward needs a
HIP device and the built library.
"""
from . import _lib, ops  # noqa: F401
from .ops import NEARTIE_TAU  # noqa: F401
from ._lib import LcrecError  # noqa: F401
from .layers import MLPLayers, kmeans, sinkhorn_algorithm  # noqa: F401
from .vq import VectorQuantizer  # noqa: F401
from .rq import ResidualVectorQuantizer  # noqa: F401
from .rqvae import RQVAE  # noqa: F401

__version__ = "0.1.0"
