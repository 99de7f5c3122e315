from .fastscnn import *      # noqa: F401,F403
from .contextnet import *    # noqa: F401,F403
from . import pspnet, aspp, lednet, esnet, bisenet   # noqa: F401
from ._fused import FusedSequential, set_compute_dtype  # noqa: F401
