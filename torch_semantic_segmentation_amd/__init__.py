"""MI355X-native hot path of torch_semantic_segmentation: FastSCNN / ContextNet conv stacks as hand-written
HIP kernels (libtss_hip.so, include/tss_hip.h) behind the reference's nn.Module API."""
from . import models                                         # noqa: F401
from .models import set_compute_dtype                        # noqa: F401
from .ops import CrossEntropyLoss, cross_entropy, argmax_confusion, upsample_cross_entropy  # noqa: F401
from .ops import SyncBatchNorm, convert_syncbn_model, upsample_argmax_confusion  # noqa: F401
from .ops import OHEMLoss, ohem_loss, Upsample  # noqa: F401
from .engine import benchmark_model, GraphedInference  # noqa: F401

__version__ = '0.1.0'
