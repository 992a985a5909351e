"""vfs_amd: the VFS (xvjiarui/VFS) training / DAVIS-evaluation hot path on MI355X (gfx950).

Host side mirrors the reference's mmaction2-style plug-in API (registries, build_model,
forward_train / forward_test / train_step); the arithmetic lives in csrc/ (hand-written HIP,
C ABI in include/vfs_hip.h).  Importing this package registers the modules; using them
requires libvfs_hip.so (python -m vfs_amd.build) -- there is no CPU fallback."""
from .builder import (build_backbone, build_head, build_loss, build_model, build_tracker)  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .registry import BACKBONES, HEADS, LOSSES, TRACKERS, Registry, build_from_cfg  # noqa: F401
from .resnet import ResNet  # noqa: F401
from .sim_loss import CosineSimLoss  # noqa: F401
from .sim_siam_head import SimSiamHead  # noqa: F401
from .trackers import BaseTracker, SimSiamBaseTracker, VanillaTracker  # noqa: F401
from .optim import SGD, build_optimizer  # noqa: F401
from .davis_eval import DavisEvaluator, evaluate_sequences  # noqa: F401
from .checkpoint import from_pretrained_keys, to_pretrained_keys  # noqa: F401
from .siamfc_heads import SiamConvFC, SiamFC  # noqa: F401
from .siamfc import SiamFCProbe  # noqa: F401
