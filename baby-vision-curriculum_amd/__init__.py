"""bvc_amd -- MI355X-native (gfx950) self-supervised video pre-training step.

The directory is named ``baby-vision-curriculum_amd`` (not importable as such); it is registered under
the alias ``bvc_amd`` by ``__graft_entry__.load_package()``.

Host-side mirror of the model interface the reference's entry points use
(pretraining/generative/pretrain_videomae.py:61-64,301-302): ``VideoMAEConfig`` /
``VideoMAEForPreTraining`` with transformers' state-dict keys, whose arithmetic runs in
``libbvc_hip.so`` (hand-written HIP, C ABI in ``include/bvc.h``).  PyTorch is used for device
memory, streams, the optimiser object and torch.distributed only.
"""
from . import _lib, _ops  # noqa: F401
from .videomae import (VideoMAEConfig, VideoMAEForPreTraining, VideoMAEForPreTrainingOutput, VideoMAEForVideoClassification,  # noqa: F401
                       get_config, get_model)
from .mask import TubeMaskingGenerator, RandomMaskingGenerator  # noqa: F401
from .ddp import DistributedDataParallel  # noqa: F401
from .ddputils import AllReduce  # noqa: F401
from .loggingtools import grad_logger  # noqa: F401
from . import optim, amp  # noqa: F401
from . import simclr, distributed, jepa, jepa_mask, checkpoint, launch, comm, probe  # noqa: F401
from . import input as input_pipeline  # noqa: F401
from .input import ClipUploadRing  # noqa: F401
