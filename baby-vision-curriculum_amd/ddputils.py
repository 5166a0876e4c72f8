"""Process-group helpers under the names the generative entry point imports (pretraining/generative/ddputils.py is where
pretrain_videomae.py:301-304 takes ``AllReduce`` from); the implementations live in ``distributed.py``."""
from .distributed import AllReduce, world  # noqa: F401


def get_rank():
    return world()[0]


def get_world_size():
    return world()[1]


def is_main_process():
    return world()[0] == 0
