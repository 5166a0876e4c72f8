"""Loss all-reduce with the reference's semantics (pretraining/generative/ddputils.py:53-68):
forward = mean over ranks (x / world, then all-reduce SUM), backward = identity."""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist() else 1


def get_rank():
    return dist.get_rank() if is_dist() else 0


def is_main_process():
    return get_rank() == 0


class AllReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if is_dist() and dist.get_world_size() > 1:
            x = x.contiguous() / dist.get_world_size()
            dist.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, grads):
        return grads
