"""Collective autograd functions with the reference's semantics (pretraining/predictive/distributed.py:49-112).
``AllGather`` is what the global-batch SimCLR loss (BASELINE config 5) needs: forward = all_gather + cat on dim 0,
backward = all_reduce(grads) then the rank's own slice.  torch.distributed's "nccl" backend is RCCL on ROCm."""
import torch
import torch.distributed as dist


def _on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if _on():
            x = x.contiguous()
            out = torch.empty((dist.get_world_size() * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(out, x)       # one contiguous buffer: no list of tensors + cat copy
            return out
        return x

    @staticmethod
    def backward(ctx, grads):
        if _on():
            n = grads.shape[0] // dist.get_world_size()
            grads = grads.contiguous()
            dist.all_reduce(grads)
            return grads[n * dist.get_rank(): n * (dist.get_rank() + 1)]
        return grads


class AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if _on():
            x = x.contiguous()
            dist.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, grads):
        return grads


class AllReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if _on():
            x = x.contiguous() / dist.get_world_size()
            dist.all_reduce(x)
        return x

    @staticmethod
    def backward(ctx, grads):
        return grads
