"""Collectives of the training step as autograd nodes, over the step's ONE communicator (comm.py): the script's process group
(torch.distributed: "nccl" = RCCL on ROCm, gloo on CPU) or, with BVC_COMM=bvc, the library's own RCCL communicator.

What the reference's entry points expect from them (interfaces only; the bodies below are this package's):
  AllReduce     pretraining/generative/ddputils.py:53-68, pretraining/predictive/distributed.py:96-112
                value -> mean over ranks, gradient passes through unchanged (the logged loss is global, the step is local)
  AllReduceSum  pretraining/predictive/distributed.py:79-93   value -> sum over ranks, gradient unchanged
  AllGather     pretraining/predictive/distributed.py:49-76   rows of every rank stacked on dim 0; backward sums the
                incoming gradient over ranks and keeps the rows this rank contributed (needed by BASELINE config 5)
With one process (or no process group) every node is the identity, so single-GPU scripts need no branches.
"""
import torch
import torch.distributed as dist

from . import comm as _comm


def world():
    """(rank, size) of the default group; (0, 1) when torch.distributed is not in use."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _reduced(t, scale):
    """Sum of `t * scale` over all ranks, as a new tensor (the collective works in place on a private copy)."""
    buf = (t * scale).contiguous() if scale != 1.0 else t.contiguous().clone()
    nc = _comm.get(buf.device, create=False) if buf.is_cuda and buf.numel() > 0 else None
    if nc is not None:
        # bvc_allreduce: on the library's communication stream, behind the gradient buckets already enqueued there (the loss
        # scalar of AllReduce follows the step's last bucket), the current stream continues after it.  ONE communicator per step
        # also for bf16 / f16 values (a half-precision loss or embedding gradient): they travel as an f32 copy instead of falling
        # back to the script's process group while buckets are in flight on the library's.
        if buf.dtype == torch.float32:
            nc.allreduce(buf.view(-1), average=False)
        else:
            wide = buf.float().contiguous()
            nc.allreduce(wide.view(-1), average=False)
            buf = wide.to(buf.dtype)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


class _PassThroughBackward(torch.autograd.Function):
    """Base of the two loss reductions: backward hands the incoming gradient on untouched."""
    scale_by_world = False

    @classmethod
    def _forward_value(cls, x):
        _, n = world()
        if n == 1:
            return x
        return _reduced(x, 1.0 / n if cls.scale_by_world else 1.0)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class AllReduce(_PassThroughBackward):
    scale_by_world = True

    @staticmethod
    def forward(ctx, x):
        return AllReduce._forward_value(x)


class AllReduceSum(_PassThroughBackward):
    scale_by_world = False

    @staticmethod
    def forward(ctx, x):
        return AllReduceSum._forward_value(x)


class AllGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        rank, n = world()
        ctx.span = (rank * x.shape[0], x.shape[0], n)
        if n == 1:
            return x
        src = x.contiguous()
        out = src.new_empty((n * src.shape[0],) + tuple(src.shape[1:]))
        nc = _comm.get(src.device, create=False) if src.is_cuda else None
        if nc is not None:
            nc.allgather(src, out)                     # bvc_allgather on the current stream (include/bvc.h)
        else:
            dist.all_gather_into_tensor(out, src)      # one contiguous destination: no list of pieces, no cat
        return out

    @staticmethod
    def backward(ctx, grad_output):
        start, rows, n = ctx.span
        if n == 1:
            return grad_output
        total = _reduced(grad_output, 1.0)
        return total.narrow(0, start, rows)
