"""Data-parallel wrapper on CPU: 2 gloo ranks, a stand-in module that implements the flat-gradient
protocol (flat_parameters / flat_grads / _bucket_hook / _after_backward) with the oracle's arithmetic.

Checks what DistributedDataParallel must guarantee at pretrain_videomae.py:180-181,312: parameters are
broadcast from rank 0 at wrap time, and after backward every rank holds the MEAN of the per-rank
gradients (== the gradient of the global batch, since the loss is a per-rank batch mean).
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FlatOracleModel:
    """CPU stand-in: computes the oracle's gradients, writes them into a flat buffer and reports
    gradient ranges tail-first exactly like bvc_videomae_backward does (include/bvc.h bvc_bucket_fn)."""

    def __init__(self, cfg, params, vo, chunks=7):
        self.cfg, self.vo = cfg, vo
        self.names = list(params.keys())
        self.sizes = [params[k].numel() for k in self.names]
        self.flat = torch.cat([params[k].reshape(-1) for k in self.names]).clone()
        self.grad = torch.zeros_like(self.flat)
        self._bucket_hook = None
        self._after_backward = None
        self.chunks = chunks

    def flat_parameters(self):
        return self.flat

    def flat_grads(self):
        return self.grad

    def params(self):
        out, o = {}, 0
        for k, n in zip(self.names, self.sizes):
            out[k] = self.flat[o:o + n].view(self.vo.param_shapes(self.cfg)[k])
            o += n
        return out

    def step(self, pixels, mask):
        loss, grads = self.vo.step(self.cfg, self.params(), pixels, mask)
        self.grad.copy_(torch.cat([grads[k].reshape(-1) for k in self.names]))
        n = self.flat.numel()
        bounds = [n * i // self.chunks for i in range(self.chunks + 1)]
        for i in reversed(range(self.chunks)):           # tail first
            if self._bucket_hook:
                self._bucket_hook(bounds[i], bounds[i + 1] - bounds[i])
        if self._after_backward:
            self._after_backward()
        return loss


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    from oracle import videomae_oracle as vo
    ddp_mod = ge.load_package().ddp
    torch.set_num_threads(2)
    cfg = vo.TINY
    # rank-dependent init: the wrapper must overwrite it with rank 0's parameters
    model = FlatOracleModel(cfg, vo.make_params(cfg, seed=rank), vo)
    ddp = ddp_mod.DistributedDataParallel(model, bucket_cap_mb=0.5)
    ref = torch.cat([v.reshape(-1) for v in vo.make_params(cfg, seed=0).values()])
    assert torch.equal(model.flat, ref), "parameters were not broadcast from rank 0"
    pixels, mask = vo.synthetic_batch(cfg, 2 * world, seed=5, mask_ratio=0.75)
    sl = slice(2 * rank, 2 * rank + 2)
    loss = model.step(pixels[sl], mask[sl])
    # loss all-reduce with the reference's semantics (ddputils.py:53-68)
    bvc = ge.load_package()
    mean_loss = bvc.AllReduce.apply(loss.clone())
    # AllGather (pretraining/predictive/distributed.py:49-76): rows of every rank stacked in rank order; backward = the incoming
    # gradient summed over ranks, own rows kept.  AllReduceSum: value summed, gradient passed through.
    x = (torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * rank).requires_grad_(True)
    y = bvc.distributed.AllGather.apply(x)
    w = torch.arange(1, 1 + y.numel(), dtype=torch.float32).view_as(y)
    (y * w).sum().backward()
    s = torch.tensor(float(rank + 1), requires_grad=True)
    t = bvc.distributed.AllReduceSum.apply(s * 3)
    t.backward()
    extra = {"gathered": y.detach().clone(), "gather_grad": x.grad.clone(), "w": w, "sum": float(t), "sum_grad": float(s.grad)}
    q.put((rank, model.grad.clone(), float(mean_loss), list(ddp.reduced_ranges), extra))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_average_matches_global_batch():
    sys.path.insert(0, ROOT)
    from oracle import videomae_oracle as vo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 2, 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = vo.TINY
    params = vo.make_params(cfg, seed=0)
    pixels, mask = vo.synthetic_batch(cfg, 4, seed=5, mask_ratio=0.75)
    loss, grads = vo.step(cfg, params, pixels, mask)          # single process, global batch
    ref = torch.cat([grads[k].reshape(-1) for k in params])
    for rank, g, ml, ranges, extra in got:
        want = torch.cat([torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * r for r in range(world)])
        assert torch.equal(extra["gathered"], want)
        assert torch.equal(extra["gather_grad"], world * extra["w"][3 * rank:3 * rank + 3])
        assert extra["sum"] == 3.0 * sum(r + 1 for r in range(world)) and extra["sum_grad"] == 3.0
        assert float((g - ref).norm() / ref.norm()) < 1e-5, rank
        assert abs(ml - float(loss)) / float(loss) < 1e-6
        # ranges were coalesced into buckets, arrive tail-first and tile the buffer exactly once
        assert len(ranges) < 7
        covered = sorted(ranges)
        assert covered[0][0] == 0 and covered[-1][1] == ref.numel()
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert torch.equal(got[0][1], got[1][1])
