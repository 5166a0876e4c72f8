"""Data-parallel wrapper on CPU: 2 gloo ranks against one process with the global batch.

What DistributedDataParallel must guarantee at the reference's call sites (pretrain_videomae.py:180-181,312;
pretrain_jepa.py:302-304,426-432; pretrain_simclr.py:227-228 with pretraining/predictive/distributed.py:49-76):
parameters are broadcast from rank 0 at wrap time, and after backward every rank holds the MEAN of the per-rank gradients
(== the gradient of the global batch, since every loss here is a per-rank batch mean or a global loss behind AllGather).

Three module shapes:
  * a flat module (stand-in for VideoMAEForPreTraining that computes the oracle's gradients and reports ranges tail-first);
  * the JEPA triple: encoder + predictor wrapped separately (both flat, real FlatParamModule plumbing), target encoder wrapped
    too (never receives gradients), EMA after the step;
  * a composite (stand-in for simclr.SimCLRViT): flat trunk + ordinary nn.Module head, global-batch InfoNCE behind AllGather.
The rendezvous port is OS-assigned; a failing rank reports its traceback through the result queue.
"""
import os
import socket
import sys
import traceback

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _load():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    return ge.load_package()


# --------------------------------------------------------------------------- stand-ins
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


def _mlp(x, w1, b1, w2, b2):
    return torch.tanh(x @ w1.t() + b1) @ w2.t() + b2


def make_flat_mlp(bvc, din, dh, dout, seed):
    """The package's own FlatParamModule (flat buffer, parameter / gradient views, bucket hooks) around a two-layer MLP in
    plain torch: what jepa.VisionTransformer / VisionTransformerPredictor are to the library, minus the library."""
    flat = bvc.flat

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, anchor, model, x):
            ctx.model = model
            ctx.save_for_backward(x)
            with torch.no_grad():
                return _mlp(x, *[model._param(n) for n in model._names])

        @staticmethod
        def backward(ctx, dout):
            (x,) = ctx.saved_tensors
            m = ctx.model
            target, accumulate = m._grad_target()
            xs = x.detach().requires_grad_(True)
            ps = [m._param(n).detach().clone().requires_grad_(True) for n in m._names]
            with torch.enable_grad():
                y = _mlp(xs, *ps)
            gs = torch.autograd.grad(y, [xs] + ps, dout)
            for (name, off, shape), g in zip(m._layout, gs[1:]):
                target[off:off + g.numel()].copy_(g.reshape(-1))
            if m._bucket_hook is not None and not accumulate:      # layer 2 first, then layer 1: tail first
                half = m._layout[2][1]
                m._bucket_hook(half, m._numel - half)
                m._bucket_hook(0, half)
            m._publish_grads(target, accumulate)
            return None, None, gs[0]

    class FlatMLP(flat.FlatParamModule):
        def __init__(self):
            nn.Module.__init__(self)
            shapes = [("l1.weight", (dh, din)), ("l1.bias", (dh,)), ("l2.weight", (dout, dh)), ("l2.bias", (dout,))]
            layout, off = [], 0
            for name, shp in shapes:
                layout.append((name, off, shp))
                off += int(torch.tensor(shp).prod())
            g = torch.Generator().manual_seed(seed)
            self._init_flat(layout, off, lambda _n, shp: torch.randn(shp, generator=g) * 0.3)
            self._ensure_flat(torch.device("cpu"))

        def forward(self, x):
            return _Fn.apply(self._param("l1.weight"), self, x)

    return FlatMLP()


def _flat_grad_of(m):
    return m.flat_grads().clone()


# --------------------------------------------------------------------------- rank bodies
def _body_videomae(rank, world, bvc):
    from oracle import videomae_oracle as vo
    cfg = vo.TINY
    # rank-dependent init: the wrapper must overwrite it with rank 0's parameters
    model = FlatOracleModel(cfg, vo.make_params(cfg, seed=rank), vo)
    ddp = bvc.ddp.DistributedDataParallel(model, bucket_cap_mb=0.5)
    ref = torch.cat([v.reshape(-1) for v in vo.make_params(cfg, seed=0).values()])
    assert torch.equal(model.flat, ref), "parameters were not broadcast from rank 0"
    pixels, mask = vo.synthetic_batch(cfg, 2 * world, seed=5, mask_ratio=0.75)
    sl = slice(2 * rank, 2 * rank + 2)
    loss = model.step(pixels[sl], mask[sl])
    # loss all-reduce with the reference's semantics (ddputils.py:53-68)
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
    return (model.grad.clone(), float(mean_loss), list(ddp.reduced_ranges), extra)


JEPA_DIMS = dict(din=12, dh=16, dz=10, dp=14)


def _jepa_batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, JEPA_DIMS["din"], generator=g)


def _jepa_models(bvc, seed):
    import copy
    enc = make_flat_mlp(bvc, JEPA_DIMS["din"], JEPA_DIMS["dh"], JEPA_DIMS["dz"], seed)
    pred = make_flat_mlp(bvc, JEPA_DIMS["dz"], JEPA_DIMS["dp"], JEPA_DIMS["dz"], seed + 100)
    tgt = copy.deepcopy(enc)                        # target_encoder = copy.deepcopy(encoder), pretrain_jepa.py:258
    tgt._ensure_flat(torch.device("cpu"))
    for p in tgt.parameters():
        p.requires_grad = False
    return enc, pred, tgt


def _jepa_step(enc, pred, tgt, x, call=lambda m: m):
    with torch.no_grad():
        h = torch.nn.functional.layer_norm(call(tgt)(x), (JEPA_DIMS["dz"],))
    z = call(pred)(call(enc)(x))
    loss = torch.nn.functional.smooth_l1_loss(z, h)
    loss.backward()
    return loss


def _ema(enc, tgt, m=0.9):
    with torch.no_grad():                            # pretrain_jepa.py:426-432
        for pq, pk in zip(enc.parameters(), tgt.parameters()):
            pk.mul_(m).add_((1.0 - m) * pq.detach())


def _body_jepa(rank, world, bvc):
    DDP = bvc.ddp.DistributedDataParallel
    enc, pred, tgt = _jepa_models(bvc, seed=7 + 3 * rank)                   # rank-dependent init
    wenc, wpred, wtgt = DDP(enc, bucket_cap_mb=1e-4), DDP(pred, bucket_cap_mb=1e-4), DDP(tgt)   # the reference's three wraps
    x = _jepa_batch(4 * world, seed=11)[4 * rank:4 * rank + 4]
    loss = _jepa_step(wenc, wpred, wtgt, x)
    g_enc, g_pred = _flat_grad_of(enc), _flat_grad_of(pred)
    # a plain SGD step on the averaged gradients, then the EMA: every rank must hold the same three parameter sets
    with torch.no_grad():
        enc.flat_parameters().add_(g_enc, alpha=-0.1)
        pred.flat_parameters().add_(g_pred, alpha=-0.1)
    _ema(enc, tgt)
    views_ok = all(p.grad is not None and p.grad.data_ptr() == enc.flat_grads()[off:].data_ptr()
                   for (name, off, _s), p in zip(enc._layout, enc.parameters()))
    return (g_enc, g_pred, enc.flat_parameters().clone(), pred.flat_parameters().clone(), tgt.flat_parameters().clone(),
            float(bvc.AllReduce.apply(loss.detach())), [list(wenc.reduced_ranges), list(wpred.reduced_ranges)], views_ok,
            (wenc.joins, wpred.joins, wtgt.joins))


SIM = dict(din=10, dh=12, p=8, T=0.1, per_rank=4)


class _SimModel(nn.Module):
    """trunk (flat) + fc (ordinary parameters): the shape of simclr.SimCLRViT"""

    def __init__(self, bvc, seed):
        super().__init__()
        self.trunk = make_flat_mlp(bvc, SIM["din"], SIM["dh"], SIM["p"], seed)
        torch.manual_seed(seed + 1)
        self.fc = nn.Sequential(nn.Linear(SIM["p"], SIM["p"]), nn.ReLU(), nn.Linear(SIM["p"], SIM["p"]))

    def forward(self, x):
        return self.fc(self.trunk(x))


def _sim_batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, SIM["din"], generator=g)


def _body_simclr(rank, world, bvc):
    from oracle import simclr_oracle as so
    model = _SimModel(bvc, seed=21 + rank)                                  # rank-dependent init
    ddp = bvc.ddp.DistributedDataParallel(model, bucket_cap_mb=1e-4)
    n = 2 * SIM["per_rank"]                                                 # rows per rank (pairs interleaved)
    x = _sim_batch(n * world, seed=31)[n * rank:n * rank + n]
    feats = ddp(x)
    masks = so.make_masks(SIM["per_rank"] * world)                          # masks of the GLOBAL batch
    loss = so.info_nce_loss(SIM["T"], masks, bvc.distributed.AllGather.apply(feats))
    loss.backward()
    loose = torch.cat([p.grad.reshape(-1) for p in model.fc.parameters()])
    state = torch.cat([model.trunk.flat_parameters()] + [p.detach().reshape(-1) for p in model.fc.parameters()])
    return (_flat_grad_of(model.trunk), loose, state, float(loss))


class _CallLog:
    """Records every torch.distributed collective a rank issues, in order: (name, element count)."""
    names = ("all_reduce", "broadcast", "all_gather_into_tensor")

    def __init__(self):
        self.calls = []
        self.saved = {n: getattr(dist, n) for n in self.names}
        for n in self.names:
            def wrapped(t, *a, _n=n, _f=self.saved[n], **k):
                self.calls.append((_n, int(t.numel())))
                return _f(t, *a, **k)
            setattr(dist, n, wrapped)

    def restore(self):
        for n, f in self.saved.items():
            setattr(dist, n, f)


class _SimModelUnused(_SimModel):
    """The composite with one more ordinary parameter that `forward` uses on even steps only: a backward in which a loose
    parameter receives no gradient must neither skip the coalesced all-reduce nor shift it into the next step."""

    def __init__(self, bvc, seed):
        super().__init__(bvc, seed)
        self.extra = nn.Parameter(torch.full((SIM["p"],), 0.5))
        self.use_extra = True

    def forward(self, x):
        y = super().forward(x)
        return y + self.extra if self.use_extra else y


def _body_robust(rank, world, bvc):
    from oracle import simclr_oracle as so
    log = _CallLog()
    try:
        model = _SimModelUnused(bvc, seed=21 + rank)
        ddp = bvc.ddp.DistributedDataParallel(model, bucket_cap_mb=1e-4)
        n = 2 * SIM["per_rank"]
        masks = so.make_masks(SIM["per_rank"] * world)
        outs = []
        for it in range(3):
            model.use_extra = it != 1                     # step 1: `extra` gets no gradient on any rank
            for p in model.parameters():
                p.grad = None
            x = _sim_batch(n * world, seed=31 + it)[n * rank:n * rank + n]
            loss = so.info_nce_loss(SIM["T"], masks, bvc.distributed.AllGather.apply(ddp(x)))
            loss.backward()
            loss_all = bvc.AllReduce.apply(loss.detach())
            outs.append((_flat_grad_of(model.trunk), torch.cat([p.grad.reshape(-1) for p in model.fc.parameters()]),
                         None if model.extra.grad is None else model.extra.grad.clone(), float(loss_all)))
        return (outs, list(log.calls))
    finally:
        log.restore()


def _one_rank_losses(model, x, rank_uses_extra):
    model.use_extra = rank_uses_extra
    return model(x).pow(2).mean()


def _body_unused_on_one_rank(rank, world, bvc):
    """`extra` enters the graph on rank 0 only.  Every rank must still issue the coalesced all-reduce (a flush armed by a parameter
    hook alone would leave a rank that reached no ordinary parameter out of the collective: a hang), and rank 1 must RECEIVE the
    average for `extra` - its optimiser steps it like rank 0's, or the replicas diverge."""
    log = _CallLog()
    try:
        model = _SimModelUnused(bvc, seed=21 + rank)
        ddp = bvc.ddp.DistributedDataParallel(model, bucket_cap_mb=1e-4)
        n = 2 * SIM["per_rank"]
        outs = []
        for it in range(2):
            for p in model.parameters():
                p.grad = None
            x = _sim_batch(n * world, seed=41 + it)[n * rank:n * rank + n]
            model.use_extra = rank == 0
            ddp(x).pow(2).mean().backward()
            outs.append((_flat_grad_of(model.trunk), torch.cat([p.grad.reshape(-1) for p in model.fc.parameters()]),
                         None if model.extra.grad is None else model.extra.grad.clone()))
        return (outs, list(log.calls), ddp.joins)
    finally:
        log.restore()


class _HeadOnly(nn.Module):
    """ordinary parameters only, one of which no rank but rank 0 reaches; the trunk-less shape (no flat child ever arms the flush)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.a = nn.Linear(6, 6)
        self.b = nn.Parameter(torch.ones(6))
        self.use_b = True

    def forward(self, x):
        y = self.a(x)
        return y * self.b if self.use_b else y


def _body_head_only(rank, world, bvc):
    model = _HeadOnly()
    ddp = bvc.ddp.DistributedDataParallel(model)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4 * world, 6, generator=g)[4 * rank:4 * rank + 4]
    model.use_b = rank == 0
    ddp(x).pow(2).mean().backward()
    return (model.a.weight.grad.clone(), model.a.bias.grad.clone(), None if model.b.grad is None else model.b.grad.clone(), ddp.joins)


BODIES = {"videomae": _body_videomae, "jepa": _body_jepa, "simclr": _body_simclr, "robust": _body_robust,
          "unused_on_one_rank": _body_unused_on_one_rank, "head_only": _body_head_only}


def _freeze(o):
    """Tensors leave the rank as numpy arrays: torch shares tensor storage between processes through file descriptors that the
    RECEIVER fetches from the sender while it is still alive - a rank that has exited by then resets the connection."""
    if isinstance(o, torch.Tensor):
        return ("__tensor__", o.detach().cpu().numpy())
    if isinstance(o, (list, tuple)):
        return type(o)(_freeze(x) for x in o)
    if isinstance(o, dict):
        return {k: _freeze(v) for k, v in o.items()}
    return o


def _thaw(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == "__tensor__":
        return torch.from_numpy(o[1])
    if isinstance(o, (list, tuple)):
        return type(o)(_thaw(x) for x in o)
    if isinstance(o, dict):
        return {k: _thaw(v) for k, v in o.items()}
    return o


def _entry(body, rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        bvc = _load()
        out = BODIES[body](rank, world, bvc)
        dist.barrier()
        q.put((rank, "ok", _freeze(out)))
        dist.destroy_process_group()
    except Exception:
        q.put((rank, "error", traceback.format_exc()))


RENDEZVOUS_ERRORS = ("Address already in use", "Connection refused", "Connection reset", "Broken pipe", "timed out", "Timed out", "EADDRINUSE")


def _run(body, world=2, attempts=2):
    """One retry with a fresh port when the failure is in the TCP rendezvous (the OS-assigned port can be taken between its probe and
    rank 0's bind on a busy host); anything else fails at once with the ranks' tracebacks."""
    for attempt in range(attempts):
        try:
            return _run_once(body, world)
        except AssertionError as e:
            if attempt + 1 < attempts and any(k in str(e) for k in RENDEZVOUS_ERRORS):
                continue
            raise


def _run_once(body, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(body, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    try:
        for _ in range(world):
            got.append(q.get(timeout=420))
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.terminate()
    errs = [g for g in got if g[1] != "ok"]
    assert not errs, "\n".join(f"rank {r}:\n{tb}" for r, _s, tb in errs)
    return [_thaw(g[2]) for g in sorted(got, key=lambda t: t[0])]


# --------------------------------------------------------------------------- tests
@pytest.mark.timeout(600)
def test_two_rank_gradient_average_matches_global_batch():
    sys.path.insert(0, ROOT)
    from oracle import videomae_oracle as vo
    world = 2
    got = _run("videomae", world)
    cfg = vo.TINY
    params = vo.make_params(cfg, seed=0)
    pixels, mask = vo.synthetic_batch(cfg, 4, seed=5, mask_ratio=0.75)
    loss, grads = vo.step(cfg, params, pixels, mask)          # single process, global batch
    ref = torch.cat([grads[k].reshape(-1) for k in params])
    for rank, (g, ml, ranges, extra) in enumerate(got):
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
    assert torch.equal(got[0][0], got[1][0])


@pytest.mark.timeout(600)
def test_jepa_three_wraps_gradients_and_ema_match_global_batch():
    bvc = _load()
    world = 2
    got = _run("jepa", world)
    # single process, global batch, rank 0's initial parameters
    enc, pred, tgt = _jepa_models(bvc, seed=7)
    loss = _jepa_step(enc, pred, tgt, _jepa_batch(4 * world, seed=11))
    ge_, gp_ = _flat_grad_of(enc), _flat_grad_of(pred)
    with torch.no_grad():
        enc.flat_parameters().add_(ge_, alpha=-0.1)
        pred.flat_parameters().add_(gp_, alpha=-0.1)
    _ema(enc, tgt)
    for rank, (g_enc, g_pred, p_enc, p_pred, p_tgt, mean_loss, ranges, views_ok, joins) in enumerate(got):
        # one join per wrapper and backward, at the END of the backward: the predictor's buckets are not waited for when its own
        # backward ends (they travel under the encoder's); the target encoder never exchanges anything
        assert joins == (1, 1, 0), joins
        assert float((g_enc - ge_).norm() / ge_.norm()) < 1e-5, rank
        assert float((g_pred - gp_).norm() / gp_.norm()) < 1e-5, rank
        assert torch.allclose(p_enc, enc.flat_parameters(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(p_pred, pred.flat_parameters(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(p_tgt, tgt.flat_parameters(), rtol=1e-5, atol=1e-6)      # target / EMA state identical everywhere
        assert abs(mean_loss - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
        assert views_ok
        for r in ranges:                                  # every byte of both gradient buffers was reduced exactly once
            cov = sorted(r)
            assert cov[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
    for k in range(5):
        assert torch.equal(got[0][k], got[1][k])


@pytest.mark.timeout(600)
def test_simclr_composite_global_batch_nce_matches_single_process():
    sys.path.insert(0, ROOT)
    bvc = _load()
    from oracle import simclr_oracle as so
    world = 2
    got = _run("simclr", world)
    model = _SimModel(bvc, seed=21)                       # rank 0's initial parameters
    n = 2 * SIM["per_rank"] * world
    loss = so.info_nce_loss(SIM["T"], so.make_masks(SIM["per_rank"] * world), model(_sim_batch(n, seed=31)))
    loss.backward()
    g_trunk = _flat_grad_of(model.trunk)
    g_loose = torch.cat([p.grad.reshape(-1) for p in model.fc.parameters()])
    state = torch.cat([model.trunk.flat_parameters()] + [p.detach().reshape(-1) for p in model.fc.parameters()])
    for rank, (gt, gl, st, ls) in enumerate(got):
        assert torch.equal(st, state), "trunk + head parameters were not broadcast from rank 0"
        assert abs(ls - float(loss)) < 1e-5 * abs(float(loss))
        assert float((gt - g_trunk).norm() / g_trunk.norm()) < 2e-5, rank      # flat trunk: bucketed all-reduce
        assert float((gl - g_loose).norm() / g_loose.norm()) < 2e-5, rank      # ordinary head parameters: coalesced all-reduce
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


@pytest.mark.timeout(600)
def test_loose_parameter_without_gradient_and_collective_call_order():
    """(1) ONE communicator, one order: both ranks issue exactly the same sequence of collectives (name, size) through three
    steps - module-state broadcasts, the all-gather, trunk buckets, the coalesced head all-reduce, the loss all-reduce.
    (2) A step in which an ordinary parameter receives no gradient still averages the others (the flush is an end-of-backward
    callback, not a count of hook firings), with the same collective sizes on every rank and step, and the next step is unaffected."""
    sys.path.insert(0, ROOT)
    bvc = _load()
    from oracle import simclr_oracle as so
    world = 2
    got = _run("robust", world)
    (o0, calls0), (o1, calls1) = got
    assert calls0 == calls1 and len(calls0) > 10
    # per step: all_gather fwd, then (trunk buckets | all-gather backward | head) all_reduces, then the loss scalar last
    per_step = [c for c in calls0 if c[0] != "broadcast"]
    assert len(per_step) % 3 == 0
    k = len(per_step) // 3
    assert per_step[:k] == per_step[k:2 * k] == per_step[2 * k:]           # same collectives, same sizes, every step
    assert per_step[k - 1] == ("all_reduce", 1)                            # the loss scalar follows the step's gradient exchange
    # single process, global batch, rank 0's parameters
    model = _SimModelUnused(bvc, seed=21)
    n = 2 * SIM["per_rank"] * world
    for it in range(3):
        model.use_extra = it != 1
        for p in model.parameters():
            p.grad = None
        loss = so.info_nce_loss(SIM["T"], so.make_masks(SIM["per_rank"] * world), model(_sim_batch(n, seed=31 + it)))
        loss.backward()
        g_trunk = _flat_grad_of(model.trunk)
        g_loose = torch.cat([p.grad.reshape(-1) for p in model.fc.parameters()])
        for rank, outs in enumerate((o0, o1)):
            gt, gl, ge, ls = outs[it]
            assert float((gt - g_trunk).norm() / g_trunk.norm()) < 2e-5, (it, rank)
            assert float((gl - g_loose).norm() / g_loose.norm()) < 2e-5, (it, rank)
            assert (ge is None) == (model.extra.grad is None), (it, rank)
            if ge is not None:
                assert torch.allclose(ge, model.extra.grad, rtol=1e-4, atol=1e-6), (it, rank)
            assert abs(ls - float(loss)) < 1e-5 * abs(float(loss))


@pytest.mark.timeout(600)
def test_loose_parameter_reached_on_one_rank_only():
    """ADVICE (round 3): the coalesced all-reduce of the ordinary parameters must be symmetric across ranks whatever the rank-local
    graph reaches, and a rank whose backward left `.grad = None` must still receive the average."""
    sys.path.insert(0, ROOT)
    bvc = _load()
    world = 2
    got = _run("unused_on_one_rank", world)
    (o0, calls0, j0), (o1, calls1, j1) = got
    assert calls0 == calls1, "ranks issued different collective sequences"
    assert j0 == j1 == 2                                   # one join per backward
    model = _SimModelUnused(bvc, seed=21)                  # rank 0's parameters
    n = 2 * SIM["per_rank"]
    for it in range(2):
        xs = _sim_batch(n * world, seed=41 + it)
        want_t, want_l, want_e = 0, 0, 0
        for r in range(world):                              # mean over ranks of the per-rank gradients
            for p in model.parameters():
                p.grad = None
            _one_rank_losses(model, xs[n * r:n * r + n], r == 0).backward()
            want_t = want_t + _flat_grad_of(model.trunk) / world
            want_l = want_l + torch.cat([p.grad.reshape(-1) for p in model.fc.parameters()]) / world
            if model.extra.grad is not None:
                want_e = want_e + model.extra.grad / world
        for rank, outs in enumerate((o0, o1)):
            gt, gl, ge = outs[it]
            assert float((gt - want_t).norm() / want_t.norm()) < 2e-5, (it, rank)
            assert float((gl - want_l).norm() / want_l.norm()) < 2e-5, (it, rank)
            assert ge is not None, f"rank {rank} never received the gradient of the parameter only rank 0 reached"
            assert torch.allclose(ge, want_e, rtol=1e-4, atol=1e-7), (it, rank)


@pytest.mark.timeout(600)
def test_ordinary_module_with_a_parameter_one_rank_reaches():
    got = _run("head_only", 2)
    (w0, b0, e0, j0), (w1, b1, e1, j1) = got
    assert torch.equal(w0, w1) and torch.equal(b0, b1)
    assert e0 is not None and e1 is not None and torch.equal(e0, e1)
    assert j0 == j1 == 1
