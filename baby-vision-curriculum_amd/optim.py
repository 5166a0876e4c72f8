"""Fused SGD / Adam / AdamW for flat parameter buffers: ONE HIP launch per flat module, whatever the parameter groups.

Parameters that live in a flat module's buffer (flat.py) with their gradients in its flat gradient buffer are updated by one
launch over the whole buffer: a static device table cuts it into segments owned by a parameter group (or by none: frozen
parameters), the groups' hyper-parameters travel per call (``bvc_op_sgd_step_segments`` / ``bvc_op_adam_step_segments``,
include/bvc.h).  The reference's JEPA optimiser (pretraining/predictive/helper.py:123-147: four groups, biases and 1-D tensors
with ``weight_decay`` 0) thereby costs two launches - encoder buffer, predictor buffer - like a single group does.  Parameters
outside any flat buffer (an ordinary ``nn.Linear`` head) take one launch per run of memory-adjacent parameters of a group.


Same constructor and update rule as ``torch.optim.SGD`` (the reference builds
``torch.optim.SGD(xmodel.parameters(), lr, weight_decay, momentum, nesterov=True)`` at
pretraining/generative/pretrain_videomae.py:187-189) and the same ``state_dict`` layout
(``momentum_buffer`` per parameter).  Works with ``torch.amp.GradScaler``: it advertises
``_step_supports_amp_scaling`` so the scaler hands over its device-side ``grad_scale`` / ``found_inf``
and the step neither unscales in a separate pass nor synchronises with the host
(``scaler.step(optimizer)`` at pretrain_videomae.py:313 is unchanged).
"""
import ctypes

import torch

from . import _lib
from . import flat as _flat


def _owner(p):
    """The flat module whose buffer holds parameter `p` with its gradient at the same offset of the flat gradient buffer."""
    if p.grad is None or not p.is_cuda:
        return None
    ptr = p.data_ptr()
    for m in list(_flat._MODULES):
        f, g = getattr(m, "_flat", None), getattr(m, "_flat_grad", None)
        if f is None or g is None or not f.is_cuda:
            continue
        base = f.data_ptr()
        if base <= ptr < base + 4 * f.numel() and p.grad.data_ptr() == g.data_ptr() + (ptr - base) and ptr + 4 * p.numel() <= base + 4 * f.numel():
            return m
    return None


class _Plan:
    """Segment table of one flat module for one optimiser: device arrays + the parameters per group, built once."""

    def __init__(self, module, items):
        # items: [(offset in elements, parameter, group index)]
        items = sorted(items, key=lambda t: t[0])
        self.module, self.items = module, items
        f = module._flat
        self.n = f.numel()
        self.base, self.gbase = f.data_ptr(), module._flat_grad.data_ptr()
        starts, groups, pos = [], [], 0
        for off, prm, gi in items:
            if off < pos:
                raise _lib.BvcError("overlapping parameters in a flat buffer")
            if off > pos:
                starts.append(pos); groups.append(-1)
            starts.append(off); groups.append(gi)
            pos = off + prm.numel()
        if pos < self.n:
            starts.append(pos); groups.append(-1)
        starts.append(self.n)
        dev = f.device
        self.nseg = len(groups)
        self.seg_start = torch.tensor(starts, dtype=torch.int64, device=dev)
        self.seg_group = torch.tensor(groups, dtype=torch.int32, device=dev)
        nblk = (self.n + 1023) // 1024
        firsts = torch.arange(nblk, dtype=torch.int64, device=dev) * 1024
        self.blk_seg = (torch.searchsorted(self.seg_start, firsts, right=True) - 1).to(torch.int32)
        self.state = None      # optimiser-specific flat state


def _build_plans(param_groups):
    """-> (plans, loose): plans = one _Plan per flat module that owns parameters of this optimiser; loose = {group index: [parameters
    outside any flat buffer]}.  More groups than the table carries: everything is loose (a launch per adjacent run, as before)."""
    loose, per_module = {}, {}
    if len(param_groups) > _lib.OPT_MAX_GROUPS:
        return [], {gi: [p for p in g["params"] if p.grad is not None] for gi, g in enumerate(param_groups)}
    for gi, group in enumerate(param_groups):
        for p in group["params"]:
            if p.grad is None:
                continue
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                raise _lib.BvcError("the bvc optimisers handle f32 CUDA parameters only")
            m = _owner(p)
            if m is None:
                loose.setdefault(gi, []).append(p)
            else:
                per_module.setdefault(id(m), (m, []))[1].append(((p.data_ptr() - m._flat.data_ptr()) // 4, p, gi))
    return [_Plan(m, items) for m, items in per_module.values()], loose


def _plans_key(param_groups):
    key = []
    for g in param_groups:
        ps = g["params"]
        if not ps:
            key.append(0)
            continue
        # (the count of parameters that HAVE a gradient is part of the key: a parameter frozen after the table was built - .grad None,
        #  torch skips it - must drop out of its segment, or the kernel would keep stepping it on the stale contents of the flat buffer)
        key.append((len(ps), ps[0].data_ptr(), ps[-1].data_ptr(),
                    ps[0].grad.data_ptr() if ps[0].grad is not None else 0,
                    ps[-1].grad.data_ptr() if ps[-1].grad is not None else 0,
                    sum(1 for p in ps if p.grad is not None)))
    return tuple(key)


class SGD(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, *, maximize=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid hyper-parameter")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                        maximize=maximize)
        super().__init__(params, defaults)
        self._runs = {}   # group index -> (key, runs) for parameters outside flat buffers
        self._plans = None   # (key, plans, loose)

    def _get_plans(self):
        key = _plans_key(self.param_groups)
        if self._plans is None or self._plans[0] != key:
            plans, loose = _build_plans(self.param_groups)
            self._plans = (key, plans, loose)
            self._runs = {}
        return self._plans[1], self._plans[2]

    def _plan_momentum(self, plan):
        """One flat momentum buffer per flat module; the per-parameter ``momentum_buffer`` entries are views into it."""
        if plan.state is None:
            flat = torch.zeros(plan.n, dtype=torch.float32, device=plan.module._flat.device)
            fresh = {}
            for off, p, gi in plan.items:
                st = self.state[p]
                old = st.get("momentum_buffer")
                if old is not None:   # e.g. after load_state_dict, or a run-wise buffer of an earlier layout
                    flat[off:off + p.numel()].copy_(old.reshape(-1))
                else:
                    fresh[gi] = True
                st["momentum_buffer"] = flat[off:off + p.numel()].view(p.shape)
            plan.state = (flat, fresh)
        return plan.state

    @staticmethod
    def _contiguous_runs(params):
        """Maximal runs of parameters that are adjacent in memory with equally adjacent gradients."""
        items = sorted((p for p in params if p.grad is not None), key=lambda p: p.data_ptr())
        runs, cur = [], []
        for p in items:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                raise _lib.BvcError("bvc SGD handles f32 CUDA parameters only")
            if cur:
                q = cur[-1]
                if (q.data_ptr() + q.numel() * 4 == p.data_ptr() and q.grad.data_ptr() + q.numel() * 4 == p.grad.data_ptr()):
                    cur.append(p)
                    continue
                runs.append(cur)
            cur = [p]
        if cur:
            runs.append(cur)
        return runs

    def _group_runs(self, gi, ps):
        key = (len(ps), ps[0].data_ptr(), ps[-1].data_ptr(),
               ps[0].grad.data_ptr() if ps[0].grad is not None else 0,
               ps[-1].grad.data_ptr() if ps[-1].grad is not None else 0)
        hit = self._runs.get(gi)
        if hit is None or hit[0] != key:
            hit = (key, self._contiguous_runs(ps))
            self._runs[gi] = hit
        return hit[1]

    def _momentum_buffer(self, run):
        """One flat buffer per run; per-parameter ``momentum_buffer`` entries are views into it."""
        first = run[0]
        st = self.state[first]
        flat = st.get("_flat_momentum")
        n = sum(p.numel() for p in run)
        fresh = False
        if flat is None or flat.numel() != n:
            # zeros: with dampening == 0 the regular update of a zero buffer IS torch's first step (buf = g), and a
            # step skipped by GradScaler (found_inf) leaves a well-defined buffer behind
            flat = torch.zeros(n, dtype=torch.float32, device=first.device)
            have = all("momentum_buffer" in self.state[p] and self.state[p]["momentum_buffer"] is not None for p in run)
            o = 0
            for p in run:
                if have:   # e.g. after load_state_dict: adopt the loaded per-parameter buffers
                    flat[o:o + p.numel()].copy_(self.state[p]["momentum_buffer"].reshape(-1))
                self.state[p]["momentum_buffer"] = flat[o:o + p.numel()].view(p.shape)
                o += p.numel()
            st["_flat_momentum"] = flat
            fresh = not have
        return flat, fresh

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        L = _lib.lib()
        stream = _lib.current_stream_ptr()
        gs = grad_scale.data_ptr() if grad_scale is not None else None
        fi = found_inf.data_ptr() if found_inf is not None else None
        plans, loose = self._get_plans()
        for plan in plans:
            G = _lib.SgdGroupsC()
            G.ngroups = len(self.param_groups)
            need_buf = any(g["momentum"] != 0 for g in self.param_groups)
            flat, fresh = self._plan_momentum(plan) if need_buf else (None, {})
            for gi, g in enumerate(self.param_groups):
                G.lr[gi], G.momentum[gi], G.dampening[gi], G.weight_decay[gi] = float(g["lr"]), float(g["momentum"]), float(g["dampening"]), float(g["weight_decay"])
                G.nesterov[gi], G.maximize[gi] = int(g["nesterov"]), int(g["maximize"])
                # torch's first step sets buf = g without dampening; only matters when dampening != 0
                G.first_step[gi] = int(bool(fresh.get(gi)) and g["dampening"] != 0 and found_inf is None)
            _lib.check(L.bvc_op_sgd_step_segments(
                plan.base, plan.gbase, flat.data_ptr() if flat is not None else None, plan.n, plan.seg_start.data_ptr(),
                plan.seg_group.data_ptr(), plan.blk_seg.data_ptr(), plan.nseg, ctypes.byref(G), gs, fi, 1,
                _flat.shadow_for(plan.base, plan.n), stream), "bvc_op_sgd_step_segments")
            if fresh:
                plan.state = (flat, {})
        for gi, ps in loose.items():
            group = self.param_groups[gi]
            for run in self._group_runs(gi, ps):
                n = sum(p.numel() for p in run)
                buf_ptr, first = None, 0
                if group["momentum"] != 0:
                    flat, fresh = self._momentum_buffer(run)
                    # torch's first step sets buf = g without dampening; only matters when dampening != 0
                    buf_ptr, first = flat.data_ptr(), int(fresh and group["dampening"] != 0 and found_inf is None)
                _lib.check(L.bvc_op_sgd_step(
                    run[0].data_ptr(), run[0].grad.data_ptr(), buf_ptr, n, float(group["lr"]), float(group["momentum"]),
                    float(group["dampening"]), float(group["weight_decay"]), int(group["nesterov"]), first,
                    int(group["maximize"]), gs, fi, 1, _flat.shadow_for(run[0].data_ptr(), n), stream), "bvc_op_sgd_step")
        return loss

    def state_dict(self):
        # shallow copies: super().state_dict() hands out the LIVE per-parameter dicts, popping from them would drop the flat
        # state of the running optimiser (reallocation + host sync on the next step)
        sd = super().state_dict()
        sd["state"] = {k: {n: v for n, v in st.items() if not n.startswith("_flat_")} for k, st in sd["state"].items()}
        return sd


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's constructor, update rule and state layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter;
    the reference builds Adam / AdamW(betas=(0.9, 0.95)) at pretrain_videomae.py:190-193), as one HIP launch per flat module (per
    contiguous run for parameters outside flat buffers).  The step count lives on the device so that a step skipped by GradScaler
    does not advance it."""
    _step_supports_amp_scaling = True
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented (the reference does not use it)")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize))
        self._runs = {}
        self._plans = None

    _contiguous_runs = staticmethod(SGD._contiguous_runs)
    _group_runs = SGD._group_runs
    _get_plans = SGD._get_plans

    def _plan_state(self, plan):
        """Flat exp_avg / exp_avg_sq per flat module, the per-group step scalars (3 per group) and the f64 upload scratch."""
        if plan.state is None:
            dev = plan.module._flat.device
            m, v = torch.zeros(plan.n, dtype=torch.float32, device=dev), torch.zeros(plan.n, dtype=torch.float32, device=dev)
            state = torch.zeros(3 * _lib.OPT_MAX_GROUPS, dtype=torch.float32, device=dev)
            hyper = torch.zeros(3 * _lib.OPT_MAX_GROUPS, dtype=torch.float64, device=dev)
            for off, p, gi in plan.items:
                st = self.state[p]
                k = p.numel()
                if "exp_avg" in st:    # after load_state_dict, or state of an earlier layout: adopt it (a group shares one step count)
                    m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                    state[3 * gi] = float(st["step"])
                st["exp_avg"], st["exp_avg_sq"] = m[off:off + k].view(p.shape), v[off:off + k].view(p.shape)
                st["step"] = state[3 * gi]
            plan.state = (m, v, state, hyper)
        return plan.state

    def _run_state(self, run):
        first = run[0]
        st = self.state[first]
        flat = st.get("_flat_adam")
        n = sum(p.numel() for p in run)
        if flat is None or flat[0].numel() != n:
            dev = first.device
            m, v = torch.zeros(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)
            state3 = torch.zeros(3, dtype=torch.float32, device=dev)
            have = all("exp_avg" in self.state[p] for p in run)
            if have:    # after load_state_dict: adopt the loaded per-parameter state (all parameters share one step count)
                state3[0] = float(self.state[first]["step"])
            o = 0
            for p in run:
                k = p.numel()
                if have:
                    m[o:o + k].copy_(self.state[p]["exp_avg"].reshape(-1))
                    v[o:o + k].copy_(self.state[p]["exp_avg_sq"].reshape(-1))
                self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"] = m[o:o + k].view(p.shape), v[o:o + k].view(p.shape)
                self.state[p]["step"] = state3[0]
                o += k
            flat = (m, v, state3)
            st["_flat_adam"] = flat
        return flat

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        gs = grad_scale.data_ptr() if grad_scale is not None else None
        fi = found_inf.data_ptr() if found_inf is not None else None
        L = _lib.lib()
        stream = _lib.current_stream_ptr()
        plans, loose = self._get_plans()
        for plan in plans:
            m, v, state, hyper = self._plan_state(plan)
            G = _lib.AdamGroupsC()
            G.ngroups = len(self.param_groups)
            for gi, g in enumerate(self.param_groups):
                b1, b2 = g["betas"]
                G.lr[gi], G.beta1[gi], G.beta2[gi], G.eps[gi], G.weight_decay[gi] = float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"])
                G.decoupled[gi], G.maximize[gi] = int(self._decoupled), int(g["maximize"])
            _lib.check(L.bvc_op_adam_step_segments(
                plan.base, plan.gbase, m.data_ptr(), v.data_ptr(), plan.n, plan.seg_start.data_ptr(), plan.seg_group.data_ptr(),
                plan.blk_seg.data_ptr(), plan.nseg, ctypes.byref(G), state.data_ptr(), hyper.data_ptr(), gs, fi, 1,
                _flat.shadow_for(plan.base, plan.n), stream), "bvc_op_adam_step_segments")
        for gi, ps in loose.items():
            group = self.param_groups[gi]
            b1, b2 = group["betas"]
            for run in self._group_runs(gi, ps):
                n = sum(p.numel() for p in run)
                m, v, state3 = self._run_state(run)
                _lib.check(L.bvc_op_adam_prepare(state3.data_ptr(), float(group["lr"]), float(b1), float(b2), fi, stream),
                           "bvc_op_adam_prepare")
                _lib.check(L.bvc_op_adam_step(
                    run[0].data_ptr(), run[0].grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, float(group["lr"]), float(b1), float(b2),
                    float(group["eps"]), float(group["weight_decay"]), int(self._decoupled), int(group["maximize"]),
                    state3.data_ptr(), gs, fi, 1, _flat.shadow_for(run[0].data_ptr(), n), stream), "bvc_op_adam_step")
        return loss

    def state_dict(self):
        # shallow copies: super().state_dict() hands out the LIVE per-parameter dicts, popping from them would drop the flat
        # state of the running optimiser (reallocation + host sync on the next step)
        sd = super().state_dict()
        sd["state"] = {k: {n: v for n, v in st.items() if not n.startswith("_flat_")} for k, st in sd["state"].items()}
        return sd


class AdamW(Adam):
    """torch.optim.AdamW: decoupled weight decay, default 1e-2."""
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize)
