"""Fused SGD / Adam / AdamW for flat parameter buffers (one HIP launch per contiguous run of parameters).

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
        self._runs = {}   # group index -> (key, runs)

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

    def _group_runs(self, gi, group):
        ps = group["params"]
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
        for gi, group in enumerate(self.param_groups):
            for run in self._group_runs(gi, group):
                n = sum(p.numel() for p in run)
                buf_ptr, first = None, 0
                if group["momentum"] != 0:
                    flat, fresh = self._momentum_buffer(run)
                    # torch's first step sets buf = g without dampening; only matters when dampening != 0
                    buf_ptr, first = flat.data_ptr(), int(fresh and group["dampening"] != 0 and found_inf is None)
                _lib.check(L.bvc_op_sgd_step(
                    run[0].data_ptr(), run[0].grad.data_ptr(), buf_ptr, n, float(group["lr"]), float(group["momentum"]),
                    float(group["dampening"]), float(group["weight_decay"]), int(group["nesterov"]), first,
                    int(group["maximize"]),
                    grad_scale.data_ptr() if grad_scale is not None else None,
                    found_inf.data_ptr() if found_inf is not None else None, 1,
                    _flat.shadow_for(run[0].data_ptr(), n), stream), "bvc_op_sgd_step")
        return loss

    def state_dict(self):
        # shallow copies: super().state_dict() hands out the LIVE per-parameter dicts, popping from them would drop the flat
        # state of the running optimiser (reallocation + host sync on the next step)
        sd = super().state_dict()
        sd["state"] = {k: {n: v for n, v in st.items() if not n.startswith("_flat_")} for k, st in sd["state"].items()}
        return sd


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's constructor, update rule and state layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter;
    the reference builds Adam / AdamW(betas=(0.9, 0.95)) at pretrain_videomae.py:190-193), as one HIP launch per contiguous
    run of parameters.  The step count lives on the device so that a step skipped by GradScaler does not advance it."""
    _step_supports_amp_scaling = True
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented (the reference does not use it)")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize))
        self._runs = {}

    _contiguous_runs = staticmethod(SGD._contiguous_runs)
    _group_runs = SGD._group_runs

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
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            for run in self._group_runs(gi, group):
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
