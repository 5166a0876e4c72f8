"""Host logic of the segmented optimisers (bvc_amd/optim.py), checked WITHOUT a GPU: the segment table a flat buffer is walked with
(one launch per module whatever the parameter groups - replaces the per-tensor loops of torch.optim.SGD / AdamW the reference builds at
pretraining/generative/pretrain_videomae.py:316-323 and pretraining/predictive/jepa_helper.py:95-131) and the key that rebuilds it."""
import types

import pytest
import torch


def _module(n):
    m = types.SimpleNamespace()
    m._flat = torch.zeros(n)
    m._flat_grad = torch.zeros(n)
    return m


def _param(m, off, size):
    p = torch.nn.Parameter(m._flat[off:off + size])
    p.grad = m._flat_grad[off:off + size]
    return p


def test_segment_table_covers_the_buffer_with_holes_marked(bvc):
    from bvc_amd import optim
    m = _module(5000)
    a, b, c = _param(m, 0, 1000), _param(m, 1500, 2000), _param(m, 3500, 700)
    plan = optim._Plan(m, [(3500, c, 0), (0, a, 0), (1500, b, 1)])                      # any order in
    assert plan.n == 5000 and plan.nseg == 5
    assert plan.seg_start.tolist() == [0, 1000, 1500, 3500, 4200, 5000]
    assert plan.seg_group.tolist() == [0, -1, 1, 0, -1]                                # -1: nobody's (padding, a frozen tensor) - left alone
    # first segment of every 1024-element block: the kernel walks forward from it
    assert plan.blk_seg.tolist() == [0, 1, 2, 2, 3]
    assert plan.seg_start.dtype == torch.int64 and plan.seg_group.dtype == torch.int32 and plan.blk_seg.dtype == torch.int32


def test_segment_table_edge_cases(bvc):
    from bvc_amd import optim
    m = _module(2048)
    whole = _param(m, 0, 2048)
    plan = optim._Plan(m, [(0, whole, 3)])
    assert plan.seg_start.tolist() == [0, 2048] and plan.seg_group.tolist() == [3] and plan.blk_seg.tolist() == [0, 0]
    # adjacent parameters of different groups inside one block, the last one ending at the buffer's end
    m = _module(1500)
    ps = [_param(m, 0, 10), _param(m, 10, 5), _param(m, 15, 1485)]
    plan = optim._Plan(m, [(0, ps[0], 0), (10, ps[1], 1), (15, ps[2], 0)])
    assert plan.seg_start.tolist() == [0, 10, 15, 1500] and plan.seg_group.tolist() == [0, 1, 0]
    assert plan.blk_seg.tolist() == [0, 2]
    with pytest.raises(bvc._lib.BvcError):
        optim._Plan(m, [(0, ps[0], 0), (5, ps[1], 1)])                                   # overlapping views


def test_plan_key_follows_frozen_parameters_and_regrouping(bvc):
    from bvc_amd import optim
    m = _module(4096)
    a, b, c = _param(m, 0, 1024), _param(m, 1024, 1024), _param(m, 2048, 2048)
    groups = [{"params": [a, b]}, {"params": [c]}, {"params": []}]
    k0 = optim._plans_key(groups)
    assert optim._plans_key(groups) == k0
    b.grad = None                                                                        # frozen after the first step: torch skips it
    k1 = optim._plans_key(groups)
    assert k1 != k0
    b.grad = m._flat_grad[1024:2048]
    assert optim._plans_key(groups) == k0
    assert optim._plans_key([{"params": [a]}, {"params": [b, c]}, {"params": []}]) != k0
    # more groups than the device table carries: no plan at all, every parameter on the per-run path
    many = [{"params": [_param(m, 16 * i, 16)]} for i in range(bvc._lib.OPT_MAX_GROUPS + 1)]
    plans, loose = optim._build_plans(many)
    assert plans == [] and sorted(loose) == list(range(len(many))) and all(len(v) == 1 for v in loose.values())
