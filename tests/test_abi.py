"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/bvc.h declares, the flat parameter layout carries transformers' 264 state-dict keys, and
the host-side mirrors of the reference's helpers behave like the originals.  No GPU compute."""
import os
import re

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle import videomae_oracle as vo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bvc():
    ge.build()
    return ge.load_package()


def test_header_symbols_are_exported(bvc):
    hdr = open(os.path.join(ROOT, "include", "bvc.h")).read()
    declared = set(re.findall(r"\b(bvc_[a-z0-9_]+)\s*\(", hdr)) - {"bvc_bucket_fn"}
    assert declared == set(bvc._lib.SYMBOLS), declared ^ set(bvc._lib.SYMBOLS)
    lib = bvc._lib.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.bvc_version().decode().startswith("gfx950")


def test_flat_layout_matches_transformers_state_dict(bvc):
    for cfg in (vo.BASE, vo.TINY):
        kw = {k: v for k, v in cfg.__dict__.items() if k != "decoder_norm_eps"}
        layout, total = bvc.videomae.param_layout(bvc.VideoMAEConfig(**kw))
        shapes = vo.param_shapes(cfg)
        assert {n for n, _, _ in layout} == set(shapes)
        end = 0
        for name, off, shape in layout:          # contiguous, no holes, 16-byte aligned starts
            assert off == end and off % 4 == 0
            assert tuple(shape) == tuple(shapes[name])
            end = off + int(np.prod(shape))
        assert end == total == sum(int(np.prod(s)) for s in shapes.values())
        # q | k | v weights are adjacent so the library can use them as one [3d][d] matrix
        names = [n for n, _, _ in layout]
        i = names.index("videomae.encoder.layer.0.attention.attention.query.weight")
        assert names[i + 1].endswith("key.weight") and names[i + 2].endswith("value.weight")


def test_model_object_has_reference_interface(bvc):
    class Args:
        architecture, num_frames, tubelet_size = "base", 16, 2
    model = bvc.get_model(224, Args())                       # pretrain_videomae.py:61-64
    c = model.config
    assert (c.image_size, c.patch_size, c.num_frames, c.tubelet_size) == (224, 16, 16, 2)   # :170-176
    sd = model.state_dict()
    assert len(sd) == 264 and sum(v.numel() for v in sd.values()) == 94_220_160
    assert len(list(model.parameters())) == 264
    model.load_state_dict(vo.make_params(vo.BASE, seed=0, perturb=False))   # :66-70 warm start
    assert torch.equal(model.state_dict()["decoder.head.weight"], vo.make_params(vo.BASE, 0)["decoder.head.weight"])
    model.train(); model.eval()
    with pytest.raises(bvc._lib.BvcError):                   # no CPU fallback: fails loudly off-GPU
        model(torch.zeros(1, 16, 3, 224, 224), bool_masked_pos=torch.zeros(1, 1568, dtype=torch.bool))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 16, 3, 224, 224))


def test_invalid_config_is_rejected(bvc):
    cc = bvc.VideoMAEConfig(hidden_size=100).to_c()           # head_dim != 64
    import ctypes
    assert bvc._lib.lib().bvc_videomae_param_count(ctypes.byref(cc)) < 0
    assert b"head_dim" in bvc._lib.lib().bvc_last_error()


def test_mask_generators_match_reference_fixture(bvc, golden_dir):
    import json
    fx = json.load(open(os.path.join(golden_dir, "tube_mask.json")))
    for c in fx["cases"]:
        gen = bvc.TubeMaskingGenerator(tuple(c["grid"]), c["ratio"], rng=np.random.RandomState(c["seed"]))
        assert gen.masked_per_slot == c["num_masks_per_frame"] and gen.total_masks == c["total_masks"]
        per = c["grid"][1] * c["grid"][2]
        for vis in c["visible_frame0"]:
            m = gen()
            assert [int(i) for i in np.nonzero(m[:per] == 0)[0]] == vis
    # global-RNG path (what the reference uses) is the same stream
    np.random.seed(7)
    a = bvc.TubeMaskingGenerator((8, 14, 14), 0.9)()
    b = bvc.TubeMaskingGenerator((8, 14, 14), 0.9, rng=np.random.RandomState(7))()
    assert np.array_equal(a, b) and int(a.sum()) == 1408
    r = bvc.RandomMaskingGenerator((8, 14, 14), 0.9, rng=np.random.RandomState(0))()
    assert r.shape == (1568,) and int(r.sum()) == int(0.9 * 1568)


def test_allreduce_and_grad_logger_single_process(bvc):
    x = torch.tensor(3.0, requires_grad=True)
    y = bvc.AllReduce.apply(x * 2)
    y.backward()
    assert float(y) == 6.0 and float(x.grad) == 2.0          # identity backward (ddputils.py:66-68)
    w = torch.nn.Parameter(torch.ones(2, 2)); w.grad = torch.full((2, 2), 2.0)
    s = bvc.grad_logger([("decoder.head.weight", w), ("other", w)])
    assert s.dec_last_layer == pytest.approx(4.0) and s.enc_first_layer == 0.0


def test_comm_group_binds_rccl_at_run_time_and_rejects_bad_arguments(bvc):
    """include/bvc.h "communication" without a GPU: the library carries no link-time RCCL dependency, finds the librccl.so this
    process already maps (torch's) and reports argument errors through the usual status + message channel."""
    import ctypes
    import subprocess
    lib = bvc._lib.lib()
    needed = subprocess.run(["readelf", "-d", bvc._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()
    where = lib.bvc_comm_library().decode()
    assert "librccl.so" in where and "version" in where
    h = ctypes.c_void_p()
    assert lib.bvc_comm_init(0, 1, None, ctypes.byref(h)) != 0 and b"null argument" in lib.bvc_last_error()
    ident = (ctypes.c_uint8 * 128)()
    assert lib.bvc_comm_init(3, 2, ident, ctypes.byref(h)) != 0 and b"rank 3 of 2" in lib.bvc_last_error()
    assert lib.bvc_comm_rank(None) == -1 and lib.bvc_comm_world(None) == -1 and lib.bvc_comm_destroy(None) == 0
    assert lib.bvc_allreduce_bucket(None, None, 0, 1, None) != 0 and lib.bvc_comm_wait(None, None) != 0
    # with gloo / without a process group the Python shim never creates a communicator
    assert bvc.comm.get() is None
