"""The oracle (oracle/videomae_oracle.py) against the committed golden fixtures.

The fixtures were written by oracle/make_golden.py from transformers 5.15.0's
VideoMAEForPreTraining (the implementation the reference's pretrain_videomae.py:61-64
instantiates) and the reference's own mask.py.  CPU only.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import videomae_oracle as vo


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


@pytest.mark.parametrize("case", ["tiny_s0", "tiny_s1", "base_b2_s0", "base_b2_s1"])
def test_oracle_matches_transformers_fixture(golden_dir, case):
    fx = _load(golden_dir, f"videomae_{case}.json")
    cfg = vo.OracleConfig(**fx["config"])
    params = vo.make_params(cfg, seed=fx["weight_seed"])
    pixels, mask = vo.synthetic_batch(cfg, fx["batch"], fx["seed"], fx["mask_ratio"])
    # the synthetic inputs themselves must be the ones the fixture was made from
    assert int(mask.sum()) == fx["input"]["mask_true"]
    assert _rel(float(pixels.double().norm()), fx["input"]["pixels"]["l2"]) < 1e-9
    assert [int(i) for i in torch.nonzero(~mask[0]).flatten()[:32]] == fx["input"]["visible_idx_row0"]

    taps = {}
    loss, grads = vo.step(cfg, params, pixels, mask, taps=taps)
    assert _rel(float(loss), fx["loss"]) < 5e-6
    for k, s in fx["taps"].items():
        t = taps[k].detach().double().flatten()
        assert t.numel() == s["numel"], k
        assert _rel(float(t.norm()), s["l2"]) < 2e-5, k
        np.testing.assert_allclose(t[:8].numpy(), np.array(s["head"]), rtol=2e-4, atol=2e-5, err_msg=k)
    gmax = max(fx["grad_l2"].values())
    for k, n in fx["grad_l2"].items():
        assert abs(float(grads[k].double().norm()) - n) <= 1e-4 * n + 1e-6 * gmax, k
    for k, n in fx["grad_probes"].items():   # grad-EFL / grad-ELL / grad-DLL of loggingtools.py:107-116
        assert _rel(float(grads[k].double().norm()), n) < 5e-5, k
        np.testing.assert_allclose(grads[k].flatten()[:4].double().numpy(), np.array(fx["grad_head"][k]),
                                   rtol=1e-3, atol=1e-7 * gmax)


def test_known_answers():
    # SURVEY.md 8c item 5
    cfg = vo.BASE
    assert cfg.seq_len == 1568 and cfg.patch_dim == 1536
    m = vo.tube_mask(cfg.grid, 0.9, np.random.RandomState(0))
    assert m.shape == (1568,) and int(m.sum()) == 1408 and int((m == 0).sum()) == 160
    assert int(0.9 * 196) == 176
    shapes = vo.param_shapes(cfg)
    assert len(shapes) == 264
    assert sum(int(np.prod(s)) for s in shapes.values()) == 94_220_160


def test_tube_mask_matches_reference_fixture(golden_dir):
    fx = _load(golden_dir, "tube_mask.json")
    for c in fx["cases"]:
        rng = np.random.RandomState(c["seed"])
        grid = tuple(c["grid"])
        per = grid[1] * grid[2]
        for vis in c["visible_frame0"]:
            m = vo.tube_mask(grid, c["ratio"], rng)
            assert int(m.sum()) == c["total_masks"]
            assert [int(i) for i in np.nonzero(m[:per] == 0)[0]] == vis
            for t in range(1, grid[0]):   # tube consistency
                assert np.array_equal(m[:per], m[t * per:(t + 1) * per])


def test_sgd_nesterov_restates_torch_optim():
    torch.manual_seed(0)
    p0 = {"a": torch.randn(5, 3), "b": torch.randn(7)}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in p0.items()}
    opt = torch.optim.SGD(ref.values(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=0.0)
    mine = {k: v.clone() for k, v in p0.items()}
    bufs = {}
    for it in range(3):
        g = {k: torch.randn_like(v) for k, v in p0.items()}
        for k in ref:
            ref[k].grad = g[k].clone()
        opt.step()
        vo.sgd_nesterov_step(mine, g, bufs, lr=0.1, momentum=0.9)
        for k in ref:
            torch.testing.assert_close(mine[k], ref[k].detach(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", ["tiny", "base"])
def test_oracle_encode_matches_transformers_fixture(golden_dir, case):
    """Encoder-only inference (benchmarks/compute_embeddings_videomae.py:78-96,253-264): oracle.encode against the
    VideoMAEForVideoClassification(num_labels=0) logits recorded by oracle/make_golden.py."""
    fx = json.load(open(os.path.join(golden_dir, "videomae_embedding.json")))["cases"][case]
    cfg = vo.TINY if case == "tiny" else vo.BASE
    params = vo.make_params(cfg, seed=fx["weight_seed"])
    g = torch.Generator().manual_seed(fx["fc_norm_seed"])
    fw = 1 + 0.1 * torch.randn(cfg.hidden_size, generator=g)
    fb = 0.05 * torch.randn(cfg.hidden_size, generator=g)
    pixels, _ = vo.synthetic_batch(cfg, fx["batch"], fx["seed"], 0.9)
    assert abs(float(pixels.double().norm()) - fx["pixels"]["l2"]) < 1e-6 * fx["pixels"]["l2"]
    with torch.no_grad():
        emb, tokens = vo.encode(cfg, params, pixels, fw, fb, fx["fc_norm_eps"])
    assert abs(float(emb.double().norm()) - fx["embedding"]["l2"]) < 1e-5 * fx["embedding"]["l2"]
    np.testing.assert_allclose(emb[0, :64].numpy(), np.array(fx["embedding_row0"]), rtol=0, atol=2e-5)
    assert abs(float(tokens.double().norm()) - fx["tokens"]["l2"]) < 1e-5 * fx["tokens"]["l2"]
