"""SimCLR oracle (oracle/simclr_oracle.py) against the fixture written from the reference's own info_nce_loss."""
import json
import os

import numpy as np
import torch

from oracle import simclr_oracle as so


def test_info_nce_matches_reference_fixture(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "simclr_info_nce.json")))
    for c in fx["cases"]:
        feats = so.synthetic_features(2 * c["B"], c["p"], c["seed"]).requires_grad_(True)
        if c["B"] <= 64:
            masks = so.make_masks(c["B"])
            assert int(masks[0].sum()) == c["n_pos"] and int(masks[1].sum()) == c["n_neg"]
            loss = so.info_nce_loss(c["temperature"], masks, feats)
            low = so.info_nce_loss_lowmem(c["temperature"], c["B"], feats.detach())
            assert abs(float(low) - float(loss)) < 2e-6 * abs(float(loss))       # the (n, n) form == the reference formulation
        else:      # 512 and 8192 rows (BASELINE config 5): the (n, n) form only - the broadcast product needs 2 GB / 550 GB
            n = 2 * c["B"]
            assert c["n_pos"] == 2 * (n - 1) and c["n_neg"] == n * n - n - 2 * (n - 1)
            loss = so.info_nce_loss_lowmem(c["temperature"], c["B"], feats)
        loss.backward()
        assert abs(float(loss) - c["loss"]) < 1e-5 * abs(c["loss"]) + 1e-6
        assert abs(float(feats.grad.double().norm()) - c["grad_l2"]) < 1e-4 * c["grad_l2"]
        np.testing.assert_allclose(feats.grad.flatten()[:6].numpy(), np.array(c["grad_head"]), rtol=1e-3, atol=1e-7)


def test_known_answer_counts():
    # B = 8 -> 30 positive and 210 negative entries (tridiagonal quirk of the reference's mask)
    pos, neg = so.make_masks(8)
    assert int(pos.sum()) == 30 and int(neg.sum()) == 210
    assert not (pos & neg).any() and not torch.diagonal(pos | neg).any()
