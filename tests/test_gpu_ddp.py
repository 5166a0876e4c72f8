"""The data-parallel wrapper's GPU path on one device: a world-size-1 RCCL group still runs every piece the
multi-GPU job uses - parameter broadcast, bucket callbacks from the library, event fences, the communication
stream and in-place all_reduce(AVG) over RCCL - so its result must equal the unwrapped model's."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

from tests import gpu_util as G   # noqa: E402
from oracle import videomae_oracle as vo   # noqa: E402

bvc = G.bvc
dev = torch.device("cuda:0")


def test_ddp_wrapper_rccl_single_rank():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = vo.TINY
        params = vo.make_params(cfg, seed=3)
        kw = {k: v for k, v in cfg.__dict__.items() if k != "decoder_norm_eps"}
        pixels, mask = vo.synthetic_batch(cfg, 4, seed=9, mask_ratio=0.75)
        px, mk = pixels.to(dev), mask.to(dev)

        plain = bvc.VideoMAEForPreTraining(bvc.VideoMAEConfig(**kw))
        plain.load_state_dict(params)
        plain.to(dev).train()
        out = plain(px, bool_masked_pos=mk)
        out.loss.backward()
        torch.cuda.synchronize()
        ref_loss, ref_grad = float(out.loss), plain.flat_grads().clone()

        model = bvc.VideoMAEForPreTraining(bvc.VideoMAEConfig(**kw))
        model.load_state_dict(params)
        model.to(dev).train()
        ddp = bvc.DistributedDataParallel(model, device_ids=[0], output_device=0, find_unused_parameters=False,
                                          bucket_cap_mb=0.25, force_collectives=True)
        assert ddp.module is model and len(list(ddp.parameters())) == len(params)
        opt = torch.optim.SGD(ddp.parameters(), lr=0.1, momentum=0.9, nesterov=True)
        scaler = torch.amp.GradScaler("cuda")
        for _ in range(2):
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o = ddp(px, bool_masked_pos=mk)
                loss = bvc.AllReduce.apply(o.loss)
            if _ == 0:
                scaler.scale(loss).backward()
                torch.cuda.synchronize()
                g = model.flat_grads() / scaler.get_scale()
                assert abs(float(loss) - ref_loss) / ref_loss < 1e-6
                assert G.rel_err(g, ref_grad) < 1e-5
                # several buckets were reduced on the comm stream and together they tile the whole buffer
                covered = sorted(ddp.reduced_ranges)
                assert len(covered) >= 2 and covered[0][0] == 0 and covered[-1][1] == ref_grad.numel()
                assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
                scaler.step(opt)
                scaler.update()
            else:
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
        torch.cuda.synchronize()
        assert torch.isfinite(model.flat_parameters()).all()
        stats = bvc.grad_logger(ddp.module.named_parameters())
        assert stats.dec_last_layer > 0
    finally:
        dist.destroy_process_group()
