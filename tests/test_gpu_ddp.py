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


class _NoTorchCollectives:
    """Inside the block every torch.distributed collective raises: with the library's communicator requested (BVC_COMM=bvc) a
    training step - buckets, loose gradients, loss all-reduce, all-gather - must not touch the script's process group at all."""
    names = ("all_reduce", "broadcast", "all_gather_into_tensor", "all_gather", "reduce_scatter_tensor", "barrier")

    def __enter__(self):
        self.saved = {n: getattr(dist, n) for n in self.names}
        for n in self.names:
            def boom(*a, _n=n, **k):
                raise AssertionError(f"torch.distributed.{_n} called inside a step although BVC_COMM=bvc")
            setattr(dist, n, boom)
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(dist, n, f)
        return False


@pytest.mark.parametrize("which", ["torch", "bvc"])
def test_ddp_wrapper_rccl_single_rank(which, monkeypatch):
    import contextlib
    monkeypatch.setenv("BVC_COMM", which)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29533" if which == "torch" else "29532"
    dist.init_process_group("nccl", rank=0, world_size=1)
    guard = _NoTorchCollectives if which == "bvc" else contextlib.nullcontext
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
          with guard():
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
                # ONE communicator per step: the library's (bvc_allreduce_bucket / bvc_allreduce) or the script's process group
                assert ddp.comm_backend == ("bvc-rccl" if which == "bvc" else "torch-nccl")
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
        bvc.comm.reset()
        dist.destroy_process_group()


def _rccl_group(port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)


@pytest.mark.parametrize("which,device_ids", [("bvc", [0]), ("bvc", None), ("torch", None)])
def test_ddp_jepa_three_wraps_rccl_single_rank(which, device_ids, monkeypatch):
    """pretrain_jepa.py:302-304 with bvc's class: encoder, predictor and target encoder wrapped separately - also in the
    reference's own call form, DDP(encoder, static_graph=True) with NO device_ids, where the wrapper must find the device itself
    and set everything up (flat buffers, communicator) in its constructor, not inside the first backward.  With one rank the
    collectives are identities, so gradients must equal the unwrapped modules'; the predictor reports its gradient ranges
    per block (bvc_predictor_backward_cb), the encoder per layer, and every byte of both buffers is reduced exactly once."""
    import copy
    from oracle import jepa_oracle as jo
    monkeypatch.setenv("BVC_COMM", which)
    _rccl_group(29534)
    try:
        cfg = jo.TINY
        B, n_ctx, n_pred = 3, 6, 4
        enc_p = jo.make_params(jo.encoder_shapes(cfg), cfg, 0)
        pred_p = jo.make_params(jo.predictor_shapes(cfg), cfg, 50)
        imgs, m_enc, m_pred = jo.synthetic_inputs(cfg, B, 0, n_ctx, n_pred)
        x = imgs.to(dev)
        me, mp = [m.to(dev) for m in m_enc], [m.to(dev) for m in m_pred]
        kw = dict(img_size=[cfg.image_size], patch_size=cfg.patch_size, num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size,
                  embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio)

        def build():
            enc = bvc.jepa.VisionTransformer(**kw)
            enc.load_state_dict(enc_p)
            tgt = copy.deepcopy(enc)
            pred = bvc.jepa.vit_predictor(sequence_shape=enc.sequence_shape, embed_dim=cfg.embed_dim, predictor_embed_dim=cfg.pred_dim,
                                          depth=cfg.pred_depth, num_heads=enc.num_heads)
            pred.load_state_dict(pred_p)
            for p in tgt.parameters():
                p.requires_grad = False
            return enc.to(dev), pred.to(dev), tgt.to(dev)

        def step(enc, pred, tgt):
            with torch.no_grad():
                h = bvc.jepa.select_targets(tgt(x), mp)
            loss = bvc.jepa.smooth_l1_loss(pred(enc(x, me), me, mp), h)
            loss.backward()
            torch.cuda.synchronize()
            return float(loss)

        enc0, pred0, tgt0 = build()
        ref_loss = step(enc0, pred0, tgt0)
        enc, pred, tgt = build()
        DDP = bvc.DistributedDataParallel
        ids = dict(device_ids=device_ids) if device_ids else {}
        wenc = DDP(enc, static_graph=True, bucket_cap_mb=0.05, force_collectives=True, **ids)
        wpred = DDP(pred, static_graph=True, bucket_cap_mb=0.05, force_collectives=True, **ids)
        wtgt = DDP(tgt, force_collectives=True, **ids)
        for w in (wenc, wpred, wtgt):
            assert w._device == dev                       # found without device_ids
            assert w.comm_backend == ("bvc-rccl" if which == "bvc" else "torch-nccl")     # decided in the constructor
        assert enc._flat is not None and pred._flat is not None
        loss = step(wenc, wpred, wtgt)
        assert abs(loss - ref_loss) / ref_loss < 1e-6
        assert G.rel_err(enc.flat_grads(), enc0.flat_grads()) < 1e-6 and G.rel_err(pred.flat_grads(), pred0.flat_grads()) < 1e-6
        for w, n in ((wenc, enc.flat_grads().numel()), (wpred, pred.flat_grads().numel())):
            cov = sorted(w.reduced_ranges)
            assert len(cov) >= 2, cov                         # more than one bucket: ranges arrived per block, not once at the end
            assert cov[0][0] == 0 and cov[-1][1] == n and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
        bvc.jepa.ema_update(enc, tgt, 0.99)
        torch.cuda.synchronize()
    finally:
        bvc.comm.reset()
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["torch", "bvc"])
def test_ddp_composite_simclr_vit_rccl_single_rank(which, monkeypatch):
    """pretrain_simclr.py:227-228 for the config-5 model: SimCLRViT = flat ViT trunk + ordinary-parameter head under ONE wrapper
    (trunk gradients through the bucket hooks, head gradients through the coalesced all-reduce), global InfoNCE behind
    AllGather.  One rank: must reproduce the unwrapped model bit for bit in the loss and to round-off in the gradients."""
    monkeypatch.setenv("BVC_COMM", which)
    _rccl_group(29535)
    try:
        torch.manual_seed(0)
        B = 8

        def build():
            m = bvc.simclr.SimCLRViT.__new__(bvc.simclr.SimCLRViT)
            torch.nn.Module.__init__(m)
            m.trunk = bvc.jepa.VisionTransformer(img_size=[64], patch_size=16, num_frames=1, tubelet_size=1, embed_dim=128, depth=2, num_heads=2)
            m.fc = bvc.simclr.ProjectionHead(128, 128)
            return m

        ref = build()
        state = {k: v.clone() for k, v in ref.state_dict().items()}
        ref.to(dev).train()
        imgs = torch.randn(2 * B, 3, 64, 64, device=dev)
        masks = bvc.simclr.make_masks(B, dev)
        l0 = bvc.simclr.global_info_nce_loss(0.1, masks, ref(imgs))
        l0.backward()
        torch.cuda.synchronize()
        model = build()
        model.load_state_dict(state)
        model.to(dev).train()
        ddp = bvc.DistributedDataParallel(model, device_ids=[0], output_device=0, find_unused_parameters=False, bucket_cap_mb=0.05,
                                          force_collectives=True)
        assert len(ddp._flats) == 1 and len(ddp._loose_grad) == 4          # trunk + fc.0.weight / bias, fc.2.weight / bias
        import contextlib
        with (_NoTorchCollectives() if which == "bvc" else contextlib.nullcontext()):
            l1 = bvc.simclr.global_info_nce_loss(0.1, masks, ddp(imgs))
            l1.backward()
            torch.cuda.synchronize()
        assert float(l1) == float(l0)
        assert G.rel_err(model.trunk.flat_grads(), ref.trunk.flat_grads()) < 1e-6
        for (k, p), (_k, q) in zip(model.fc.named_parameters(), ref.fc.named_parameters()):
            assert G.rel_err(p.grad, q.grad) < 1e-6, k
        cov = sorted(ddp.reduced_ranges)
        assert cov[0][0] == 0 and cov[-1][1] == model.trunk.flat_grads().numel()
    finally:
        bvc.comm.reset()
        dist.destroy_process_group()


def test_library_communicator_single_rank(monkeypatch):
    """include/bvc.h "communication" through the Python shim: rendezvous over the process group, bucket all-reduce on the
    library's stream with its fences, the result-consumed-next collectives hopping to that stream and back."""
    monkeypatch.setenv("BVC_COMM", "bvc")
    _rccl_group(29537)
    try:
        c = bvc.comm.get(dev)
        assert c is not None and (c.rank, c.world) == (0, 1)
        assert "rccl" in c.library.lower()
        x = torch.randn(1 << 20, device=dev)
        want = x.clone()
        y = x * 2.0                      # producer kernel on the current stream; the bucket must be ordered after it
        c.allreduce_bucket(y, average=True)
        c.wait()
        z = y + 1.0                      # consumer on the current stream, ordered after the bucket by wait()
        torch.cuda.synchronize()
        assert torch.equal(z, want * 2.0 + 1.0)
        g = torch.empty(1 * 6, device=dev)
        c.allgather(torch.arange(6, device=dev, dtype=torch.float32), g)
        b = torch.arange(10, device=dev, dtype=torch.float32)
        c.broadcast(b, 0)
        s = torch.full((8,), 3.0, device=dev)
        c.allreduce(s, average=False)
        torch.cuda.synchronize()
        assert torch.equal(g, torch.arange(6, device=dev, dtype=torch.float32)) and torch.equal(b, torch.arange(10, device=dev, dtype=torch.float32))
        assert torch.equal(s, torch.full((8,), 3.0, device=dev))
        # the autograd node of the SimCLR global-batch loss takes the same route and stays the identity with one rank
        e = torch.randn(8, 16, device=dev, requires_grad=True)
        out = bvc.distributed.AllGather.apply(e)
        out.sum().backward()
        assert torch.equal(out, e) and torch.equal(e.grad, torch.ones_like(e))
    finally:
        bvc.comm.reset()
        dist.destroy_process_group()


@pytest.mark.parametrize("value", [None, "torch"])
def test_torch_distributed_is_the_default(monkeypatch, value):
    """The library communicator is opt-in (BVC_COMM=bvc) until a multi-rank run on GPUs has been recorded."""
    if value is None:
        monkeypatch.delenv("BVC_COMM", raising=False)
    else:
        monkeypatch.setenv("BVC_COMM", value)
    _rccl_group(29538)
    try:
        assert bvc.comm.get(dev) is None
    finally:
        bvc.comm.reset()
        dist.destroy_process_group()


def test_bench_line_under_the_forced_wrapper_one_rank():
    """`bench.py` exactly as the driver starts a rank (torch.distributed.run, one process), with the data-parallel wrapper forced on a
    world of one: the line must carry the communication report AND the extra legs (round 4: the extra block freed the headline model
    before the report's two profiled steps ran - a crash only this combination reached)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BVC_FORCE_DDP="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-by-batch"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["comm"]["buckets_last_step"], d.get("comm")
    assert set(d["extra"]) == {"jepa_vit_large_b16", "jepa_vit_large_b256", "simclr_vit_base_512"}
    assert not any("error" in v for v in d["extra"].values()), d["extra"]
