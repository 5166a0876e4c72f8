"""Headline benchmark: VideoMAE-base 16x224^2 pre-training clips/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = the reference's loop body (pretraining/generative/pretrain_videomae.py:292-317): host tube
masks -> device, zero_grad, forward, loss all-reduce, GradScaler-scaled backward (with the bucketed RCCL
gradient all-reduce overlapped when N > 1), scaler.step(SGD-Nesterov), scaler.update - on synthetic clips
that are already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

GFLOP_PER_CLIP = 202.295          # algorithmic fwd+bwd FLOPs per clip, BASELINE.md section 3
PEAK_BF16_TFLOPS = 2500.0         # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0             # HBM3E peak, MI355X_MICROARCH.md (6.29 TB/s measured by a float4 copy)


def library_hash():
    """Content hash of the HIP sources the loaded library was built from (what profiles/traffic_b*.json are stamped with)."""
    return ge._source_hash()


def measured_traffic(batch):
    """Fabric (HBM + Infinity-Cache) bytes per step from the committed PMC passes, or None.

    bench.py cannot collect TCC counters on itself; `tools/gpu_check.sh traffic` runs THIS script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes, as the
    microarchitecture guide prescribes) and tools/pmc/pmc_traffic.py reduces them (KiB -> bytes, FETCH_SIZE x2 on
    gfx950) into profiles/traffic_b<batch>.json, which is what is reported here, per step like `achieved`."""
    path = os.path.join(ROOT, "profiles", f"traffic_b{batch}.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("source_hash") != library_hash():
            # measured on other kernels than the ones that just ran: say so instead of printing stale bytes
            return None, os.path.relpath(path, ROOT) + " (stale: source hash differs)"
        return float(t["hbm_bytes_per_step"]), os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError):
        return None, None


def kernel_roofline(bvc, batch, device):
    """Per-kernel roofline of the step, measured in this run (baby-vision-curriculum_amd/probe.py): every product / attention call /
    LayerNorm of one step launched alone with HIP events on the launch stream, attributed to the kernel instantiation that runs it
    (named as rocprofv3 names it).  The FIRST row is the step's dominant kernel; the decoder fc1 + GELU product (round 2's probe)
    stays as `fc1_gelu`."""
    rows, total_us = bvc.probe.step_kernels(batch, device)
    top = rows[0]
    fc1 = next((p for r in rows for p in r["products"] if p["name"] == "dec fc1+GELU"), None)
    return {"kernel": top["kernel"], "launch_us": round(top["us_per_step"] / top["launches_per_step"], 1),
            "kernel_launches_per_step": top["launches_per_step"], "kernel_us_per_step": top["us_per_step"],
            "kernel_bound": top["bound"], "kernel_frac": top["frac"], "kernel_tflops": top["tflops"], "kernel_gb_per_s": top["gb_per_s"],
            "kernel_products": top["products"],
            "kernels": [{k: r[k] for k in ("kernel", "launches_per_step", "us_per_step", "share_of_probed", "tflops", "gb_per_s", "bound", "frac")}
                        for r in rows[:10]],
            "probed_us_per_step": round(total_us, 1), "fc1_gelu": fc1}


def load_launcher():
    """baby-vision-curriculum_amd/launch.py by path: the launcher parent must not import the package (which maps the HIP library)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bvc_launch", os.path.join(ROOT, "baby-vision-curriculum_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic_clips(batch, seed, device):
    """uint8 ~ U{0..255} frames through the loader's transform (x/255 - 0.5)/0.25 (homeview.py:218-231)."""
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, 16, 3, 224, 224), generator=g, dtype=torch.uint8)
    return ((u8.float() / 255.0 - 0.5) / 0.25).to(device)


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup quota and by the 16-core
    share a one-GPU box grants (the box shows 256 logical CPUs; oversubscribing them stalls the oracle)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(batch=4, steps=3):
    """The oracle's fp32 CPU step (forward + backward + SGD-Nesterov) on a bounded sample of the same workload."""
    from oracle import videomae_oracle as vo
    torch.set_num_threads(host_cores())
    cfg = vo.BASE
    params = vo.make_params(cfg, seed=0, perturb=False)
    pixels, mask = vo.synthetic_batch(cfg, batch, seed=1234, mask_ratio=0.9)
    bufs = {}

    def one():
        _, grads = vo.step(cfg, params, pixels, mask)
        vo.sgd_nesterov_step(params, grads, bufs, lr=0.1, momentum=0.9)

    t0 = time.perf_counter()
    one()
    warm = time.perf_counter() - t0
    if warm > 8.0:          # keep the default run within minutes on a slow host
        steps = 1
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed full steps (fwd+bwd+SGD) of VideoMAE-base at batch {batch}, fp32 torch CPU oracle, after 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256,
                    help="clips per GPU.  Throughput grows with the per-GPU batch (profiles/r02_f_batch_sweep.txt: 16 / 64 / 128 / 256 "
                         "clips -> 0.15 / 0.20 / 0.21 / 0.22 of the MFMA roof) and 288 GB of HBM holds far more than the reference's "
                         "16 clips (slurm_dev_def.bash:52), so the default is 256 (~40 GB); BASELINE.md lists 16 and 64 as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-by-batch", action="store_true", help="skip the 64- and 16-clip legs reported as `by_batch`")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the `extra` block: BASELINE configs 4 and 5 on this GPU (I-JEPA ViT-L/16 at 16 and 256 samples, SimCLR "
                         "ViT-B at 512 images), a few seconds each, after the headline loop")
    ap.add_argument("--no-probe", action="store_true",
                    help="skip the per-kernel probes behind `roofline.kernel` / `roofline.kernels` (profiling runs: only the timed steps' kernels)")
    ap.add_argument("--torch-sgd", action="store_true", help="use torch.optim.SGD instead of the fused HIP update")
    ap.add_argument("--per-step", action="store_true", help="diagnostic: per-step HIP-event and host-enqueue times to stderr")
    ap.add_argument("--bucket-mb", type=float, default=25.0, help="gradient all-reduce bucket size of the data-parallel wrapper")
    ap.add_argument("--stream-input", action="store_true",
                    help="feed every step a fresh uint8 batch from pinned host memory through the upload ring (copy stream + events) "
                         "instead of re-using clips resident in HBM; reported beside the resident-input number, never as `value`")
    args = ap.parse_args()

    # BVC_FORCE_LAUNCH=1: take the launcher path for one rank too (rehearses parent -> child -> RCCL on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("BVC_FORCE_LAUNCH")) and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (as the reference's __main__ does with mp.spawn,
        # pretrain_videomae.py:509-513).  Nothing in this process has touched the GPU; N fresh interpreters run the ranks.
        sys.exit(load_launcher().spawn_ranks([os.path.abspath(__file__), *sys.argv[1:]], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_ddp = world > 1 or bool(os.environ.get("BVC_FORCE_DDP"))   # the env switch exercises the RCCL path on one GPU
    # A one-GPU box shows 256 logical CPUs but grants a 16-core cgroup quota.  torch's default intra-op pool (one thread per
    # logical CPU) then spins in every parallel CPU op - here the (B, 1568) mask conversion once B * 1568 passes the 32768-element
    # grain - burns the quota and gets the whole process throttled for the rest of the 100 ms scheduler period: measured as a
    # 40-60 ms stall every 3rd-4th step at B >= 24 (1271 vs 2088 clips/s at B = 32, profiles/r01_d_cpu_throttle_diag.txt).
    torch.set_num_threads(min(8, host_cores()))
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # no launcher (BVC_FORCE_DDP on one GPU): an OS-assigned port, so back-to-back runs never meet in TIME_WAIT
            os.environ["MASTER_PORT"] = str(load_launcher().free_port())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    if local_rank == 0:
        ge.build()          # one builder per node; the other ranks wait for the library
    if use_ddp:
        dist.barrier()
    bvc = ge.load_package()
    torch.manual_seed(0)
    model = bvc.VideoMAEForPreTraining(bvc.VideoMAEConfig()).to(dev).train()
    model._ensure_flat(dev)
    xmodel = bvc.DistributedDataParallel(model, device_ids=[local_rank], force_collectives=use_ddp,
                                         bucket_cap_mb=args.bucket_mb) if use_ddp else model
    # same constructor arguments as the reference's torch.optim.SGD (pretrain_videomae.py:187-189); the update is one
    # HIP launch over the flat parameter buffer (--torch-sgd switches back to torch.optim.SGD)
    SGD = torch.optim.SGD if args.torch_sgd else bvc.optim.SGD
    opt = SGD(xmodel.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=0.0)
    # the reference's GradScaler object (pretrain_videomae.py:197); bvc.amp.GradScaler is the same class with its inf check
    # done as one read-only pass over the flat gradient buffer (--torch-sgd also switches back to the stock scaler)
    scaler = torch.amp.GradScaler("cuda") if args.torch_sgd else bvc.amp.GradScaler("cuda")
    B = args.batch
    mask_gen = bvc.TubeMaskingGenerator((8, 14, 14), 0.9, rng=np.random.RandomState(1234 + rank))
    ring, host_batches = None, []
    if args.stream_input:
        # what a DataLoader(pin_memory=True) of uint8 frames hands over: three different pinned batches, cycled; every step's
        # clips cross PCIe on the copy stream while the previous step computes (baby-vision-curriculum_amd/input.py)
        g = torch.Generator().manual_seed(1234 + rank)
        host_batches = [torch.randint(0, 256, (B, 16, 3, 224, 224), generator=g, dtype=torch.uint8).pin_memory() for _ in range(3)]
        ring = bvc.ClipUploadRing((B, 16, 3, 224, 224), dev, depth=3)
        ring.stage(host_batches[0])
        clips = None
    else:
        clips = synthetic_clips(B, 1234 + rank, dev)
    nstep = [0]

    def step():
        # (B and clips are re-bound by the by_batch legs below)
        bool_masked = np.zeros((B, 1568))
        for i in range(B):
            bool_masked[i, :] = mask_gen()
        # pinned + non_blocking: a pageable .to(device) makes the host wait for the stream, so it could not run ahead
        bool_masked_pos = torch.from_numpy(bool_masked).bool().pin_memory().to(dev, non_blocking=True)
        if ring is not None:
            nstep[0] += 1
            ring.stage(host_batches[nstep[0] % len(host_batches)])     # the NEXT step's clips start crossing PCIe now
            x = ring.get()
        else:
            x = clips
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = xmodel(x, bool_masked_pos=bool_masked_pos)
        # the reference reduces the loss before backward (pretrain_videomae.py:303,312); AllReduce's backward is the identity
        # (ddputils.py:64-68), so the gradients are those of the LOCAL loss either way.  Here backward is enqueued first and
        # the scalar all-reduce (only logging reads it) follows: the collective no longer sits between forward and backward.
        scaler.scale(out.loss).backward()
        loss = bvc.AllReduce.apply(out.loss.detach())
        scaler.step(opt)
        scaler.update()
        if ring is not None:
            ring.release()
        return loss

    for _ in range(args.warmup):
        step()

    def fence():
        # drain first: the barrier is a collective on the script's process group, and with BVC_COMM=bvc the step's own collectives
        # run on another communicator - the two never have work in flight together
        torch.cuda.synchronize()
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    e0.record()
    marks, host = [e0], []
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = step()
        host.append(time.perf_counter() - h0)
        if args.per_step:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
    e1.record()
    fence()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1)    # HIP events on the stream every kernel of the step is launched on
    if args.per_step and rank == 0:
        print("per-step gpu ms :", " ".join(f"{marks[i].elapsed_time(marks[i + 1]):.2f}" for i in range(args.steps)), file=sys.stderr)
        print("per-step host ms:", " ".join(f"{1e3 * h:.2f}" for h in host), file=sys.stderr)
    final_loss = float(loss.detach())
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_ddp:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    # The same loop at the reference's per-GPU batch (16 clips, slurm_dev_def.bash:52) and at round 1's default (64), in the same
    # run and the same line, so that rounds stay comparable whatever the headline batch is: 3 untimed + 10 timed steps each.
    by_batch = {}
    if world == 1 and not args.stream_input and not args.no_by_batch:
        full = clips
        for b in (64, 16):
            if b >= B:
                continue
            clips, B = full[:b], b
            for _ in range(3):
                step()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            f0.record()
            for _ in range(10):
                step()
            f1.record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - w0
            ms = f0.elapsed_time(f1) / 10
            tf = GFLOP_PER_CLIP * 1e9 * b / (ms * 1e-3) / 1e12
            tr, _src = measured_traffic(b)
            by_batch[str(b)] = {"value": round(b * 10 / wall, 2), "unit": "clips/s", "ms_per_step": round(1e3 * wall / 10, 4), "steps": 10,
                                "roofline": {"bound": "mfma", "achieved": round(tf, 2), "frac": round(tf / PEAK_BF16_TFLOPS, 4), "traffic": tr}}
        clips, B = full, args.batch
    comm = None
    if use_ddp:
        # two more steps OUTSIDE the timed region with per-bucket events on the communication stream: bytes, time, ring bus bandwidth
        xmodel.profile_buckets = True
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        rep = xmodel.bucket_report()
        native = bvc.comm.get(dev)
        comm = {"backend": ("rccl via libbvc_hip.so (bvc_allreduce_bucket / bvc_allreduce; BVC_COMM=bvc): " + native.library)
                if xmodel.comm_backend == "bvc-rccl" else "rccl via torch.distributed (nccl), the script's process group (default)",
                "communicators_in_step": 1,
                "ranks": dist.get_world_size(), "bucket_cap_mb": args.bucket_mb, "buckets_last_step": rep[-1] if rep else [],
                # per rank: what of the gradient exchange is NOT hidden under backward (end of the last bucket's collective minus end of
                # the backward's kernels, ms; rank 0's value here, max over ranks below).  Filled by the first run with more than one
                # rank on GPUs - none has been recorded yet ("what is and is not known about scaling")
                "exposed_comm_ms": (xmodel.exposed_comm_report() or [None])[-1]}
        if comm["exposed_comm_ms"] is not None and world > 1:
            t = torch.tensor([comm["exposed_comm_ms"]], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            comm["exposed_comm_ms_max_over_ranks"] = round(float(t), 4)
        nc = native if xmodel.comm_backend == "bvc-rccl" else None
        comm["rccl_ranks"] = nc.world if nc is not None and hasattr(nc, "world") else dist.get_world_size()
        xmodel.profile_buckets = False
    # BASELINE configs 4 / 5 on this GPU, outside every timed region of the headline metric (tools/bench_legs.py): their own
    # metrics, units and FLOP counts; never part of `value`
    # (after the data-parallel report above: this block frees the headline model)
    extra = None
    if world == 1 and not args.stream_input and not args.no_extra:
        from tools import bench_legs
        del clips
        model = xmodel = opt = None
        torch.cuda.empty_cache()
        extra = {}
        for key, fn in (("jepa_vit_large_b16", lambda: bench_legs.jepa_leg(bvc, dev, "vit_large", 16)),
                        ("jepa_vit_large_b256", lambda: bench_legs.jepa_leg(bvc, dev, "vit_large", 256, steps=5)),
                        ("simclr_vit_base_512", lambda: bench_legs.simclr_leg(bvc, dev, 512))):
            try:
                extra[key] = fn()
            except Exception as e:      # noqa: BLE001 - the headline line must not depend on an extra leg
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
    stream_info = None
    if ring is not None:
        stream_info = {"bytes_per_step": B * 16 * 3 * 224 * 224, "ring_depth": ring.depth,
                       "h2d_gb_per_s_sustained": round(B * 16 * 3 * 224 * 224 * args.steps / dt / 1e9, 2)}

    if rank == 0:
        clips_s = world * B * args.steps / dt
        step_ms_gpu = gpu_ms / args.steps
        achieved = GFLOP_PER_CLIP * 1e9 * B / (step_ms_gpu * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(B) if world == 1 else (None, None)
        line = {
            "metric": "video clips/sec (node) VideoMAE-base 16x224^2 bf16 pretraining",
            "value": round(clips_s, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "VideoMAE-base (ViT-B/16 encoder, 4-layer decoder), 16x224x224 clips, tubelet 2, 90% tube mask, "
                                   "full training step (fwd + MSE + bwd + GradScaler + SGD-Nesterov)",
                       "clips_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}",
                       "weights": "random init N(0,0.02), seed 0", "final_loss": round(final_loss, 5)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "scope": "one training step = the fwd+bwd kernel sequence of libbvc_hip.so on the compute stream",
                         "flops_per_launch": GFLOP_PER_CLIP * 1e9 * B, "launch_ms": round(step_ms_gpu, 4)},
        }
        if world == 1:
            # the dominant kernel of the step = the kernel instantiation with the largest time share, from per-product launches
            # timed alone in this run and named as rocprofv3 names them (compare: profiles/r03_*_roofline_table_b256.txt row 1)
            if not args.no_probe:
                line["roofline"].update(kernel_roofline(bvc, B, dev))
            if by_batch:
                line["by_batch"] = by_batch
            if extra:
                line["extra"] = extra
        if comm is not None:
            line["comm"] = comm
        if stream_info is not None:
            line["input_stream"] = stream_info
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
