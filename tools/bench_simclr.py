"""SimCLR loss throughput at the global-batch-4096 shape of BASELINE config 5: info_nce over (8192, 2048) features
(similarity GEMM 275 GFLOP forward; backward = two more products) + projection head, forward + backward."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
bvc = ge.load_package()
dev = torch.device("cuda:0")
torch.set_num_threads(8)
if "--vit" in sys.argv:
    # BASELINE config 5 on ONE GPU: ViT-B trunk (the reference's video ViT, one frame), 512 images (256 pairs), token mean,
    # head, info_nce over the local rows (N > 1 gathers them first), backward, fused SGD.  104.8 GFLOP / image.
    n = int(os.environ.get("BVC_IMAGES", "512"))
    torch.manual_seed(0)
    model = bvc.simclr.SimCLRViT("vit_base", image_size=224).to(dev).train()
    model.trunk._ensure_flat(dev)
    opt = bvc.optim.SGD([{"params": [p for p in model.trunk.parameters() if p.requires_grad]},
                         {"params": list(model.fc.parameters())}], lr=0.1, momentum=0.9, nesterov=True)
    scaler = torch.amp.GradScaler("cuda")
    masks = bvc.simclr.make_masks(n // 2, dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (n, 3, 224, 224), generator=g, dtype=torch.uint8).to(dev)     # uint8 frames, normalised on the GPU
    def vstep():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = bvc.AllReduce.apply(bvc.simclr.global_info_nce_loss(0.1, masks, model(x)))
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        return loss
    for _ in range(3):
        vstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 10
    for _ in range(it):
        loss = vstep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(json.dumps({"workload": f"SimCLR ViT-B/16 224^2 full step, {n} images/GPU", "images_per_s": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                      "tflops": round(104.8 * n / dt / 1e3, 1), "loss": round(float(loss.detach()), 4)}))
    sys.exit(0)
out = []
for n, p in ((8192, 2048), (1024, 2048), (64, 512)):
    B = n // 2
    head = bvc.simclr.ProjectionHead(p, p).to(dev)
    masks = bvc.simclr.make_masks(B, dev)
    x = torch.randn(n, p, device=dev, requires_grad=True)
    def step():
        loss = bvc.simclr.info_nce_loss(0.1, masks, head(x))
        loss.backward()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 10
    for _ in range(it):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / it * 1e3
    flops = 3 * 2.0 * n * n * p + 6 * 2.0 * n * p * p   # 3 n x n x p products + head fwd/bwd
    out.append({"rows": n, "width": p, "ms": round(ms, 3), "tflops": round(flops / ms / 1e9, 1), "loss": round(float(loss.detach()), 4)})
    print(out[-1], flush=True)
print(json.dumps({"workload": "SimCLR head + info_nce fwd+bwd", "results": out}))
