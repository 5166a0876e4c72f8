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
