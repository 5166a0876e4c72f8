"""Embedding-extraction throughput (clips/s) on one MI355X: VideoMAE-base encoder on all 1568 tokens of 16x224^2 clips,
mean-pool + fc_norm (the path benchmarks/compute_embeddings_videomae.py runs between curriculum stages).  Synthetic clips."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
args = ap.parse_args()
ge.build()
bvc = ge.load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = bvc.VideoMAEForVideoClassification(bvc.VideoMAEConfig(num_labels=0)).to(dev).eval()
B = args.batch
g = torch.Generator().manual_seed(1)
clips = ((torch.randint(0, 256, (B, 16, 3, 224, 224), generator=g, dtype=torch.uint8).float() / 255 - 0.5) / 0.25).to(dev)
for _ in range(args.warmup):
    out = m(pixel_values=clips).logits
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    out = m(pixel_values=clips).logits
torch.cuda.synchronize()
dt = time.perf_counter() - t0
# encoder forward on all tokens: patch embed 1568x1536x768 + 12 x (24 N D^2 + 4 N^2 D), N = 1568, D = 768
gflop = (2 * 1568 * 1536 * 768 + 12 * (24 * 1568 * 768 ** 2 + 4 * 1568 ** 2 * 768)) / 1e9
print(json.dumps({"metric": "embedding clips/s (VideoMAE-base encoder, all 1568 tokens, bf16)", "value": round(B * args.steps / dt, 1),
                  "batch": B, "ms_per_batch": round(1e3 * dt / args.steps, 3), "gflop_per_clip": round(gflop, 2),
                  "tflops": round(gflop * B * args.steps / dt / 1e3, 1), "finite": bool(torch.isfinite(out).all())}))
