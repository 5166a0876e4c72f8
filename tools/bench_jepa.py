"""JEPA training-step throughput (samples/s) on one MI355X: target encoder (all 392 tokens, no grad) + context encoder +
predictor + smooth-L1 + backward + SGD-Nesterov + EMA, synthetic 2-frame 224^2 inputs, fixed mask sizes (N_ctx, N_pred)."""
import argparse, copy, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vit_base")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--nctx", type=int, default=100)
ap.add_argument("--npred", type=int, default=25)
args = ap.parse_args()
ge.build()
bvc = ge.load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc, pred = bvc.jepa.get_model(dev, patch_size=16, tubelet_size=1, num_frames=2, model_name=args.model, image_size=224)
tgt = copy.deepcopy(enc).to(dev)
for p in tgt.parameters():
    p.requires_grad = False
for m in (enc, pred, tgt):
    m._ensure_flat(dev)
opt = bvc.optim.SGD([{"params": [p for p in enc.parameters() if p.requires_grad]},
                     {"params": [p for p in pred.parameters() if p.requires_grad]}], lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-6)
scaler = torch.amp.GradScaler("cuda")
B = args.batch
g = torch.Generator().manual_seed(1)
imgs = ((torch.randint(0, 256, (B, 2, 3, 224, 224), generator=g, dtype=torch.uint8).float() / 255 - 0.5) / 0.25).to(dev)
me = [torch.stack([torch.sort(torch.randperm(196, generator=g)[:args.nctx]).values for _ in range(B)]).to(dev)]
mp = [(torch.stack([torch.sort(torch.randperm(196, generator=g)[:args.npred]).values for _ in range(B)]) + 196).to(dev) for _ in range(4)]

def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.no_grad():
            h = bvc.jepa.select_targets(tgt(imgs), mp)
        z = pred(enc(imgs, me), me, mp)
        loss = bvc.AllReduce.apply(bvc.jepa.smooth_l1_loss(z, h))
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    opt.zero_grad()
    bvc.jepa.ema_update(enc, tgt, 0.996)
    return loss

for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
gf = {"vit_base": 160.6, "vit_large": 473.2}.get(args.model, 0.0)   # GFLOP / sample at N_ctx=100, N_pred=25
print(json.dumps({"workload": f"JEPA {args.model} 2x224^2, B={B}, N_ctx={args.nctx}, N_pred={args.npred}", "samples_per_s": round(B * args.steps / dt, 1),
                  "ms_per_step": round(1e3 * dt / args.steps, 3), "tflops": round(gf * B * args.steps / dt / 1e3, 1), "loss": round(float(loss.detach()), 5)}))
