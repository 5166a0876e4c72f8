"""Which ops of the training loop body issue device-to-device / host copies (the __amd_rocclr_copyBuffer launches of the kernel stats)?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
bvc = ge.load_package()
dev = torch.device("cuda:0")
B = 16
torch.manual_seed(0)
model = bvc.VideoMAEForPreTraining(bvc.VideoMAEConfig()).to(dev).train()
model._ensure_flat(dev)
opt = bvc.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, nesterov=True, weight_decay=0.0)
scaler = bvc.amp.GradScaler("cuda")
mask_gen = bvc.TubeMaskingGenerator((8, 14, 14), 0.9, rng=np.random.RandomState(0))
clips = torch.randn(B, 16, 3, 224, 224, device=dev)
def step():
    bm = np.zeros((B, 1568))
    for i in range(B): bm[i, :] = mask_gen()
    m = torch.from_numpy(bm).bool().pin_memory().to(dev, non_blocking=True)
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(clips, bool_masked_pos=m)
    scaler.scale(out.loss).backward()
    loss = bvc.AllReduce.apply(out.loss.detach())
    scaler.step(opt); scaler.update()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.key[:70]:70s} count {e.count:5d} cpu {e.cpu_time_total:9.0f}us dev {e.device_time_total:9.0f}us")
