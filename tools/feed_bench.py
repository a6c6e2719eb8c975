"""PCIe-inclusive step rate: the MIM train step of bench.py with a NEW volume arriving from host memory every step.
  resident : the volume stays in HBM (what bench.py's `value` measures)
  fp32-h2d : what the reference's Trainer does - a pinned fp32 [D,1,H,W] tensor, .to(device) on the compute stream each step (335.5 MB)
  feeder   : smb_vision_amd.feed.VolumeFeeder - int16 raw volume (83.9 MB at 512x512x320... 167.8 MB), pinned ring, copy stream, device-side
             ScaleIntensityRange/pad/crop/permute, overlapped with the previous step
usage: feed_bench.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from smb_vision_amd.optim import SmbvAdamW
from smb_vision_amd.feed import VolumeFeeder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
device = torch.device("cuda:0")
torch.cuda.set_device(device)
model, pix, mask, n_masked = bench.build(0, device, dict(bench.VIT_BASE))
params = list(model.parameters())
opt = SmbvAdamW(params, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)

def train(p):
    out = model(pixel_values=p, bool_masked_pos=mask, num_masked=n_masked)
    out.loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return out.loss

def timed(name, it, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    for p in it:
        loss = train(p)
        k += 1
        if k == n:
            break
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"{name:9s}: {dt*1e3:7.2f} ms/step  {1/dt:6.2f} volumes/s   loss {float(loss):.4f}", flush=True)

for _ in range(3):
    train(pix)
timed("resident", (pix for _ in range(steps)), steps)

host32 = [torch.rand(1, 320, 1, 512, 512).pin_memory() for _ in range(2)]
def fp32_h2d():
    i = 0
    while True:
        yield host32[i & 1].to(device, non_blocking=True)
        i += 1
timed("fp32-h2d", fp32_h2d(), steps)

rng = np.random.default_rng(0)
raws = [torch.from_numpy(rng.integers(-1000, 1000, size=(512, 512, 320), dtype=np.int16)).pin_memory() for _ in range(4)]   # HU volumes as EnsureChannelFirstd leaves them, read into page-locked memory by the loader
def source():
    i = 0
    while True:
        yield raws[i & 3], None
        i += 1
feeder = VolumeFeeder(source(), device, 512, 320)
it = (p for p, _ in feeder)
for _ in range(2):
    train(next(it))
timed("feeder", it, steps)
it.close()
