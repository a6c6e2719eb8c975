"""Host-side timeline of the bench step: how long the host spends issuing each phase (no device sync inside a step) against the device's
step time - is the launch thread ahead of the GPU, and where does it fall behind?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from smb_vision_amd.optim import SmbvAdamW
from smb_vision_amd.masking import split_mask_indices_host
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
model, pix, mask, gen = bench.build(0, device, dict(bench.VIT_BASE))
params = list(model.parameters())
opt = SmbvAdamW(params, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)
vis_h, msk_h = split_mask_indices_host(mask.cpu())
vis_d, msk_d = vis_h.to(device), msk_h.to(device)
def step(rec):
    t0 = time.perf_counter()
    out = model(pixel_values=pix, bool_masked_pos=mask, visible_index=vis_d, masked_index=msk_d)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    t4 = time.perf_counter()
    rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
for _ in range(3):
    step([])
torch.cuda.synchronize()
rec = []
w0 = time.perf_counter()
for _ in range(10):
    step(rec)
h1 = time.perf_counter()
torch.cuda.synchronize()
w1 = time.perf_counter()
print(f"10 steps: host issue time {1e3 * (h1 - w0) / 10:.2f} ms/step, wall {1e3 * (w1 - w0) / 10:.2f} ms/step")
for i, r in enumerate(rec):
    print(f"  step {i}: forward {1e3 * r[0]:6.2f}  backward {1e3 * r[1]:6.2f}  opt.step {1e3 * r[2]:6.2f}  zero_grad {1e3 * r[3]:6.2f} ms (host)")
# device-side: time between "backward has finished" and "optimizer has finished" on the compute stream (events, no host sync inside)
evs = []
for _ in range(6):
    out = model(pixel_values=pix, bool_masked_pos=mask, visible_index=vis_d, masked_index=msk_d)
    ea, eb, ec, ed = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    ea.record()
    out.loss.backward()
    eb.record()
    opt.step()
    ec.record()
    opt.zero_grad(set_to_none=True)
    evs.append((ea, eb, ec))
torch.cuda.synchronize()
for i, (ea, eb, ec) in enumerate(evs):
    print(f"  device step {i}: backward {ea.elapsed_time(eb):7.3f} ms, optimizer tail {eb.elapsed_time(ec):7.3f} ms")
