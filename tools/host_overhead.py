"""How far ahead of the GPU does the host run?  Enqueue time per step (host returns from step()) vs wall time per step, and the split
between forward call, backward call and optimizer.  If enqueue ~ wall the step is launch-bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from smb_vision_amd.optim import SmbvAdamW

device = torch.device("cuda:0")
torch.cuda.set_device(device)
model, pix, mask, n_masked = bench.build(0, device, dict(bench.VIT_BASE))
params = list(model.parameters())
opt = SmbvAdamW(params, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)

def step(tm):
    t0 = time.perf_counter()
    out = model(pixel_values=pix, bool_masked_pos=mask, num_masked=n_masked)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    opt.step()
    opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += t3 - t2

for _ in range(3):
    step([0, 0, 0])
torch.cuda.synchronize()
K = 10
tm = [0.0, 0.0, 0.0]
t0 = time.perf_counter()
for _ in range(K):
    step(tm)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_enq/K*1e3:.2f} ms/step (fwd {tm[0]/K*1e3:.2f}  bwd {tm[1]/K*1e3:.2f}  opt {tm[2]/K*1e3:.2f})   wall {t_all/K*1e3:.2f} ms/step")
# synchronous variant: GPU time of each phase when the host never runs ahead
ph = [0.0, 0.0, 0.0]
for _ in range(K):
    torch.cuda.synchronize(); a = time.perf_counter()
    out = model(pixel_values=pix, bool_masked_pos=mask, num_masked=n_masked); torch.cuda.synchronize(); b = time.perf_counter()
    out.loss.backward(); torch.cuda.synchronize(); c = time.perf_counter()
    opt.step(); opt.zero_grad(set_to_none=True); torch.cuda.synchronize(); d = time.perf_counter()
    ph[0] += b - a; ph[1] += c - b; ph[2] += d - c
print(f"synchronous phases: fwd {ph[0]/K*1e3:.2f}  bwd {ph[1]/K*1e3:.2f}  opt {ph[2]/K*1e3:.2f}  sum {sum(ph)/K*1e3:.2f} ms/step")
