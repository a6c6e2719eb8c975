"""Is the host ahead of the GPU when the data-parallel collectives are enqueued?    One process, RCCL at world size 1
(DataParallelEngine(force_collectives=True): an averaging all-reduce over one rank is the identity, the event-gated bucket loop and the collectives are
the real ones), the bench workload (ViT-base MIM, 512x512x320, B = 1).  Per bucket: its bytes and whether its gradient-group event had NOT fired yet at the
moment its wait + all-reduce were enqueued (True = the collective was in the queue before its gradients existed: it starts the moment they do).
usage: python tools/dp_enqueue_lead.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29617")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
from smb_vision_amd.dp import DataParallelEngine
from smb_vision_amd.masking import split_mask_indices_host
from smb_vision_amd.optim import SmbvAdamW

model, pix, mask, gen = bench.build(0, device, dict(bench.VIT_BASE))
opt = SmbvAdamW(list(model.parameters()), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)
eng = DataParallelEngine(model, force_collectives=True)
t0 = time.perf_counter()
eng.broadcast_parameters(model)
torch.cuda.synchronize()
print(f"parameter broadcast: {eng.broadcast_collectives} collective(s) for {sum(p.numel() for p in model.parameters()) * 4 / 1e6:.0f} MB, {1e3 * (time.perf_counter() - t0):.1f} ms (first call: communicator set-up included)")
vis, msk = (t.to(device) for t in split_mask_indices_host(mask.cpu()))
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model(pixel_values=pix, bool_masked_pos=mask, visible_index=vis, masked_index=msk)
    t1 = time.perf_counter()
    out.loss.backward()
    t2 = time.perf_counter()
    opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    if i >= 2:
        sizes = [f"{(b - a) * 4 / 1e6:.0f}" for a, b, _ in eng._buckets]
        print(f"step {i}: host enqueue forward {1e3 * (t1 - t0):.1f} ms, backward + collectives {1e3 * (t2 - t1):.1f} ms, step wall {1e3 * (t3 - t0):.1f} ms; "
              f"{len(sizes)} buckets (MB: {' '.join(sizes)}); enqueued before their event fired: {['y' if x else 'N' for x in eng.enqueue_lead]}")
dist.destroy_process_group()
