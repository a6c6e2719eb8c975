"""Where the fed step's extra time over the resident step comes from: resident volume with a constant mask (= bench `value`), resident volume with a new
host-built mask every step, feeder volumes with a constant mask, feeder volumes with a new mask (= bench `fed`).  Same box, same process, 10 steps each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from smb_vision_amd.optim import SmbvAdamW
from smb_vision_amd.masking import split_mask_indices_host
from smb_vision_amd.feed import VolumeFeeder
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
cfg = dict(bench.VIT_BASE)
model, pix, mask, gen = bench.build(0, device, cfg)
opt = SmbvAdamW(list(model.parameters()), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)
vis_h, msk_h = split_mask_indices_host(mask.cpu())
vis_d, msk_d = vis_h.to(device), msk_h.to(device)
def train(p, m, v, k):
    out = model(pixel_values=p, bool_masked_pos=m, visible_index=v, masked_index=k)
    out.loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
def new_mask():
    m = gen().unsqueeze(0)
    v, k = split_mask_indices_host(m)
    return m.pin_memory(), v.pin_memory(), k.pin_memory()
feeder = VolumeFeeder(bench._FedSource(0, gen, cfg["image_size"], cfg["num_frames"]), device, cfg["image_size"], cfg["num_frames"])
it = iter(feeder)
def s_resident():
    train(pix, mask, vis_d, msk_d)
def s_newmask():
    m, v, k = new_mask()
    train(pix, m.to(device, non_blocking=True), v, k)
def s_feed_constmask():
    p, ex = next(it)
    train(p, mask, vis_d, msk_d)
def s_feed():
    p, ex = next(it)
    train(p, ex["mask"].to(device, non_blocking=True), ex["vis"], ex["msk"])
for name, fn in (("feeder volume, new mask (first leg)", s_feed), ("resident volume, constant mask", s_resident), ("resident volume, new mask", s_newmask), ("feeder volume, constant mask", s_feed_constmask),
                 ("feeder volume, new mask", s_feed), ("resident volume, constant mask", s_resident)):
    for _ in range(3):
        fn()
    dt = bench.timed_region(fn, 10, 1, device)
    print(f"{name:34s} {dt / 10 * 1e3:7.2f} ms/step", flush=True)
it.close()
