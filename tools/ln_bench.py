"""LayerNorm forward / backward at the step's shapes, standalone (preallocated outputs, HIP events), against the bytes each launch must move and a
device-to-device copy of the same byte count as the yardstick.  usage: ln_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import _lib
from smb_vision_amd.ops import _p, _stream, check
lib = _lib.load()
flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda")  # larger than the 256 MiB Infinity Cache
def t_cold(fn, iters=12):
    """per-launch HIP events with the Infinity Cache flushed (a 768 MiB fill) before every launch: what the kernel sees inside the step, where its
    operands were written many kernels ago"""
    fn(); ts = []
    for _ in range(iters):
        flush.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
def t_of(fn, iters=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for rows, D in ((7168, 768), (20480, 384), (13312, 384), (20480, 768)):
    x = torch.randn(rows, D, generator=g, device="cuda")
    gamma, beta = torch.randn(D, generator=g, device="cuda"), torch.randn(D, generator=g, device="cuda")
    dy = torch.randn(rows, D, generator=g, device="cuda").bfloat16()
    dres = torch.randn(rows, D, generator=g, device="cuda")
    y = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    dx, dxb = torch.empty(rows, D, device="cuda"), torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    dg, db = torch.empty(D, device="cuda"), torch.empty(D, device="cuda")
    nws = lib.smbv_layernorm_bwd_workspace_bytes(rows, D)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    fwd = lambda: check(lib.smbv_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, 1e-6, 0, 0, 0, _stream()), "ln_fwd")
    bwd = lambda: check(lib.smbv_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb), _p(dg), _p(db), rows, D, 0, 0, 0,
                                               _p(ws), nws, _stream()), "ln_bwd")
    fwd()
    bytes_f = rows * D * (4 + 2)
    bytes_b = rows * D * (2 + 4 + 4 + 4 + 2)
    src = torch.empty(bytes_b // 2, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
    tc = t_of(lambda: dst.copy_(src))
    tf, tb = t_of(fwd), t_of(bwd)
    tfc, tbc, tcc = t_cold(fwd), t_cold(bwd), t_cold(lambda: dst.copy_(src))
    print(f"rows {rows} D {D}: fwd {tf:6.1f} us {bytes_f/tf/1e6:5.2f} TB/s | bwd (+reduce_cols) {tb:6.1f} us {bytes_b/tb/1e6:5.2f} TB/s | copy of {bytes_b/1e6:.0f} MB (read+write) {tc:6.1f} us "
          f"{bytes_b/tc/1e6:5.2f} TB/s || cache-cold (single launches): fwd {tfc:6.1f} us {bytes_f/tfc/1e6:5.2f} TB/s | bwd {tbc:6.1f} us {bytes_b/tbc/1e6:5.2f} TB/s | copy {tcc:6.1f} us "
          f"{bytes_b/tcc/1e6:5.2f} TB/s", flush=True)
