"""Debug helper: checks the fp8 quantisation pre-passes byte for byte against a torch emulation, then the attention kernel against a reference built
from the SAME quantised operands."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import _lib
from smb_vision_amd.ops import _ws, _p, _stream, check
lib = _lib.load()
B, S, H = 1, 256, 2
w = H * 64
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * w, generator=g) * 1.5).bfloat16().cuda()
O = torch.zeros(B * S, w, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, S, dtype=torch.float32, device="cuda")
ws = _ws(lib.smbv_attention_fp8_workspace_bytes(B, S, H), "cuda")
base = qkv.data_ptr()
check(lib.smbv_attention_fwd_fp8(base, base + 2 * w, base + 4 * w, 3 * w, _p(O), w, _p(lse), B, S, H, 64, 0.125, 0, 0, _p(ws), ws.numel(), _stream()), "fp8")
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
rows = B * S
Sp = (S + 63) // 64 * 64
off = 0
Q8 = ws[off:off + rows * w].cpu(); off += al(rows * w)
K8 = ws[off:off + rows * w].cpu(); off += al(rows * w)
sQ = ws[off:off + rows * (w // 32)].cpu(); off += al(rows * (w // 32))
sK = ws[off:off + rows * (w // 32)].cpu(); off += al(rows * (w // 32))
V8T = ws[off:off + B * H * 64 * Sp].cpu(); off += al(B * H * 64 * Sp)
sV = ws[off:off + B * H * 64 * (Sp // 64) * 2].cpu()
f = qkv.float().cpu()
q, k, v = f[:, :w], f[:, w:2 * w], f[:, 2 * w:]
c = 0.125 * math.log2(math.e)

def mxq(x):  # x [..., 32] -> (bytes uint8, scale byte)
    amax = x.abs().amax(-1, keepdim=True)
    e = torch.ceil(torch.log2(torch.clamp(amax, min=1e-30) / 448.0))
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    q8 = (x * torch.pow(2.0, -e)).to(torch.float8_e4m3fn)
    return q8.view(torch.uint8), (e + 127).to(torch.uint8).squeeze(-1)

for name, x, got8, gots in (("Q", q * c, Q8, sQ), ("K", k, K8, sK)):
    b8, sc = mxq(x.reshape(rows, w // 32, 32))
    if name == "K":  # K scales are stored [batch, head][S][2]
        sc = sc.view(B, S, H, 2).permute(0, 2, 1, 3).contiguous()
    nb = int((b8.reshape(-1) != got8).sum()); ns = int((sc.reshape(-1) != gots).sum())
    print(f"{name}: {nb} / {b8.numel()} byte mismatches, {ns} / {sc.numel()} scale mismatches")
    if nb:
        i = int((b8.reshape(-1) != got8).nonzero()[0])
        print("   first at", i, "want", int(b8.reshape(-1)[i]), "got", int(got8[i]), "value", float((x.reshape(-1))[i]), "scale want", int(sc.reshape(-1)[i // 32]), "got", int(gots[i // 32]))
# V: [B,H,64 ch][Sp keys'] : per (channel, 32 consecutive keys) scale; inside a 64-key group byte 32 h + 16 t + 4 g + e <- key 32 t + 8 g + 4 h + e
vv = torch.zeros(B, Sp, H, 64); vv[:, :S] = v.view(B, S, H, 64)
vch = vv.permute(0, 2, 3, 1).reshape(B, H, 64, Sp // 64, 2, 32)                 # [.., grp, t, 32 keys]
b8n, sc = mxq(vch)                                                               # natural key order
order = torch.tensor([32 * t + 8 * g + 4 * h + e for h in range(2) for t in range(2) for g in range(4) for e in range(4)])
b8 = b8n.reshape(B, H, 64, Sp // 64, 64)[..., order]
scv = sc.permute(0, 1, 3, 2, 4).contiguous()   # stored [B,H][grp][ch][2]
nb = int((b8.reshape(-1) != V8T).sum()); ns = int((scv.reshape(-1) != sV).sum())
print(f"V: {nb} / {b8.numel()} byte mismatches, {ns} / {sc.numel()} scale mismatches")
# attention from the dequantised operands
def deq(b8, sc):
    return b8.view(torch.float8_e4m3fn).float() * torch.pow(2.0, sc.float() - 127).unsqueeze(-1)
qd = deq(Q8.view(rows, w // 32, 32), sQ.view(rows, w // 32)).view(B, S, H, 64).permute(0, 2, 1, 3)
kd = deq(K8.view(rows, w // 32, 32), sK.view(B, H, S, 2).permute(0, 2, 1, 3).reshape(rows, w // 32)).view(B, S, H, 64).permute(0, 2, 1, 3)
inv = torch.argsort(order)
vnat = V8T.view(B, H, 64, Sp // 64, 64)[..., inv].reshape(B, H, 64, Sp // 64, 2, 32)
vd = deq(vnat, sV.view(B, H, Sp // 64, 64, 2).permute(0, 1, 3, 2, 4)).reshape(B, H, 64, Sp)[..., :S].permute(0, 1, 3, 2)
s2 = torch.matmul(qd, kd.transpose(-1, -2))          # exp2 domain
p = torch.softmax(s2 * math.log(2.0), -1)
ref = torch.matmul(p, vd)
got = O.float().cpu().view(B, S, H, 64).permute(0, 2, 1, 3)
d = got - ref
print("O vs dequantised reference: max", float(d.abs().max()), "ref max", float(ref.abs().max()), "rms rel", float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
lref = torch.logsumexp(s2 * math.log(2.0), -1)
print("LSE max diff", float((lse.cpu() - lref).abs().max()))
print("got[0,0,:2,:8]", got[0, 0, :2, :8]); print("ref[0,0,:2,:8]", ref[0, 0, :2, :8])
