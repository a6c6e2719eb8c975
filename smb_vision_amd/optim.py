"""``SmbvAdamW`` — torch.optim-compatible AdamW whose step (global-norm clip + update of every parameter) is two HIP
launches in libsmbv.so (``smbv_grad_sqnorm`` + ``smbv_adamw_step``) over a device-resident chunk table.

Semantics are ``torch.nn.utils.clip_grad_norm_(params, max_grad_norm)`` followed by ``torch.optim.AdamW.step()`` — what HF
``Trainer`` runs for the reference (``scripts/training/run_mim.sh``: adamw_torch, ``--max_grad_norm 1.0``).  State is kept under
the usual keys (``step``, ``exp_avg``, ``exp_avg_sq``) so optimizer checkpoints interoperate.  With HF Trainer pass
``optimizers=(SmbvAdamW(...), None)`` and ``--max_grad_norm 0`` (clipping happens inside ``step``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

CHUNK = 1 << 16


class SmbvAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) > 8:
            raise ValueError("SmbvAdamW supports at most 8 parameter groups")
        b = {(g["betas"], g["eps"]) for g in self.param_groups}
        if len(b) != 1:
            raise ValueError("SmbvAdamW: betas/eps must be the same in every group")
        self.max_grad_norm = max_grad_norm
        self._table = None
        self._sig = None
        self.last_grad_norm = None
        self._pending_sq = None

    def _build(self, items):
        dev = items[0][0].device
        rows = []
        for p, gi in items:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            n = p.numel()
            for t in (p, p.grad, st["exp_avg"], st["exp_avg_sq"]):  # the kernels use 16-byte vector accesses
                if t.data_ptr() % 16 or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                    raise _lib.SmbvError("SmbvAdamW: parameters, gradients and moments must be contiguous, 16-byte aligned fp32 tensors on one HIP device")
            for off in range(0, n, CHUNK):
                rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr() + 4 * off,
                             st["exp_avg_sq"].data_ptr() + 4 * off, min(CHUNK, n - off), gi))
        arr = np.zeros(len(rows), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"), ("group", "<i4")]))
        for i, r in enumerate(rows):
            arr[i] = r
        assert arr.dtype.itemsize == C.sizeof(_lib.AdamChunk)
        table = torch.from_numpy(arr.view(np.uint8).copy()).to(dev)
        partial = torch.empty(len(rows), dtype=torch.float32, device=dev)
        return table, partial, len(rows)

    def _items(self):
        items = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.SmbvError("SmbvAdamW needs contiguous fp32 parameters and gradients on a HIP device")
                items.append((p, gi))
        return items

    def _chunk_table(self, items):
        # the chunk table holds raw pointers: rebuild it whenever ANY of them changes (new .grad tensors, or moments replaced by
        # load_state_dict after a step)
        def sig_of():
            return tuple((p.data_ptr(), p.grad.data_ptr(), gi, self.state[p]["exp_avg"].data_ptr() if "exp_avg" in self.state[p] else 0,
                          self.state[p]["exp_avg_sq"].data_ptr() if "exp_avg_sq" in self.state[p] else 0) for p, gi in items)
        if sig_of() != self._sig:
            self._table = self._build(items)
            self._sig = sig_of()
        return self._table

    @torch.no_grad()
    def compute_grad_norm(self):
        """Global L2 norm of the current gradients as a 0-dim device tensor (what ``clip_grad_norm_`` returns), computed by ``smbv_grad_sqnorm``.
        The following ``step()`` reuses this value for its clip instead of measuring again."""
        items = self._items()
        if not items:
            return None
        table, partial, n = self._chunk_table(items)
        sq = torch.empty(1, dtype=torch.float32, device=table.device)
        _lib.check(_lib.load().smbv_grad_sqnorm(table.data_ptr(), n, partial.data_ptr(), sq.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "smbv_grad_sqnorm")
        self._pending_sq = (self._sig, sq)
        self.last_grad_norm = sq
        return sq.sqrt().reshape(())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items = self._items()
        if not items:
            return loss
        lib = _lib.load()
        table, partial, n = self._chunk_table(items)
        for p, _ in items:
            self.state[p]["step"] += 1
        t = float(self.state[items[0][0]]["step"])
        if float(self.state[items[-1][0]]["step"]) != t:  # ONE bias correction per launch: all parameters share the step count
            raise _lib.SmbvError("SmbvAdamW: parameters are at different step counts (optimizer states mixed by hand?)")
        g0 = self.param_groups[0]
        h = _lib.AdamHyper()
        for gi, g in enumerate(self.param_groups):
            h.lr[gi] = g["lr"]
            h.weight_decay[gi] = g["weight_decay"]
        h.beta1, h.beta2, h.eps = g0["betas"][0], g0["betas"][1], g0["eps"]
        h.bias_corr1, h.bias_corr2 = 1.0 - g0["betas"][0] ** t, 1.0 - g0["betas"][1] ** t
        h.max_grad_norm = float(self.max_grad_norm) if self.max_grad_norm else 0.0
        stream = torch.cuda.current_stream().cuda_stream
        sq = None
        if h.max_grad_norm > 0:
            pend, self._pending_sq = self._pending_sq, None
            if pend is not None and pend[0] == self._sig:  # measured by compute_grad_norm() on these same gradient tensors
                sq = pend[1]
            else:
                sq = torch.empty(1, dtype=torch.float32, device=table.device)
                _lib.check(lib.smbv_grad_sqnorm(table.data_ptr(), n, partial.data_ptr(), sq.data_ptr(), stream), "smbv_grad_sqnorm")
            self.last_grad_norm = sq
        _lib.check(lib.smbv_adamw_step(table.data_ptr(), n, C.byref(h), sq.data_ptr() if sq is not None else None, stream), "smbv_adamw_step")
        return loss
