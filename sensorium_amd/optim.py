"""Fused multi-tensor AdamW (+ parameter EMA) on the HIP kernel ``dwn_adamw_ema_multi``.

Semantics: ``torch.optim.AdamW`` as named by the reference config (configs/true_batch_001.py:45-48) — decoupled
weight decay, bias-corrected moments, scalar arithmetic in double on the host exactly as torch does — and,
optionally in the same pass, ``ModelEma.update`` (src/ema.py:47-55) for the parameters.  One kernel launch per
parameter group instead of ~200 tensors x several ops.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import numpy as np
import torch

from . import _lib as L

_ENTRY_DTYPE = np.dtype([("param", "<u8"), ("grad", "<u8"), ("exp_avg", "<u8"), ("exp_avg_sq", "<u8"),
                         ("ema", "<u8"), ("numel", "<i8"), ("is_int64", "<i4"), ("pad", "<i4")])
assert _ENTRY_DTYPE.itemsize == C.sizeof(L.TensorEntry)


class _TableCache:
    """Device copy of a pointer table, re-uploaded only when its contents change.

    The table of a training step is almost always identical to the previous step's (parameters, optimizer state and
    EMA tensors never move; the caching allocator hands the gradients the same blocks every step).  A blocking
    host-to-device copy per step would also stall the host until the whole backward has drained, leaving the GPU idle
    while the next step's first kernels are being queued — so a changed table goes through pinned memory, asynchronously.
    """

    def __init__(self):
        self._bytes = None
        self._dev = None
        self._pinned = None          # kept alive until the next upload: the async copy reads it

    def get(self, entries: np.ndarray, device) -> torch.Tensor:
        raw = entries.view(np.uint8).tobytes()
        if self._dev is not None and self._bytes == raw and self._dev.device == device:
            return self._dev
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8).pin_memory()
        dev = torch.empty(len(raw), dtype=torch.uint8, device=device)
        dev.copy_(host, non_blocking=True)
        self._bytes, self._dev, self._pinned = raw, dev, host
        return dev


_lerp_cache = _TableCache()


class FusedAdamWEma(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, ema_params: Optional[List[torch.Tensor]] = None,
                 ema_decay: float = 0.999, max_blocks: int = 1024):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.ema_decay = float(ema_decay)
        self.max_blocks = int(max_blocks)
        self.grad_scale = 1.0
        self._ema_of = {}
        self._ema_owner = None          # the ModelEma whose parameter copies ride in this optimizer's kernel (or None)
        self._tables = {}
        self._range_of = None
        if ema_params is not None:
            self.bind_ema(ema_params, ema_decay)

    def set_shard_map(self, range_of):
        """Sharded optimizer (ddp.GradBuckets(shard_optional=True)): ``range_of(p)`` returns ``None`` (update all of ``p``)
        or the flattened element range ``(lo, hi)`` of ``p`` this rank owns: only that range is updated (moments exist for it
        alone) and only that range of ``p.grad`` is read; the other ranks' slices arrive by all-gather."""
        self._range_of = range_of
        self._tables = {}

    def bind_ema(self, ema_params: Optional[List[torch.Tensor]], ema_decay: float, owner=None):
        """(Re)attach the EMA copies of the optimised parameters without touching the Adam moments / step counts:
        the reference assigns ``model.model_ema`` at any time (scripts/train.py:53), also after the optimizer exists.
        ``ema_params=None`` detaches (ModelEma.update then lerps the parameters itself)."""
        flat = [p for g in self.param_groups for p in g["params"]]
        if ema_params is not None and len(ema_params) != len(flat):
            raise ValueError("ema_params must align one-to-one with the optimised parameters")
        self._ema_of = {id(p): e for p, e in zip(flat, ema_params)} if ema_params is not None else {}
        self._ema_owner = owner if ema_params is not None else None
        self.ema_decay = float(ema_decay)
        self._tables = {}

    def folds_ema_of(self, owner) -> bool:
        """True when this optimizer's step also updates the parameter EMA of ``owner`` (a ModelEma)."""
        return owner is not None and self._ema_owner is owner and bool(self._ema_of)

    def state_for(self, p):
        return self.state[p]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            dev = live[0].device
            if not live[0].is_cuda:
                raise RuntimeError("FusedAdamWEma: parameters must be on a GPU (no CPU fallback)")
            # one launch per distinct step count: parameters a step left without a gradient (the other mice's readouts under
            # forward(x, index), dwiseneuro.py:404-405) are skipped like torch.optim.AdamW skips them, so their bias
            # corrections lag; ordinarily every parameter shares one count and this is a single launch
            by_step = {}
            for p in live:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdamWEma: fp32 contiguous parameters only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    rng = self._range_of(p) if self._range_of is not None else None
                    if rng is None:
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                    else:                             # moments for the owned slice only
                        st["exp_avg"] = torch.zeros(rng[1] - rng[0], dtype=p.dtype, device=p.device)
                        st["exp_avg_sq"] = torch.zeros(rng[1] - rng[0], dtype=p.dtype, device=p.device)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append(p)
            b1, b2 = group["betas"]
            for step, params in sorted(by_step.items()):
                entries = np.zeros(len(params), dtype=_ENTRY_DTYPE)
                keep = []
                n_live = 0
                for p in params:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep.append(g)
                    ema = self._ema_of.get(id(p))
                    rng = self._range_of(p) if self._range_of is not None else None
                    lo, n = (0, p.numel()) if rng is None else (rng[0], rng[1] - rng[0])
                    if n == 0:
                        continue                      # another rank's slice
                    entries[n_live] = (p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, st["exp_avg"].data_ptr(),
                                       st["exp_avg_sq"].data_ptr(), 0 if ema is None else ema.data_ptr() + 4 * lo, n, 0, 0)
                    n_live += 1
                if n_live == 0:
                    continue
                entries = entries[:n_live]
                # the table cache is keyed by the set of parameters taking part, so alternating sets do not thrash it
                key = (gi, tuple(id(p) for p in params)) if len(params) != len(group["params"]) else (gi, None)
                table = self._tables.setdefault(key, _TableCache()).get(entries, dev)
                L.check(L.lib.dwn_adamw_ema_multi(table.data_ptr(), n_live, self.max_blocks, float(group["lr"]),
                                                  float(b1), float(b2), float(group["eps"]),
                                                  float(group["weight_decay"]), int(step), self.ema_decay,
                                                  float(self.grad_scale), dev.index,
                                                  torch.cuda.current_stream(dev).cuda_stream),
                        "dwn_adamw_ema_multi")
                del keep
        return loss


def ema_lerp_state(ema_tensors: List[torch.Tensor], model_tensors: List[torch.Tensor], decay: float,
                   max_blocks: int = 16):
    """e <- decay*e + (1-decay)*m over a list of state tensors in ONE launch (src/ema.py:47-55).
    float32 tensors are lerped; int64 tensors (``num_batches_tracked``) follow the reference's float-then-truncate."""
    if not ema_tensors:
        return
    dev = ema_tensors[0].device
    entries = np.zeros(len(ema_tensors), dtype=_ENTRY_DTYPE)
    for i, (e, m) in enumerate(zip(ema_tensors, model_tensors)):
        if e.dtype == torch.int64:
            is_int = 1
        elif e.dtype == torch.float32:
            is_int = 0
        else:
            raise RuntimeError(f"ema_lerp_state: unsupported dtype {e.dtype}")
        if not (e.is_contiguous() and m.is_contiguous() and e.is_cuda and m.is_cuda):
            raise RuntimeError("ema_lerp_state: contiguous GPU tensors only")
        entries[i] = (m.data_ptr(), 0, 0, 0, e.data_ptr(), e.numel(), is_int, 0)
    table = _lerp_cache.get(entries, dev)
    L.check(L.lib.dwn_ema_lerp_multi(table.data_ptr(), len(ema_tensors), max_blocks, float(decay), dev.index,
                                     torch.cuda.current_stream(dev).cuda_stream), "dwn_ema_lerp_multi")
