"""Data-parallel gradient exchange for the DwiseNeuro step: one process per GPU, flat fp32 gradient buckets,
RCCL all-reduce over xGMI launched from autograd hooks so it overlaps with the rest of backward.

The reference has no distributed code; this is the one exchange step data parallelism adds:
an all-reduce (mean) of the parameter gradients per iteration.  Buckets are built in *reverse* registration
order — readouts (≈16 M params each, ready first in backward) lead, then the cortex and the ~200 small core tensors
in ≈12 MB buckets — so the big transfers are issued while the core backward (≈95 % of the step) is still running
and only the last small bucket (first blocks + stem) is exposed after it.
BatchNorm statistics stay local to each rank (standard DDP semantics).

Parameters that a step may legitimately leave without a gradient — the per-mouse readouts when the model is called as
``forward(x, index)`` (dwiseneuro.py:404-405) — live in *optional* buckets: those are not launched by their own hooks
(ranks training different mice would issue the collectives in different orders) but all together, in bucket order and with
zeros for the missing gradients, at the moment the first mandatory bucket (the cortex: always used, and complete only after
every readout's backward) is launched — the same point of the backward pass on every rank.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, module: torch.nn.Module, bucket_cap_mb: float = 12.0, process_group=None,
                 broadcast_init: bool = True, optional_prefixes=("readouts.",)):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        params = [p for _, p in named]
        optional = {id(p) for n, p in named if any(n.startswith(pre) for pre in optional_prefixes)}
        if self.world > 1 and broadcast_init:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
        self.buckets: List[dict] = []
        cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        cur: List[torch.nn.Parameter] = []
        cur_n = 0
        for p in reversed(params):
            kind_changes = bool(cur) and ((id(p) in optional) != (id(cur[0]) in optional))
            if cur and (kind_changes or (cur_n + p.numel() > cap and cur_n * 8 > cap)):      # tiny leftovers ride with the next tensor
                self._add_bucket(cur, id(cur[0]) in optional)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._add_bucket(cur, id(cur[0]) in optional)
        self._handles: List = []
        self._hooks = []
        if self.world > 1:
            for bi, b in enumerate(self.buckets):
                for p in b["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _add_bucket(self, params, optional: bool = False):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.buckets.append(dict(params=params, flat=flat, pending=len(params), count=len(params), optional=optional,
                                 launched=False))

    def _views(self, b):
        off = 0
        for p in b["params"]:
            yield p, b["flat"][off:off + p.numel()].view_as(p)
            off += p.numel()

    def _launch(self, bi: int):
        b = self.buckets[bi]
        if b["launched"]:
            return
        b["launched"] = True
        # gather the bucket's gradients with ONE concatenation (autograd handed each parameter a fresh tensor;
        # accumulating ~200 gradients into pre-assigned views would cost one small add kernel per parameter);
        # parameters of an optional bucket that took no part in this step contribute zeros
        parts = [p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), dtype=p.dtype, device=p.device)
                 for p in b["params"]]
        torch.cat(parts, out=b["flat"])
        op = dist.ReduceOp.AVG if dist.get_backend(self.pg) == "nccl" else dist.ReduceOp.SUM
        self._handles.append((dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=True), bi, op))

    def _launch_optional(self):
        for bi, b in enumerate(self.buckets):
            if b["optional"]:
                self._launch(bi)

    def _make_hook(self, bi: int):
        def hook(_param):
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0 and not b["optional"]:
                self._launch_optional()          # same point of backward on every rank, before the first mandatory bucket
                self._launch(bi)
        return hook

    def zero_grad(self, n_backward: int = 1):
        """Drop ``.grad`` so that autograd hands over freshly produced gradient tensors (no accumulate kernels).
        ``n_backward`` = backward passes that accumulate into this step's gradients (argus ``iter_size``): a bucket is
        reduced when every parameter has been visited that many times."""
        for b in self.buckets:
            b["pending"] = b["count"] * int(n_backward)
            b["expect"] = b["pending"]
            b["launched"] = False
            for p in b["params"]:
                p.grad = None

    def finish(self):
        """Wait for the outstanding all-reduces (call after backward, before the optimizer step); afterwards every
        ``p.grad`` is a view of its bucket and holds the rank-averaged gradient."""
        if self.world > 1:
            self._launch_optional()              # a model without mandatory parameters after the readouts: nothing triggered them
            for bi, b in enumerate(self.buckets):
                if not b["optional"] and not b["launched"] and b["pending"] not in (0, b.get("expect", b["count"])):
                    raise RuntimeError("GradBuckets: a bucket saw only part of its gradients this step (a parameter outside "
                                       "the optional prefixes took no part in the forward pass)")
        for handle, bi, op in self._handles:
            handle.wait()
            if op == dist.ReduceOp.SUM:
                self.buckets[bi]["flat"].div_(self.world)
        reduced = {bi for _, bi, _ in self._handles}
        self._handles.clear()
        if self.world > 1:
            for bi, b in enumerate(self.buckets):
                if bi in reduced:
                    for p, v in self._views(b):
                        p.grad = v

    def num_elements(self) -> int:
        return sum(b["flat"].numel() for b in self.buckets)
