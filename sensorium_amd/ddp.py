"""Data-parallel gradient exchange for the DwiseNeuro step: one process per GPU, flat fp32 gradient buckets,
RCCL all-reduce over xGMI launched from autograd hooks so it overlaps with the rest of backward.

The reference has no distributed code; this is the one exchange step data parallelism adds:
an all-reduce (mean) of the parameter gradients per iteration.  Buckets are built in *reverse* registration
order — readouts (≈16 M params each, ready first in backward) lead, then the cortex and the ~200 small core tensors
in ≈12 MB buckets — so the big transfers are issued while the core backward (≈95 % of the step) is still running
and only the last small bucket (first blocks + stem) is exposed after it.
BatchNorm statistics stay local to each rank (standard DDP semantics).

No gather pass: every parameter owns a slice of its bucket (``param._dwn_grad_slot``) and the HIP backward writes the
gradient *there* (``ops.grad_out``); autograd then adopts that tensor as ``param.grad``, so when the last gradient of a
bucket has arrived the flat buffer is already complete and the collective starts on it as it is.  (A gradient autograd
had to materialise elsewhere — accumulation into an existing ``.grad``, a parameter used twice — is copied into its slice
by the hook; results are the same.)

``comm_dtype=torch.bfloat16`` exchanges bf16 copies of the buckets (half the bytes on xGMI; the fp32 bucket is the master
copy: rounded once before the exchange, the averaged values are widened back on arrival).

Parameters that a step may legitimately leave without a gradient — the per-mouse readouts when the model is called as
``forward(x, index)`` (dwiseneuro.py:404-405) — live in *optional* buckets: those are not launched by their own hooks
(ranks training different mice would issue the collectives in different orders) but all together, in bucket order, at the
moment the first mandatory bucket (the cortex: always used, and complete only after every readout's backward) is launched —
the same point of the backward pass on every rank.  A locally unused parameter contributes zeros plus a "not used" flag
that rides in the bucket's tail: a parameter no rank used keeps ``grad = None`` (exactly what a single process sees, so
the optimizer skips it on every rank alike); one that another rank used receives that rank's gradient / world.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, module: torch.nn.Module, bucket_cap_mb: float = 12.0, process_group=None,
                 broadcast_init: bool = True, optional_prefixes=("readouts.",), comm_dtype: Optional[torch.dtype] = None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_dtype = None if comm_dtype in (None, torch.float32) else comm_dtype
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        params = [p for _, p in named]
        optional = {id(p) for n, p in named if any(n.startswith(pre) for pre in optional_prefixes)}
        if self.world > 1 and broadcast_init:
            self._broadcast(list(module.parameters()) + list(module.buffers()))
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
                for p, off in zip(b["params"], b["offsets"]):
                    p._dwn_grad_slot = (b["flat"], off)       # ops.grad_out: the HIP backward writes the gradient here
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _broadcast(self, tensors):
        """rank 0's parameters and buffers to every rank: one collective per dtype (flattened) instead of one per tensor"""
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t.data)
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=0, group=self.pg)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    def _add_bucket(self, params, optional: bool = False):
        # every slice starts on a 16-byte boundary (the backward kernels clear / write their outputs with 16-byte stores)
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        n = off
        # optional buckets carry one "used by this rank" flag per parameter behind the gradients
        flat = torch.zeros(n + (len(params) if optional else 0), dtype=params[0].dtype, device=params[0].device)
        views = [flat[o:o + p.numel()].view_as(p) for p, o in zip(params, offsets)]
        comm = torch.zeros_like(flat, dtype=self.comm_dtype) if self.comm_dtype is not None else None
        self.buckets.append(dict(params=params, flat=flat, offsets=offsets, views=views, numel=n, comm=comm,
                                 true_numel=sum(p.numel() for p in params),
                                 index={id(p): i for i, p in enumerate(params)}, arrived=[False] * len(params),
                                 flags_sent=None, pending=len(params), count=len(params), optional=optional, launched=False))

    def _views(self, b):
        return zip(b["params"], b["views"])

    def _launch(self, bi: int):
        b = self.buckets[bi]
        if b["launched"]:
            return
        b["launched"] = True
        if b["optional"]:
            # parameters that took no part in this rank's step: zeros, and a 0 in the flag tail
            for i, ok in enumerate(b["arrived"]):
                if not ok:
                    b["views"][i].zero_()
            flags = tuple(1.0 if ok else 0.0 for ok in b["arrived"])
            if flags != b["flags_sent"]:                 # constant in ordinary training: uploaded once, through pinned memory
                host = torch.tensor(flags, dtype=b["flat"].dtype)
                if b["flat"].is_cuda:
                    host = host.pin_memory()
                    b["_pinned"] = host                  # kept alive until the next upload: the async copy reads it
                if b.get("flags_dev") is None:
                    b["flags_dev"] = torch.empty(b["count"], dtype=b["flat"].dtype, device=b["flat"].device)
                b["flags_dev"].copy_(host, non_blocking=True)
                b["flags_sent"] = flags
            b["flat"][b["numel"]:].copy_(b["flags_dev"])   # the previous step's reduced tail is overwritten
        if b["flat"].is_cuda:
            from .ops import side_join
            side_join(b["flat"].device)              # gradients still being written on the backward's side stream
        buf = b["flat"]
        if b["comm"] is not None:
            b["comm"].copy_(b["flat"])                   # one rounding to the exchange type
            buf = b["comm"]
        op = dist.ReduceOp.AVG if dist.get_backend(self.pg) == "nccl" else dist.ReduceOp.SUM
        self._handles.append((dist.all_reduce(buf, op=op, group=self.pg, async_op=True), bi, op))

    def _launch_optional(self):
        for bi, b in enumerate(self.buckets):
            if b["optional"]:
                self._launch(bi)

    def _make_hook(self, bi: int):
        def hook(param):
            b = self.buckets[bi]
            i = b["index"][id(param)]
            view = b["views"][i]
            g = param.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                # autograd produced / accumulated this gradient in a tensor of its own: move it into the bucket
                view.copy_(g)
                param.grad = view
            b["arrived"][i] = True
            b["pending"] -= 1
            if b["pending"] == 0 and not b["optional"]:
                self._launch_optional()          # same point of backward on every rank, before the first mandatory bucket
                self._launch(bi)
        return hook

    def zero_grad(self, n_backward: int = 1):
        """Drop ``.grad`` so that the next backward writes straight into the buckets (no accumulate kernels).
        ``n_backward`` = backward passes that accumulate into this step's gradients (argus ``iter_size``): a bucket is
        reduced when every parameter has been visited that many times."""
        for b in self.buckets:
            b["pending"] = b["count"] * int(n_backward)
            b["expect"] = b["pending"]
            b["launched"] = False
            b["arrived"] = [False] * b["count"]
            for p in b["params"]:
                p.grad = None

    def finish(self):
        """Wait for the outstanding all-reduces (call after backward, before the optimizer step); afterwards every
        ``p.grad`` is a view of its bucket and holds the rank-averaged gradient (``None`` for an optional parameter no rank
        used)."""
        if self.world > 1:
            self._launch_optional()              # a model without mandatory parameters after the readouts: nothing triggered them
            for bi, b in enumerate(self.buckets):
                if not b["optional"] and not b["launched"] and b["pending"] not in (0, b.get("expect", b["count"])):
                    raise RuntimeError("GradBuckets: a bucket saw only part of its gradients this step (a parameter outside "
                                       "the optional prefixes took no part in the forward pass)")
        for handle, bi, op in self._handles:
            handle.wait()
            b = self.buckets[bi]
            if b["comm"] is not None:
                b["flat"].copy_(b["comm"])
            if op == dist.ReduceOp.SUM:
                b["flat"].div_(self.world)
        reduced = {bi for _, bi, _ in self._handles}
        self._handles.clear()
        if self.world > 1:
            for bi, b in enumerate(self.buckets):
                if bi not in reduced:
                    continue
                used = None
                if b["optional"] and not all(b["arrived"]):
                    # only a rank that skipped a parameter has to ask whether somebody else used it (one small readback)
                    used = (b["flat"][b["numel"]:] > 0).tolist()
                for i, (p, v) in enumerate(self._views(b)):
                    if used is not None and not b["arrived"][i] and not used[i]:
                        p.grad = None
                    else:
                        p.grad = v

    def num_elements(self) -> int:
        return sum(b["true_numel"] for b in self.buckets)

    def bytes_on_wire_per_step(self) -> int:
        """Bytes one rank sends (= receives) per step for the ring all-reduce of every bucket: 2 (N-1)/N x bucket bytes."""
        esize = 4 if self.comm_dtype is None else torch.empty((), dtype=self.comm_dtype).element_size()
        total = sum(b["flat"].numel() for b in self.buckets) * esize
        return int(2 * (self.world - 1) / max(self.world, 1) * total)
