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

``shard_optional=True`` (the ten-readout model: 161 M readout parameters = 646 MB of fp32 gradients, 1.9 GB of
Adam moments + EMA copies and a 5.8 GB optimizer pass on every rank): the optional buckets — one per readout module — are
*reduce-scattered* instead of all-reduced, each rank runs the fused AdamW/EMA kernel on the 1/N slice of the bucket it owns
(``owned_range``; moments exist for that slice only) and the updated parameters travel back with an *all-gather* straight
into the parameter storage (the readout parameters are re-pointed into one flat buffer per bucket), launched after the
optimizer step and awaited by a forward pre-hook on the readout — the last module of the network — so it is hidden behind
the next step's whole core forward.  Bytes on xGMI are those of the all-reduce (a ring all-reduce IS reduce-scatter +
all-gather); the optimizer's HBM traffic, its state and the EMA copies of the readouts shrink by the world size.  The EMA
copies of a sharded bucket are current for the owned slice only until ``gather_ema`` (validation / prediction / checkpoint
time) brings the slices together.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


import weakref

def init_rccl(device: torch.device, **kwargs) -> None:
    """``dist.init_process_group("nccl", device_id=device)`` with the RCCL kernels on a HIGH-PRIORITY stream.

    Not a tuning nicety: HIP maps streams onto a few hardware queues (``GPU_MAX_HW_QUEUES``, 4 by default) and the stream
    ProcessGroupNCCL takes from torch's normal-priority pool landed on the SAME hardware queue as the compute stream — every
    collective then ran strictly between two compute kernels (rocprofv3, one rank over RCCL on an MI355X: 0 us of the 156 us
    of collective kernels overlapped, ~10 us of idle on either side of each), i.e. the gradient exchange would not overlap
    with backward at all.  On the high-priority pool's stream (or with ``GPU_MAX_HW_QUEUES=8`` in the environment before the
    first HIP call — bench.py sets both) it gets a queue of its own and overlaps completely (tools/rccl_queue_probe.sh).
    Call before anything else creates the default process group."""
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", device_id=device, pg_options=opts, **kwargs)


_LIVE = weakref.WeakSet()          # GradBuckets with an exchange to run (flush_ready_all)


def flush_ready_all() -> None:
    """Called by the backward of a core block once its kernels are queued (sensorium_amd/ops.py): start the exchange of every
    bucket that has become complete.  See ``GradBuckets._flush_ready`` for why not earlier."""
    for gb in list(_LIVE):
        if gb._ready:
            gb._flush_ready()


class GradBuckets:
    def __init__(self, module: torch.nn.Module, bucket_cap_mb: float = 12.0, process_group=None,
                 broadcast_init: bool = True, optional_prefixes=("readouts.",), comm_dtype: Optional[torch.dtype] = None,
                 shard_optional: bool = False, single_rank: bool = False):
        """``single_rank``: run the whole exchange machinery (hooks, collectives in place, sharded optimizer) on a process
        group of ONE rank too — every collective is then the identity, but it goes through the backend: this is how the RCCL
        code path is exercised on a one-GPU box (tests/test_gpu_ddp.py)."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.active = self.world > 1 or (bool(single_rank) and dist.is_initialized())
        self.shard = bool(shard_optional) and self.active
        self.module = module
        self.comm_dtype = None if comm_dtype in (None, torch.float32) else comm_dtype
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        params = [p for _, p in named]
        optional = {id(p) for n, p in named if any(n.startswith(pre) for pre in optional_prefixes)}
        name_of = {id(p): n for n, p in named}

        def owner_module(p):          # "readouts.3.layer.1.weight" -> "readouts.3": sharded buckets never span two readouts
            return ".".join(name_of[id(p)].split(".")[:2])
        if self.active and broadcast_init:
            self._broadcast(list(module.parameters()) + list(module.buffers()))
        self.buckets: List[dict] = []
        cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        cur: List[torch.nn.Parameter] = []
        cur_n = 0
        for p in reversed(params):
            kind_changes = bool(cur) and ((id(p) in optional) != (id(cur[0]) in optional))
            if cur and self.shard and id(p) in optional and id(cur[0]) in optional:
                split = owner_module(p) != owner_module(cur[0])             # one sharded bucket per readout, whatever its size
            else:
                split = cur_n + p.numel() > cap and cur_n * 8 > cap         # tiny leftovers ride with the next tensor
            if cur and (kind_changes or split):
                self._add_bucket(cur, id(cur[0]) in optional, owner_module(cur[0]))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._add_bucket(cur, id(cur[0]) in optional, owner_module(cur[0]))
        self._handles: List = []
        self._ready: List = []                     # complete buckets whose collective has not been started yet (_flush_ready)
        self._param_handles: List = []
        self._flag_handle = None
        self._hooks = []
        self._module_hooks = []
        self._owned = {}
        self.ema_dirty = False
        for b in self.buckets:
            if b["sharded"]:
                lo, hi = b["shard"]
                for p, off in zip(b["params"], b["offsets"]):
                    a, z = max(lo, off) - off, min(hi, off + p.numel()) - off
                    self._owned[id(p)] = (a, z) if z > a else (0, 0)
                sub = module.get_submodule(b["owner"])
                self._module_hooks.append(sub.register_forward_pre_hook(lambda m, args: self.wait_params()))
        if self.active:
            _LIVE.add(self)
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

    def _add_bucket(self, params, optional: bool = False, owner: str = ""):
        # every slice starts on a 16-byte boundary (the backward kernels clear / write their outputs with 16-byte stores)
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        n = off
        sharded = optional and self.shard
        shard, pflat = None, None
        if sharded:
            # equal 16-byte aligned slices, one per rank; the "used" flags travel in a collective of their own
            per = (n + 4 * self.world - 1) // (4 * self.world) * 4
            n = per * self.world
            shard = (self.rank * per, (self.rank + 1) * per)
            flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
            pflat = torch.zeros_like(flat)             # the parameters themselves, contiguous: all-gather target
            for p, o in zip(params, offsets):
                dst = pflat[o:o + p.numel()].view_as(p)
                dst.copy_(p.data)
                p.data = dst
        else:
            # optional buckets carry one "used by this rank" flag per parameter behind the gradients
            flat = torch.zeros(n + (len(params) if optional else 0), dtype=params[0].dtype, device=params[0].device)
        views = [flat[o:o + p.numel()].view_as(p) for p, o in zip(params, offsets)]
        comm = torch.zeros_like(flat, dtype=self.comm_dtype) if self.comm_dtype is not None else None
        self.buckets.append(dict(params=params, flat=flat, offsets=offsets, views=views, numel=n, comm=comm,
                                 true_numel=sum(p.numel() for p in params),
                                 index={id(p): i for i, p in enumerate(params)}, arrived=[False] * len(params),
                                 flags_sent=None, pending=len(params), count=len(params), optional=optional, launched=False,
                                 sharded=sharded, shard=shard, pflat=pflat, eflat=None, owner=owner, used=None))

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
        if b["optional"] and not b["sharded"]:
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
        buf = b["flat"]
        if b["comm"] is not None:
            b["comm"].copy_(b["flat"])                   # one rounding to the exchange type
            buf = b["comm"]
        op = dist.ReduceOp.AVG if dist.get_backend(self.pg) == "nccl" else dist.ReduceOp.SUM
        if b["sharded"]:
            lo, hi = b["shard"]                          # in place: this rank's slice of the bucket receives the sum
            h = dist.reduce_scatter_tensor(buf[lo:hi], buf, op=op, group=self.pg, async_op=True)
        else:
            h = dist.all_reduce(buf, op=op, group=self.pg, async_op=True)
        self._handles.append((h, bi, op))

    def _launch_optional(self):
        sharded = []
        for bi, b in enumerate(self.buckets):
            if b["optional"]:
                if b["sharded"] and not b["launched"]:
                    sharded.append(b)
                self._launch(bi)
        if sharded:
            # which parameters of the sharded buckets did ANY rank use: one small all-reduce for all of them
            flags = [1.0 if ok else 0.0 for b in sharded for ok in b["arrived"]]
            key = tuple(flags)
            if getattr(self, "_flags_key", None) != key:
                host = torch.tensor(flags, dtype=torch.float32)
                dev = sharded[0]["flat"].device
                if dev.type == "cuda":
                    host = host.pin_memory()
                    self._flags_pinned = host
                self._flags_src = torch.empty(len(flags), dtype=torch.float32, device=dev)
                self._flags_src.copy_(host, non_blocking=True)
                self._flags_key = key
            self._flags_buf = self._flags_src.clone()
            self._flag_handle = (dist.all_reduce(self._flags_buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), sharded)

    def _flush_ready(self):
        """Start the exchange of the buckets that became complete earlier.  A collective costs the host 50-100 us
        (ProcessGroup bookkeeping).  The head of the network (readout, cortex: short kernels) leaves the GPU waiting for the
        host as it is, so the two big early buckets' collectives issued from their hooks were ~150 us of GPU idle time per
        step (rocprofv3, one rank over RCCL).  They are started instead when the next core block's backward has just been
        queued (``flush_ready_all`` from ops.BlockFn.backward: ~1.5 ms of kernels ahead of the host), and at ``finish()``."""
        ready, self._ready = self._ready, []
        for kind, bi in ready:
            if kind == "optional":
                self._launch_optional()
            else:
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
            param._dwn_slot_out = False          # ops.grad_out hands the slot out once per backward pass
            b["arrived"][i] = True
            b["pending"] -= 1
            if b["pending"] == 0 and not b["optional"]:
                # same point of backward on every rank: the optional buckets go before the first mandatory one
                self._ready.append(("optional", bi))
                self._ready.append(("bucket", bi))
        return hook

    def zero_grad(self, n_backward: int = 1):
        """Drop ``.grad`` so that the next backward writes straight into the buckets (no accumulate kernels).
        ``n_backward`` = backward passes that accumulate into this step's gradients (argus ``iter_size``): a bucket is
        reduced when every parameter has been visited that many times."""
        self._ready = []
        for b in self.buckets:
            b["pending"] = b["count"] * int(n_backward)
            b["expect"] = b["pending"]
            b["launched"] = False
            b["arrived"] = [False] * b["count"]
            for p in b["params"]:
                p.grad = None
                p._dwn_slot_out = False

    def finish(self):
        """Wait for the outstanding all-reduces (call after backward, before the optimizer step); afterwards every
        ``p.grad`` is a view of its bucket and holds the rank-averaged gradient (``None`` for an optional parameter no rank
        used)."""
        self._flush_ready()
        if self.active:
            self._launch_optional()              # a model without mandatory parameters after the readouts: nothing triggered them
            for bi, b in enumerate(self.buckets):
                if not b["optional"] and not b["launched"] and b["pending"] not in (0, b.get("expect", b["count"])):
                    raise RuntimeError("GradBuckets: a bucket saw only part of its gradients this step (a parameter outside "
                                       "the optional prefixes took no part in the forward pass)")
        for handle, bi, op in self._handles:
            handle.wait()
            b = self.buckets[bi]
            tgt = b["flat"]
            if b["sharded"]:
                lo, hi = b["shard"]
                tgt = tgt[lo:hi]                         # only the owned slice holds reduced values
            if b["comm"] is not None:
                tgt.copy_(b["comm"][lo:hi] if b["sharded"] else b["comm"])
            if op == dist.ReduceOp.SUM:
                tgt.div_(self.world)
        reduced = {bi for _, bi, _ in self._handles}
        self._handles.clear()
        if self._flag_handle is not None:
            h, sharded = self._flag_handle
            h.wait()
            self._flag_handle = None
            if not all(ok for b in sharded for ok in b["arrived"]):       # somebody here skipped a parameter: read the verdict
                used = (self._flags_buf > 0).tolist()
            else:
                used = [True] * sum(b["count"] for b in sharded)
            k = 0
            for b in sharded:
                b["used"] = used[k:k + b["count"]]
                k += b["count"]
        if self.active:
            for bi, b in enumerate(self.buckets):
                if bi not in reduced:
                    continue
                if b["sharded"]:
                    # p.grad is the bucket view as always, but only ``owned_range(p)`` of it holds the averaged gradient
                    for i, (p, v) in enumerate(self._views(b)):
                        p.grad = v if b["used"][i] else None
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

    # ---- sharded optimizer support (shard_optional=True) -------------------------------------------------
    def owned_range(self, p):
        """``None``: the optimizer updates all of ``p``.  ``(lo, hi)``: element range of ``p`` (flattened) this rank owns and
        updates — possibly empty — the rest arrives by all-gather (``gather_params``)."""
        return self._owned.get(id(p))

    def gather_params(self):
        """After the optimizer step: every rank's updated slice to everybody, straight into the parameter storage."""
        for b in self.buckets:
            if b["sharded"]:
                lo, hi = b["shard"]
                self._param_handles.append(dist.all_gather_into_tensor(b["pflat"], b["pflat"][lo:hi], group=self.pg,
                                                                       async_op=True))
        self.ema_dirty = True

    def wait_params(self):
        for h in self._param_handles:
            h.wait()
        self._param_handles.clear()

    def adopt_ema(self, ema_module: torch.nn.Module):
        """Re-point the EMA copies of the sharded parameters into one flat buffer per bucket (same layout as the
        parameters) so that ``gather_ema`` is one in-place all-gather per bucket."""
        if not self.shard:
            return
        by_name = dict(ema_module.named_parameters())
        name_of = {id(p): n for n, p in self.module.named_parameters()}
        for b in self.buckets:
            if not b["sharded"]:
                continue
            b["eflat"] = torch.zeros_like(b["pflat"])
            for p, o in zip(b["params"], b["offsets"]):
                e = by_name[name_of[id(p)]]
                dst = b["eflat"][o:o + p.numel()].view_as(e)
                dst.copy_(e.data)
                e.data = dst
        self.ema_dirty = False

    def gather_ema(self):
        """The EMA copies of sharded parameters are current for the owned slice only; bring them together (before the EMA
        network is evaluated or saved)."""
        if not self.ema_dirty:
            return
        self.wait_params()
        for b in self.buckets:
            if b["sharded"] and b["eflat"] is not None:
                lo, hi = b["shard"]
                dist.all_gather_into_tensor(b["eflat"], b["eflat"][lo:hi], group=self.pg)
        self.ema_dirty = False

    def num_elements(self) -> int:
        return sum(b["true_numel"] for b in self.buckets)

    def bytes_on_wire_per_step(self) -> int:
        """Bytes one rank sends (= receives) per step for the ring all-reduce of every bucket: 2 (N-1)/N x bucket bytes."""
        esize = 4 if self.comm_dtype is None else torch.empty((), dtype=self.comm_dtype).element_size()
        ring = (self.world - 1) / max(self.world, 1)
        total = 0.0
        for b in self.buckets:
            n = b["flat"].numel()
            if b["sharded"]:                 # reduce-scatter of the gradients (exchange type) + all-gather of fp32 parameters
                total += ring * n * esize + ring * n * 4
            else:
                total += 2 * ring * n * esize
        return int(total)
