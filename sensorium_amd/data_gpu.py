"""Device-resident trials and on-device batch assembly.

The reference builds every training sample on the CPU — ``np.load`` of four ``.npy`` files, ``StackInputsProcessor``
(src/inputs.py:15-36), ``responses_to_tensor`` (src/responses.py:25-29), ``CutMix`` (src/mixers.py:52-67),
``construct_mice_sample`` (src/datasets.py:172-187) — in 8 DataLoader workers, collates ~200 MB per step (dense targets for
all ten mice, nine of them zeros) and copies it to the GPU (src/argus_models.py:49).  Here the trials live in HBM in their
on-disk layout (the whole Sensorium 2023 training set is a few tens of GB; one MI355X has 288 GB), the host only draws
*which* windows and boxes to use (a few hundred bytes per step, uploaded asynchronously through pinned memory) and two
HIP launches (``dwn_assemble_inputs`` / ``dwn_assemble_targets``, include/dwn.h) write the batch exactly as the
reference's collate would have produced it — same tensors, bit for bit, same ``(input, (targets, mice_weights))``
structure ``MouseModel.train_step`` takes.

Random draws are explicit (a ``numpy.random.RandomState`` handed in by the caller) where the reference reseeds the global
generators from the wall clock per sample (src/utils.py:12-15, src/datasets.py:104-113): same distributions, same draw
order inside one cut-mix decision (``Mixer.use`` -> beta -> randint(width) -> randint(height)), reproducible.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterator, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .predictors import IndexesGenerator


@dataclass
class DeviceTrial:
    video: torch.Tensor                      # (H0, W0, L) uint8 or float32, CUDA
    behavior: torch.Tensor                   # (2, L) float32
    pupil_center: torch.Tensor               # (2, L) float32
    responses: Optional[torch.Tensor]        # (N, L) float32 or None (unlabeled split)
    length: int                              # usable frames (no NaN tail: src/data.py:61-70)
    stride: int                              # L: last-axis extent of the arrays


class DeviceTrialStore:
    """trials[mouse_index] -> list of DeviceTrial, uploaded once."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceTrialStore keeps trials in GPU memory (no CPU path)")
        self.trials: Dict[int, List[DeviceTrial]] = {}

    def add_trial(self, mouse_index: int, video: np.ndarray, behavior: np.ndarray, pupil_center: np.ndarray,
                  responses: Optional[np.ndarray] = None, length: Optional[int] = None, compact: bool = True) -> int:
        if video.ndim != 3:
            raise ValueError("video must be (H, W, L)")
        stride = video.shape[-1]
        for name, arr, rows in (("behavior", behavior, 2), ("pupil_center", pupil_center, 2)):
            if arr.shape != (rows, stride):
                raise ValueError(f"{name} must be ({rows}, {stride}), got {arr.shape}")
        if responses is not None and (responses.ndim != 2 or responses.shape[1] != stride):
            raise ValueError(f"responses must be (N, {stride})")
        length = stride if length is None else int(length)
        if not 0 < length <= stride:
            raise ValueError("length must be in (0, L]")
        if video.dtype != np.uint8:
            v32 = video.astype(np.float32)                   # what the reference feeds the model (inputs.py:26)
            # the dataset stores integral grey levels as floats: keep one byte per pixel when that is lossless
            # (frames past `length` may hold NaN padding — they are never addressed)
            head = v32[..., :length]
            if compact and np.all(np.isfinite(head)) and np.array_equal(head, np.clip(np.rint(head), 0, 255)):
                video = np.nan_to_num(v32, nan=0.0, posinf=0.0, neginf=0.0).clip(0, 255).astype(np.uint8)
            else:
                video = v32
        dev = self.device
        trial = DeviceTrial(
            video=torch.from_numpy(np.ascontiguousarray(video)).to(dev),
            behavior=torch.from_numpy(np.ascontiguousarray(behavior, dtype=np.float32)).to(dev),
            pupil_center=torch.from_numpy(np.ascontiguousarray(pupil_center, dtype=np.float32)).to(dev),
            responses=None if responses is None else
            torch.from_numpy(np.ascontiguousarray(responses, dtype=np.float32)).to(dev),
            length=length, stride=stride)
        self.trials.setdefault(mouse_index, []).append(trial)
        return len(self.trials[mouse_index]) - 1

    def add_mouse_data(self, mouse_index: int, mouse_data: dict):
        """``mouse_data`` as ``get_mouse_data`` returns it (src/data.py:34-73): trials with ``*_path`` to ``.npy`` files."""
        for t in mouse_data["trials"]:
            self.add_trial(mouse_index, np.load(t["video_path"]), np.load(t["behavior_path"]),
                           np.load(t["pupil_center_path"]),
                           np.load(t["response_path"]) if "response_path" in t else None, length=t["length"])

    def nbytes(self) -> int:
        n = 0
        for trials in self.trials.values():
            for t in trials:
                for a in (t.video, t.behavior, t.pupil_center, t.responses):
                    if a is not None:
                        n += a.numel() * a.element_size()
        return n


class ClipPick(NamedTuple):
    mouse: int
    trial: int
    end_frame: int                                        # the window's reference index (position "last": its last frame)
    mix: Optional[Tuple[int, int]] = None                 # (trial, end_frame) of the mixing partner, same mouse
    box: Optional[Tuple[int, int, int, int]] = None       # CutMix: (bbx1, bby1, bbx2, bby2) as rand_bbox returns them
    lam: Optional[float] = None                           # Mixup: the drawn blend factor


def cutmix_box(rng: np.random.RandomState, height: int, width: int, alpha: float, prob: Optional[float] = None
               ) -> Optional[Tuple[int, int, int, int]]:
    """The draws of ``CutMix.__call__`` / ``rand_bbox`` (src/mixers.py:36-49,58-62), in order; with ``prob`` also the
    preceding ``Mixer.use`` (src/mixers.py:15-16)."""
    if prob is not None and not (rng.random_sample() < prob):
        return None
    lam = rng.beta(alpha, alpha)
    cut_rat = np.sqrt(lam)
    cut_w = int(width * cut_rat)
    cut_h = int(height * cut_rat)
    cx = rng.randint(width)
    cy = rng.randint(height)
    clip = lambda v, hi: int(min(max(v, 0), hi))          # noqa: E731
    return clip(cx - cut_w // 2, width), clip(cy - cut_h // 2, height), clip(cx + cut_w // 2, width), \
        clip(cy + cut_h // 2, height)


def parse_mixer(spec) -> Optional[dict]:
    """Normalise a mixer description: ``{"alpha", "prob"}`` (the ``cutmix`` entry of configs/true_batch_001.py:64-67),
    ``("cutmix" | "mixup", {...})`` or ``("random_choice", {"mixers": [...], "choice_probs": [...], "prob": p})``
    — the three classes of src/mixers.py."""
    if spec is None:
        return None
    if isinstance(spec, dict) and "kind" not in spec:
        spec = ("cutmix", spec)
    if isinstance(spec, dict):
        return spec
    kind, params = spec
    params = dict(params)
    if kind == "cutmix":
        return dict(kind="cutmix", alpha=float(params.get("alpha", 1.0)), prob=float(params.get("prob", 1.0)))
    if kind == "mixup":
        return dict(kind="mixup", alpha=float(params.get("alpha", 0.4)), prob=float(params.get("prob", 1.0)))
    if kind == "random_choice":
        subs = [parse_mixer(m) for m in params["mixers"]]
        probs = [float(p) for p in params["choice_probs"]]
        if len(subs) != len(probs) or not subs:
            raise ValueError("random_choice: one probability per mixer")
        return dict(kind="random_choice", mixers=subs, choice_probs=probs, prob=float(params.get("prob", 1.0)))
    raise ValueError(f"unknown mixer {kind!r}")


def mixer_call(rng: np.random.RandomState, mixer: dict, height: int, width: int):
    """The random draws of ``mixer.__call__`` (after ``use`` and after the partner sample was drawn, src/datasets.py:126-128):
    returns (box, lam) — a CutMix box or a Mixup factor."""
    if mixer["kind"] == "cutmix":
        return cutmix_box(rng, height, width, mixer["alpha"]), None
    if mixer["kind"] == "mixup":
        return None, float(rng.beta(mixer["alpha"], mixer["alpha"]))                 # src/mixers.py:30
    index = int(rng.choice(len(mixer["mixers"]), p=mixer["choice_probs"]))            # src/mixers.py:77
    return mixer_call(rng, mixer["mixers"][index], height, width)


class _PinnedRing:
    """Host staging for the per-step descriptor table: pinned so the upload is asynchronous; a few slots so the buffer
    of step k is not rewritten while its copy may still be in flight."""

    def __init__(self, slots: int = 4):
        self.slots: List[Optional[torch.Tensor]] = [None] * slots
        self.events: List[Optional[torch.cuda.Event]] = [None] * slots
        self.i = 0

    def upload(self, raw: bytes, device) -> torch.Tensor:
        k = self.i
        self.i = (self.i + 1) % len(self.slots)
        if self.events[k] is not None:
            self.events[k].synchronize()
        host = self.slots[k]
        if host is None or host.numel() < len(raw):
            host = torch.empty(max(len(raw), 4096), dtype=torch.uint8).pin_memory()
            self.slots[k] = host
        host[:len(raw)].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        dev = torch.empty(len(raw), dtype=torch.uint8, device=device)
        dev.copy_(host[:len(raw)], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[k] = ev
        return dev


class BatchAssembler:
    """Builds ``(input, (targets, mice_weights))`` batches on the device from a ``DeviceTrialStore``.

    ``num_neurons[m]`` sizes the m-th target tensor (``constants.num_neurons``); ``frame_stack`` / ``size`` /
    ``pad_fill_value`` / ``cutmix`` are the config entries of configs/true_batch_001.py:49-68; ``mixer`` (instead of
    ``cutmix``) takes any of the reference's three mixers, see ``parse_mixer``.
    """

    def __init__(self, store: DeviceTrialStore, num_neurons: Sequence[int], frame_stack: dict,
                 size: Tuple[int, int], pad_fill_value: float = 0.0, cutmix: Optional[dict] = None, mixer=None):
        self.store = store
        self.num_neurons = [int(n) for n in num_neurons]
        self.gen = IndexesGenerator(**frame_stack)
        self.window = int(frame_stack["size"])
        self.size = (int(size[0]), int(size[1]))                       # (W, H) like the reference
        self.pad_fill_value = float(pad_fill_value)
        if cutmix and mixer is not None:
            raise ValueError("give either cutmix or mixer")
        self.mixer = parse_mixer(mixer if mixer is not None else (cutmix or None))
        self._ring = _PinnedRing()
        self._n_dev = torch.tensor(self.num_neurons, dtype=torch.int32, device=store.device)

    # -- which clips -------------------------------------------------------------------------------------------
    def _draw_clip(self, rng: np.random.RandomState, mouse: int) -> Tuple[int, int]:
        """TrainMouseVideoDataset.get_indexes (src/datasets.py:104-113): uniform trial, uniform admissible frame."""
        trials = self.store.trials[mouse]
        trial = int(rng.randint(0, len(trials)))
        frame = int(rng.randint(self.gen.behind, trials[trial].length - self.gen.ahead))
        return trial, frame

    def draw_train_picks(self, rng: np.random.RandomState, mice: Sequence[int]) -> List[ClipPick]:
        """One pick per entry of ``mice`` (TrainMouseVideoDataset.__getitem__, src/datasets.py:121-129)."""
        w, h = self.size
        picks = []
        for mouse in mice:
            trial, frame = self._draw_clip(rng, mouse)
            mix = box = lam = None
            # Mixer.use, then the partner sample, then the mixer's own draws (src/datasets.py:126-128)
            if self.mixer is not None and rng.random_sample() < self.mixer["prob"]:
                mix = self._draw_clip(rng, mouse)
                box, lam = mixer_call(rng, self.mixer, h, w)
            picks.append(ClipPick(int(mouse), trial, frame, mix, box, lam))
        return picks

    def val_picks(self, mouse: int) -> List[ClipPick]:
        """ValMouseVideoDataset (src/datasets.py:132-163): consecutive non-overlapping windows of every trial."""
        width = self.gen.width
        picks = []
        for ti, t in enumerate(self.store.trials[mouse]):
            for k in range(t.length // width):
                picks.append(ClipPick(mouse, ti, self.gen.behind + k * width))
        return picks

    # -- the batch ------------------------------------------------------------------------------------------------
    def _fill_src(self, dst: L.ClipSrc, mouse: int, trial: int, end_frame: int, need_responses: bool, h0w0):
        t = self.store.trials[mouse][trial]
        start = end_frame - self.gen.behind
        last = start + (self.window - 1) * self.gen.step
        if start < 0 or last >= t.stride:
            raise IndexError(f"window [{start}, {last}] outside trial of {t.stride} frames")
        if tuple(t.video.shape[:2]) != h0w0:
            raise ValueError("all videos of a batch must share one frame size")
        if need_responses and t.responses is None:
            raise ValueError("trial has no responses (unlabeled split)")
        dst.video = t.video.data_ptr()
        dst.behavior = t.behavior.data_ptr()
        dst.pupil_center = t.pupil_center.data_ptr()
        dst.responses = 0 if t.responses is None else t.responses.data_ptr()
        dst.length = t.stride
        dst.video_dtype = L.VID_U8 if t.video.dtype == torch.uint8 else L.VID_F32
        dst.frame_start, dst.frame_step, dst.valid = start, self.gen.step, 1
        return t

    def assemble(self, picks: Sequence[ClipPick], with_targets: bool = True):
        dev = self.store.device
        B, T = len(picks), self.window
        if B == 0:
            raise ValueError("empty batch")
        W, H = self.size
        first = self.store.trials[picks[0].mouse][picks[0].trial]
        h0w0 = tuple(first.video.shape[:2])
        descs = (L.ClipDesc * B)()
        for d, p in zip(descs, picks):
            t = self._fill_src(d.src, p.mouse, p.trial, p.end_frame, with_targets, h0w0)
            if with_targets and t.responses.shape[0] != self.num_neurons[p.mouse]:
                raise ValueError(f"mouse {p.mouse}: trial has {t.responses.shape[0]} neurons, expected "
                                 f"{self.num_neurons[p.mouse]}")
            d.mouse = p.mouse
            d.one_minus_lam, d.lam, d.mix_mode = 1.0, 0.0, L.MIX_BOX
            if p.box is not None:
                self._fill_src(d.mix, p.mouse, p.mix[0], p.mix[1], with_targets, h0w0)
                d.bbx1, d.bby1, d.bbx2, d.bby2 = (int(v) for v in p.box)
                lam = (d.bbx2 - d.bbx1) * (d.bby2 - d.bby1) / (H * W)          # python float, like mixers.py:64
                d.one_minus_lam, d.lam = float(np.float32(1 - lam)), float(np.float32(lam))
            elif p.lam is not None:                                                # Mixup (src/mixers.py:30-32)
                self._fill_src(d.mix, p.mouse, p.mix[0], p.mix[1], with_targets, h0w0)
                d.mix_mode = L.MIX_BLEND
                d.one_minus_lam, d.lam = float(np.float32(1 - p.lam)), float(np.float32(p.lam))
        x = torch.empty(B, 5, T, H, W, dtype=torch.float32, device=dev)
        targets = weights = None
        raw = bytes(descs)
        off_ptr = len(raw)
        if with_targets:
            targets = [torch.empty(B, n, T, dtype=torch.float32, device=dev) for n in self.num_neurons]
            weights = torch.empty(B, len(self.num_neurons), dtype=torch.float32, device=dev)
            raw += np.array([t.data_ptr() for t in targets], dtype=np.uint64).tobytes()
        table = self._ring.upload(raw, dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        L.check(L.lib.dwn_assemble_inputs(table.data_ptr(), B, T, h0w0[0], h0w0[1], H, W, self.pad_fill_value,
                                          x.data_ptr(), dev.index, stream), "dwn_assemble_inputs")
        if with_targets:
            L.check(L.lib.dwn_assemble_targets(table.data_ptr(), B, T, table.data_ptr() + off_ptr,
                                               self._n_dev.data_ptr(), len(self.num_neurons), max(self.num_neurons),
                                               weights.data_ptr(), dev.index, stream), "dwn_assemble_targets")
            # the host knows the owner of every sample (the kernel writes exactly these one-hot rows): MouseModel.train_step
            # uses the copy to run each readout's backward on its own samples only, without reading the device tensor back
            host = torch.zeros(B, len(self.num_neurons), dtype=torch.float32)
            host[torch.arange(B), torch.tensor([p.mouse for p in picks])] = 1.0
            weights._dwn_host = host
            weights._dwn_host_version = weights._version      # an in-place edit afterwards invalidates the host copy
            return x, (targets, weights)
        return x


class DeviceBatchLoader:
    """Iterable the ``fit`` loop consumes in place of ``DataLoader(ConcatMiceVideoDataset(...), shuffle=True)``
    (scripts/train.py:74-105): ``epoch_size`` samples per epoch, ``epoch_size // n_mice`` per mouse, shuffled across
    mice, cut in batches of ``batch_size`` (last batch may be short, like DataLoader's default)."""

    def __init__(self, assembler: BatchAssembler, batch_size: int, epoch_size: int, seed: int = 0,
                 mice: Optional[Sequence[int]] = None):
        self.assembler = assembler
        self.batch_size = int(batch_size)
        self.mice = list(mice) if mice is not None else sorted(assembler.store.trials)
        self.per_mouse = int(epoch_size) // len(self.mice)
        self.rng = np.random.RandomState(seed)

    def __len__(self) -> int:
        n = self.per_mouse * len(self.mice)
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator:
        order = np.repeat(np.array(self.mice, dtype=np.int64), self.per_mouse)
        self.rng.shuffle(order)
        for i in range(0, len(order), self.batch_size):
            yield self.assembler.assemble(self.assembler.draw_train_picks(self.rng, order[i:i + self.batch_size]))


class DeviceValLoader:
    """``DataLoader(ConcatMiceVideoDataset([ValMouseVideoDataset...]), shuffle=False)`` (scripts/train.py:84-111)."""

    def __init__(self, assembler: BatchAssembler, batch_size: int, mice: Optional[Sequence[int]] = None):
        self.assembler = assembler
        self.batch_size = int(batch_size)
        mice = list(mice) if mice is not None else sorted(assembler.store.trials)
        self.picks = [p for m in mice for p in assembler.val_picks(m)]

    def __len__(self) -> int:
        return (len(self.picks) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator:
        for i in range(0, len(self.picks), self.batch_size):
            yield self.assembler.assemble(self.picks[i:i + self.batch_size])
