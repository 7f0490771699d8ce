"""CPU restatement (numpy) of the reference's per-sample batch assembly — TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this file; the product path
(``sensorium_amd/data_gpu.py`` -> ``dwn_assemble_inputs`` / ``dwn_assemble_targets``) never does.

Follows, function by function:
  * ``window_indexes``        src/indexes.py:23-30      frame indexes of one window
  * ``stack_inputs``          src/inputs.py:15-36       centre-pad the (H0,W0,T) video into (5,T,H,W); behaviour / pupil
                                                         scalars broadcast over the frame
  * ``responses_to_target``   src/responses.py:25-29    float32 + relu
  * ``cutmix_draw``           src/mixers.py:36-49,58-62 the random draws of ``Mixer.use`` + ``CutMix.__call__`` in the
                                                         reference's order (random, beta, randint(width), randint(height))
  * ``cutmix_apply``          src/mixers.py:52-67       note the reference pastes rows ``bbx1:bbx2`` (the "x" range goes
                                                         to the second-to-last axis) and columns ``bby1:bby2``; lam is
                                                         recomputed from the clipped box
  * ``mice_sample``           src/datasets.py:172-187   per-mouse target list with zeros for the other mice + one-hot weights
  * ``collate``               torch default_collate of those samples (stack along a new batch axis)

Pinned by tests/golden/data_pipeline.npz, generated from the reference's own ``StackInputsProcessor`` and ``CutMix``
(oracle/make_golden_data.py).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def window_indexes(index: int, size: int, step: int, position: str = "last") -> List[int]:
    if position == "first":
        behind, ahead = 0, size - 1
    elif position == "middle":
        behind = size // 2
        ahead = size - behind - 1
    elif position == "last":
        behind, ahead = size - 1, 0
    else:
        raise ValueError(position)
    return list(range(index - behind * step, index + ahead * step + 1, step))


def stack_inputs(frames: np.ndarray, behavior: np.ndarray, pupil_center: np.ndarray, size: Tuple[int, int],
                 pad_fill_value: float = 0.0) -> np.ndarray:
    """frames (H0, W0, T) any real dtype, behavior (2, T), pupil_center (2, T); size = (W, H) -> (5, T, H, W) float32."""
    length = frames.shape[-1]
    out = np.full((5, length, size[1], size[0]), pad_fill_value, dtype=np.float32)
    fr = np.transpose(frames.astype(np.float32), (2, 0, 1))
    h0, w0 = fr.shape[-2:]
    hs, ws = (size[1] - h0) // 2, (size[0] - w0) // 2
    out[0, :, hs:hs + h0, ws:ws + w0] = fr
    out[1:3] = behavior[:, :, None, None]
    out[3:] = pupil_center[:, :, None, None]
    return out


def responses_to_target(responses: np.ndarray) -> np.ndarray:
    return np.maximum(responses.astype(np.float32), 0.0)


def cutmix_draw(rng: np.random.RandomState, height: int, width: int, alpha: float, prob: Optional[float]
                ) -> Optional[Tuple[int, int, int, int]]:
    """Returns None when the mixer is not used, else (bbx1, bby1, bbx2, bby2) exactly as ``rand_bbox`` clips them.
    ``prob=None``: only ``CutMix.__call__``'s own draws (the mixer was chosen by ``RandomChoiceMixer``, src/mixers.py:77-79)."""
    if prob is not None and not (rng.random_sample() < prob):
        return None
    lam = rng.beta(alpha, alpha)
    cut_rat = np.sqrt(lam)
    cut_w = (width * cut_rat).astype(int)
    cut_h = (height * cut_rat).astype(int)
    cx = rng.randint(width)
    cy = rng.randint(height)
    return (int(np.clip(cx - cut_w // 2, 0, width)), int(np.clip(cy - cut_h // 2, 0, height)),
            int(np.clip(cx + cut_w // 2, 0, width)), int(np.clip(cy + cut_h // 2, 0, height)))


def cutmix_apply(inputs1: np.ndarray, target1: np.ndarray, inputs2: np.ndarray, target2: np.ndarray,
                 box: Tuple[int, int, int, int]) -> Tuple[np.ndarray, np.ndarray]:
    bbx1, bby1, bbx2, bby2 = box
    h, w = inputs1.shape[-2:]
    inputs = inputs1.copy()
    inputs[..., bbx1:bbx2, bby1:bby2] = inputs2[..., bbx1:bbx2, bby1:bby2]
    lam = (bbx2 - bbx1) * (bby2 - bby1) / (h * w)
    target = ((1 - lam) * target1 + lam * target2).astype(np.float32)
    return inputs, target


def mixup_draw(rng: np.random.RandomState, alpha: float, prob: float) -> Optional[float]:
    """``Mixer.use`` + ``Mixup.__call__``'s draw (src/mixers.py:15-16,30): None when unused, else lam."""
    if not (rng.random_sample() < prob):
        return None
    return float(rng.beta(alpha, alpha))


def mixup_apply(inputs1: np.ndarray, target1: np.ndarray, inputs2: np.ndarray, target2: np.ndarray, lam: float
                ) -> Tuple[np.ndarray, np.ndarray]:
    """src/mixers.py:31-32 on float32 tensors: torch multiplies by the scalar cast to float32; mul, mul, add rounded
    separately."""
    a, b = np.float32(1 - lam), np.float32(lam)
    return (a * inputs1.astype(np.float32) + b * inputs2.astype(np.float32)).astype(np.float32), \
        (a * target1.astype(np.float32) + b * target2.astype(np.float32)).astype(np.float32)


def mice_sample(mouse_index: int, target: np.ndarray, num_neurons: Sequence[int]) -> Tuple[List[np.ndarray], np.ndarray]:
    temporal = [target.shape[-1]] if target.ndim == 2 else []
    targets = [target if m == mouse_index else np.zeros((n, *temporal), dtype=np.float32)
               for m, n in enumerate(num_neurons)]
    weights = np.zeros(len(num_neurons), dtype=np.float32)
    weights[mouse_index] = 1.0
    return targets, weights


def collate(samples):
    """samples: list of (input (5,T,H,W), (targets list, weights)) -> (B,5,T,H,W), ([ (B,N_m,T) ], (B,n_mice))."""
    x = np.stack([s[0] for s in samples])
    n_mice = len(samples[0][1][0])
    targets = [np.stack([s[1][0][m] for s in samples]) for m in range(n_mice)]
    weights = np.stack([s[1][1] for s in samples])
    return x, (targets, weights)


def assemble_batch(trials, picks, num_neurons, size, pad_fill_value, window, boxes, lams=None):
    """End-to-end restatement of ``ConcatMiceVideoDataset.__getitem__`` + collate for explicit picks.

    trials[mouse][trial] = dict(video (H0,W0,L), behavior (2,L), pupil_center (2,L), responses (N,L));
    picks = [(mouse, trial, end_frame, partner or None)] with partner = (trial, end_frame) of the same mouse;
    window = (size, step); boxes[i] = cut-mix box or None; lams[i] = Mixup factor or None.
    """
    samples = []
    lams = lams if lams is not None else [None] * len(picks)
    for (mouse, trial, end, partner), box, lam in zip(picks, boxes, lams):
        def one(tr, e):
            d = trials[mouse][tr]
            idx = window_indexes(e, *window)
            return (stack_inputs(d["video"][..., idx], d["behavior"][..., idx], d["pupil_center"][..., idx], size,
                                 pad_fill_value), responses_to_target(d["responses"][..., idx]))
        x, t = one(trial, end)
        if box is not None:
            x2, t2 = one(*partner)
            x, t = cutmix_apply(x, t, x2, t2, box)
        elif lam is not None:
            x2, t2 = one(*partner)
            x, t = mixup_apply(x, t, x2, t2, lam)
        samples.append((x, mice_sample(mouse, t, num_neurons)))
    return collate(samples)
