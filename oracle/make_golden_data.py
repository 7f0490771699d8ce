#!/usr/bin/env python3
"""Golden vectors for the batch-assembly path, from the REAL reference (build container only; needs /root/reference).

Loads ``src/inputs.py`` and ``src/mixers.py`` by file path (both depend on numpy/torch only), drives
``StackInputsProcessor`` and ``CutMix`` on seeded synthetic trials, checks ``oracle/data_oracle.py`` against them and
stores inputs + reference outputs in tests/golden/data_pipeline.npz.  ``responses_to_tensor`` (responses.py:25-29) sits
in a module that imports ``src.constants`` through the package ``__init__`` (which needs argus), so its two lines
(float32 + relu) are applied here with torch directly.
"""
from __future__ import annotations

import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
from oracle import data_oracle as dorc  # noqa: E402


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def mixup_fixture(ref_inputs, ref_mixers):
    """tests/golden/data_mixup.npz: ``Mixup`` and ``RandomChoiceMixer([CutMix, Mixup])`` (src/mixers.py:22-33,70-79) driven on
    seeded synthetic trials; the oracle's restatement is checked against them here."""
    rng = np.random.default_rng(20231123)
    out, case = {}, 0
    for (h0, w0, size, fill, vid_dtype) in [(36, 64, (64, 64), 0.0, np.uint8), (9, 13, (16, 12), 3.5, np.float32)]:
        length, n = 40, 11
        trial = []
        for _ in range(2):
            video = rng.integers(0, 256, size=(h0, w0, length)).astype(vid_dtype)
            beh = (rng.normal(size=(2, length)) * 10 + 20).astype(np.float32)
            pup = (rng.normal(size=(2, length)) * 20 + 90).astype(np.float32)
            resp = (rng.normal(size=(n, length)) * 5).astype(np.float32)
            trial.append((video, beh, pup, resp))
        proc = ref_inputs.StackInputsProcessor(size=size, pad_fill_value=fill)
        idx = (dorc.window_indexes(35, 8, 2), dorc.window_indexes(29, 8, 2))
        s = []
        for (video, beh, pup, resp), ix in zip(trial, idx):
            s.append((proc(video[..., ix], beh[..., ix], pup[..., ix]), torch.relu(torch.from_numpy(resp[..., ix].astype(np.float32)))))
        mixup = ref_mixers.Mixup(alpha=0.4, prob=0.7)
        choice = ref_mixers.RandomChoiceMixer([ref_mixers.CutMix(alpha=1.0), ref_mixers.Mixup(alpha=0.4)], [0.5, 0.5], prob=1.0)
        for seed in range(5):
            key = f"c{case}_s{seed}"
            np.random.seed(2000 + seed + 31 * case)
            used = mixup.use()
            rs = np.random.RandomState(2000 + seed + 31 * case)
            lam = dorc.mixup_draw(rs, 0.4, 0.7)
            assert (lam is not None) == bool(used)
            out[key + "_mixup_used"] = np.array(used)
            if used:
                xm, tm = mixup(s[0], s[1])
                mx, mt = dorc.mixup_apply(s[0][0].numpy(), s[0][1].numpy(), s[1][0].numpy(), s[1][1].numpy(), lam)
                assert np.array_equal(mx, xm.numpy()) and np.array_equal(mt, tm.numpy()), key
                out[key + "_mixup_lam"], out[key + "_mixup_x"], out[key + "_mixup_t"] = np.array(lam), xm.numpy(), tm.numpy()
            # RandomChoiceMixer: use, choice, then the chosen mixer's draws
            np.random.seed(3000 + seed + 31 * case)
            assert choice.use()
            xm, tm = choice(s[0], s[1])
            rs = np.random.RandomState(3000 + seed + 31 * case)
            assert rs.random_sample() < 1.0
            which = int(rs.choice(2, p=[0.5, 0.5]))
            out[key + "_choice"] = np.array(which)
            if which == 0:
                box = dorc.cutmix_draw(rs, s[0][0].shape[-2], s[0][0].shape[-1], 1.0, None)
                mx, mt = dorc.cutmix_apply(s[0][0].numpy(), s[0][1].numpy(), s[1][0].numpy(), s[1][1].numpy(), box)
                out[key + "_choice_box"] = np.array(box)
            else:
                lam2 = float(rs.beta(0.4, 0.4))
                mx, mt = dorc.mixup_apply(s[0][0].numpy(), s[0][1].numpy(), s[1][0].numpy(), s[1][1].numpy(), lam2)
                out[key + "_choice_lam"] = np.array(lam2)
            assert np.array_equal(mx, xm.numpy()) and np.array_equal(mt, tm.numpy()), key
            out[key + "_choice_x"], out[key + "_choice_t"] = xm.numpy(), tm.numpy()
        for i, (video, beh, pup, resp) in enumerate(trial):
            out[f"c{case}_video{i}"], out[f"c{case}_beh{i}"] = video, beh
            out[f"c{case}_pup{i}"], out[f"c{case}_resp{i}"] = pup, resp
        out[f"c{case}_meta"] = np.array([h0, w0, size[0], size[1], 35, 29, 8, 2], dtype=np.int64)
        out[f"c{case}_fill"] = np.array(fill, dtype=np.float32)
        case += 1
    out["num_cases"] = np.array(case)
    path = ROOT / "tests" / "golden" / "data_mixup.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size, "bytes")


def main():
    ref_inputs = load_by_path("ref_inputs", "src/inputs.py")
    ref_mixers = load_by_path("ref_mixers", "src/mixers.py")
    rng = np.random.default_rng(20231122)
    out = {}
    case = 0
    for (h0, w0, size, fill, vid_dtype) in [(36, 64, (64, 64), 0.0, np.float32), (36, 64, (64, 36), 0.0, np.uint8),
                                            (9, 13, (16, 12), 3.5, np.float64), (5, 8, (8, 5), 0.0, np.uint8)]:
        length, n = 40, 11
        trial = []
        for _ in range(2):
            video = rng.integers(0, 256, size=(h0, w0, length)).astype(vid_dtype)
            beh = (rng.normal(size=(2, length)) * 10 + 20).astype(np.float32)
            pup = (rng.normal(size=(2, length)) * 20 + 90).astype(np.float32)
            resp = (rng.normal(size=(n, length)) * 5).astype(np.float32)
            trial.append((video, beh, pup, resp))
        proc = ref_inputs.StackInputsProcessor(size=size, pad_fill_value=fill)
        idx1 = dorc.window_indexes(35, 8, 2)
        idx2 = dorc.window_indexes(29, 8, 2)
        s = []
        for (video, beh, pup, resp), idx in zip(trial, (idx1, idx2)):
            x = proc(video[..., idx], beh[..., idx], pup[..., idx])
            t = torch.relu(torch.from_numpy(resp[..., idx].astype(np.float32)))
            s.append((x, t))
            mine = dorc.stack_inputs(video[..., idx], beh[..., idx], pup[..., idx], size, fill)
            assert np.array_equal(mine, x.numpy())
            assert np.array_equal(dorc.responses_to_target(resp[..., idx]), t.numpy())
        mixer = ref_mixers.CutMix(alpha=1.0, prob=0.5)
        for seed in range(6):
            np.random.seed(1000 + seed + 17 * case)
            used = mixer.use()
            key = f"c{case}_s{seed}"
            out[key + "_used"] = np.array(used)
            rs = np.random.RandomState(1000 + seed + 17 * case)
            box = dorc.cutmix_draw(rs, s[0][0].shape[-2], s[0][0].shape[-1], 1.0, 0.5)
            assert (box is not None) == bool(used)
            if used:
                xm, tm = mixer(s[0], s[1])
                mx, mt = dorc.cutmix_apply(s[0][0].numpy(), s[0][1].numpy(), s[1][0].numpy(), s[1][1].numpy(), box)
                assert np.array_equal(mx, xm.numpy()), key
                assert np.array_equal(mt, tm.numpy()), key
                out[key + "_box"] = np.array(box)
                out[key + "_x"] = xm.numpy()
                out[key + "_t"] = tm.numpy()
        for i, (video, beh, pup, resp) in enumerate(trial):
            out[f"c{case}_video{i}"], out[f"c{case}_beh{i}"] = video, beh
            out[f"c{case}_pup{i}"], out[f"c{case}_resp{i}"] = pup, resp
        out[f"c{case}_x0"], out[f"c{case}_t0"] = s[0][0].numpy(), s[0][1].numpy()
        out[f"c{case}_meta"] = np.array([h0, w0, size[0], size[1], 35, 29, 8, 2], dtype=np.int64)
        out[f"c{case}_fill"] = np.array(fill, dtype=np.float32)
        case += 1
    mixup_fixture(ref_inputs, ref_mixers)
    out["num_cases"] = np.array(case)
    path = ROOT / "tests" / "golden" / "data_pipeline.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size, "bytes;",
          sum(1 for k in out if k.endswith("_box")), "cut-mix cases")


if __name__ == "__main__":
    main()
