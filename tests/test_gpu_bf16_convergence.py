"""Does bf16 storage (the benchmarked dtype) TRAIN like fp32?  (round-4 verdict: 7-9 % of rounding noise on every core weight
gradient, against 30 steps on one memorised batch as the only convergence evidence.)

A fixed random *teacher* DwiseNeuro (full width: nine blocks, expansion 7, 7863 neurons) turns 16 + 4 distinct synthetic
batches (B = 8, T = 16, 36x64) into response targets — a learnable task, unlike noise targets.  A student of the same
architecture and another seed is trained for 320 ``MouseModel.train_step``s (AdamW + EMA, the reference's lr rule 3e-4 * B / 4,
argus_models.py:43-71) twice from the same state: fp32 storage and bf16 storage.  Compared: the loss curves (same batches in
the same order, so step by step) and the single-trial correlation (``corr``, src/metrics.py:11-31) of the EMA network's
predictions on the four HELD-OUT batches against the teacher's responses.  The report goes to gpurun_out/ (copied to
profiles/r5_bf16_convergence.json)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dwiseneuro_oracle as orc  # noqa: E402
from tests.gpu_helpers import dev, synth_inputs  # noqa: E402

N = 7863
B, T, H, W = 8, 16, 36, 64
N_TRAIN, N_HELD, STEPS = 16, 4, 320


def _teacher_targets(data_seed=505, teacher_seed=71):
    """Inputs and the teacher's responses: the teacher runs in train mode (batch statistics — a random-init network has no
    meaningful running statistics) in fp32, no_grad; its softplus outputs are the targets."""
    from sensorium_amd import DwiseNeuro
    teacher = DwiseNeuro(readout_outputs=(N,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    teacher.load_state_dict(orc.make_state_dict(readout_outputs=(N,), expansion_ratio=7, seed=teacher_seed), strict=True)
    teacher = teacher.to(dev()).train()
    rng = np.random.default_rng(data_seed)
    xs, ts = [], []
    with torch.no_grad():
        for _ in range(N_TRAIN + N_HELD):
            x, _, _ = synth_inputs(rng, B, T, H, W, (N,))
            x = torch.from_numpy(x).to(dev())
            xs.append(x)
            ts.append(teacher(x)[0].float().clone())
    del teacher
    return xs, ts


def _corr_heldout(module, xs, ts, bf16):
    preds, tgts = [], []
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        for x, t in zip(xs, ts):
            p = module(x)[0].float()
            preds.append(p.permute(0, 2, 1).reshape(-1, N).cpu().numpy())
            tgts.append(t.permute(0, 2, 1).reshape(-1, N).cpu().numpy())
    return float(orc.corr(np.concatenate(preds), np.concatenate(tgts), axis=0).mean())


# (tag, data seed, teacher seed, student seed, dropout, drop-path, bound on the epoch-mean loss gap / drop, bounds on |corr difference| EMA / raw)
CONFIGS = [
    ("plain", 505, 71, 11, 0.0, 0.0, 1e-2, 2e-3, 5e-3),
    # round-5 verdict item 7c: a second seed WITH the reference's regularisation on (Dropout1d 0.4 in the readout, DropPath 0.1 in core
    # and cortex: dwiseneuro.py:256,317,382; true_batch_001.py:36-37).  Both modes draw the same masks (the torch generator is
    # re-seeded before each run and both make the same draws per step).  Same bounds as the plain run; measured: epoch-mean loss gap
    # 4e-4 of the drop, held-out correlation of the EMA network 0.71015 (fp32) vs 0.71024 (bf16): profiles/r6_bf16_convergence_drop.json
    ("drop", 606, 72, 12, 0.4, 0.1, 1e-2, 2e-3, 5e-3),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_bf16_training_converges_like_fp32_on_a_teacher_task(cfg):
    from sensorium_amd.argus_models import MouseModel
    tag, data_seed, teacher_seed, student_seed, drop_rate, drop_path, b_gap, b_ema, b_raw = cfg
    xs, ts = _teacher_targets(data_seed, teacher_seed)
    w = torch.ones(B, 1, device=dev())
    held_x, held_t = xs[N_TRAIN:], ts[N_TRAIN:]
    losses, corr_ema, corr_raw = {}, {}, {}
    for mode, bf in (("fp32", False), ("bf16", True)):
        torch.manual_seed(1234)                      # the same dropout / drop-path draws in both modes
        params = {"nn_module": ("dwiseneuro", dict(readout_outputs=(N,), expansion_ratio=7, drop_rate=drop_rate, drop_path_rate=drop_path)),
                  "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 3e-4 * B / 4, "weight_decay": 0.05}),
                  "device": "cuda:0", "amp": bf, "iter_size": 1}
        mm = MouseModel(params)
        mm.nn_module.load_state_dict(orc.make_state_dict(readout_outputs=(N,), expansion_ratio=7, seed=student_seed), strict=True)
        mm.set_ema(0.98)
        cur = []
        for step in range(STEPS):
            i = step % N_TRAIN
            cur.append(mm.train_step([xs[i], [[ts[i].clone()], w]])["loss"])
        losses[mode] = np.array(cur)
        # held-out single-trial correlation: the EMA network in eval mode (what val_step evaluates, argus_models.py:73-87) and
        # the trained network with batch statistics
        mm.eval()
        corr_ema[mode] = _corr_heldout(mm.model_ema.ema.eval(), held_x, held_t, bf)
        mm.nn_module.train()
        corr_raw[mode] = _corr_heldout(mm.nn_module, held_x, held_t, bf)
        del mm
        torch.cuda.empty_cache()
    # epoch means (16 steps): the step-by-step curves carry the batch-to-batch spread of the task itself
    ep32 = losses["fp32"].reshape(-1, N_TRAIN).mean(1)
    ep16 = losses["bf16"].reshape(-1, N_TRAIN).mean(1)
    drop = float(ep32[0] - ep32[-1])
    gap_epoch = float(np.max(np.abs(ep16 - ep32)))
    gap_step = float(np.max(np.abs(losses["bf16"] - losses["fp32"])))
    report = {
        "task": f"teacher -> student, full width, B={B} T={T} {H}x{W}, {N_TRAIN} train + {N_HELD} held-out batches, {STEPS} steps, "
                f"AdamW lr {3e-4 * B / 4:g} wd 0.05, EMA 0.98; seeds data {data_seed} teacher {teacher_seed} student {student_seed}; "
                f"dropout {drop_rate}, drop-path {drop_path}",
        "loss_first_epoch": {"fp32": float(ep32[0]), "bf16": float(ep16[0])},
        "loss_last_epoch": {"fp32": float(ep32[-1]), "bf16": float(ep16[-1])},
        "loss_drop_fp32": drop,
        "max_epoch_mean_gap": gap_epoch, "max_epoch_mean_gap_over_drop": gap_epoch / drop,
        "max_step_gap": gap_step, "max_step_gap_over_drop": gap_step / drop,
        "heldout_corr_ema": corr_ema, "heldout_corr_trained_net_batch_stats": corr_raw,
        "epoch_mean_loss_fp32": [float(v) for v in ep32], "epoch_mean_loss_bf16": [float(v) for v in ep16],
    }
    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    if out.is_dir():
        (out / f"r6_bf16_convergence_{tag}.json").write_text(json.dumps(report, indent=1))
    print(json.dumps({k: report[k] for k in ("loss_drop_fp32", "max_epoch_mean_gap_over_drop", "max_step_gap_over_drop",
                                             "heldout_corr_ema", "heldout_corr_trained_net_batch_stats")}))
    assert drop > 0 and ep16[-1] < ep16[0], "the task must be learnable in both modes"
    assert corr_raw["fp32"] > 0.2, ("the student must actually have learnt the teacher", corr_raw)
    # the bars the round-4 verdict set: loss curves within 1 % of the drop, held-out correlation within 2e-3
    assert gap_epoch <= b_gap * drop, (gap_epoch, drop)
    assert abs(corr_ema["bf16"] - corr_ema["fp32"]) <= b_ema, corr_ema
    # the un-averaged network after its last step jitters more than its EMA (three runs on three boxes: bf16 - fp32 = +1.3e-3,
    # +2.4e-3, +2.4e-3 against +7e-4, +3e-4 for the EMA network): reported, bounded at twice the largest difference seen
    assert abs(corr_raw["bf16"] - corr_raw["fp32"]) <= b_raw, corr_raw
