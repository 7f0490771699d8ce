#!/usr/bin/env python3
"""bf16 (benchmark dtype) vs fp32 (parity mode) on the HIP path at FULL width and depth (expansion 7, 9 blocks, 7863 neurons):
  * forward / loss / gradient-norm errors of the bf16 run against the reference's digest (tests/golden/full_width_digest.npz),
  * per-parameter gradient cosine similarity bf16 vs fp32 (worst per block) and run-to-run gradient noise of each mode,
  * a 30-step training trajectory (AdamW + EMA, seeded synthetic batch) in both modes: loss curves, prediction correlation.
Writes profiles/r2_bf16_parity.json; tests/test_gpu_bf16_depth.py asserts bounds derived from these measurements.
Lives under tests/ because it drives the CPU oracle (test infrastructure): run as `python tests/bf16_parity_report.py` on a GPU box."""
import json
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

from oracle import dwiseneuro_oracle as orc
from tests.gpu_helpers import analytically_zero_grad, synth_inputs

dev = torch.device("cuda", 0)


def build(seed=11):
    from sensorium_amd import DwiseNeuro
    sd = orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=seed)
    m = DwiseNeuro(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).train()


def fwd_bwd(model, x, target, weights, bf16):
    from sensorium_amd import MicePoissonLoss
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
        preds = model(x)
        loss = MicePoissonLoss()(preds, ([target], weights))
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), preds[0].detach().float(), {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}


def main():
    out = {}
    z = np.load(ROOT / "tests/golden/full_width_digest.npz")
    rng = np.random.default_rng(20231122)
    x, targets, _ = synth_inputs(rng, 2, 8, 36, 64, (7863,))
    xd = torch.from_numpy(x).to(dev); td = torch.from_numpy(targets[0]).to(dev); wd = torch.ones(2, 1, device=dev)
    model = build()
    runs = {}
    for mode, bf in (("fp32", False), ("bf16", True)):
        a = fwd_bwd(model, xd, td, wd, bf)
        b = fwd_bwd(model, xd, td, wd, bf)
        runs[mode] = a
        tot = math.sqrt(sum(float(g.norm()) ** 2 for g in a[2].values()))
        # analytically-zero gradients (a per-channel shift in front of a BatchNorm) are pure rounding noise: skipped
        noise = {k: float((a[2][k] - b[2][k]).norm() / a[2][k].norm()) for k in a[2] if not analytically_zero_grad(k)}
        out[mode] = {"loss_rel_err_vs_reference": abs(a[0] - float(z["loss"])) / abs(float(z["loss"])),
                     "pred_l2_rel_err_vs_reference": abs(float(a[1].double().norm()) - float(z["pred_l2"])) / float(z["pred_l2"]),
                     "grad_total_norm_rel_err_vs_reference": abs(tot - float(z["grad_total_norm"])) / float(z["grad_total_norm"]),
                     "run_to_run_grad_noise_max": max(noise.values()), "run_to_run_grad_noise_worst": max(noise, key=noise.get),
                     "run_to_run_loss_diff": abs(a[0] - b[0]) / abs(a[0])}
    g32, g16 = runs["fp32"][2], runs["bf16"][2]
    tot = math.sqrt(sum(float(g.norm()) ** 2 for g in g32.values()))
    cos, relerr = {}, {}
    for k in g32:
        if analytically_zero_grad(k):
            continue                      # direction is noise
        cos[k] = float((g32[k] * g16[k]).sum() / (g32[k].norm() * g16[k].norm() + 1e-30))
        relerr[k] = float((g32[k] - g16[k]).norm() / g32[k].norm())
    per_block = {}
    for k, v in cos.items():
        parts = k.split(".")
        blk = ".".join(parts[:3]) if parts[0] == "core" and parts[1] == "blocks" else ".".join(parts[:2])
        per_block.setdefault(blk, []).append((v, relerr[k], k))
    out["bf16_vs_fp32_gradients"] = {b: {"min_cosine": min(v)[0], "worst_param": min(v)[2], "max_rel_err": max(e for _, e, _ in v)}
                                     for b, v in sorted(per_block.items())}
    out["bf16_vs_fp32_pred_rel_err"] = float((runs["bf16"][1] - runs["fp32"][1]).norm() / runs["fp32"][1].norm())
    # ---- 30-step trajectory
    from sensorium_amd.argus_models import MouseModel
    traj = {}
    finals = {}
    for mode, bf in (("fp32", False), ("bf16", True)):
        params = {"nn_module": ("dwiseneuro", dict(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)),
                  "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 3e-4, "weight_decay": 0.05}), "device": "cuda:0",
                  "amp": bf, "iter_size": 1}
        mm = MouseModel(params)
        mm.nn_module.load_state_dict(orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=11), strict=True)
        mm.set_ema(0.99)
        losses = []
        for _ in range(30):
            losses.append(mm.train_step([xd, [[td], wd]])["loss"])
        traj[mode] = losses
        mm.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
            finals[mode] = mm.nn_module(xd)[0].float().cpu().numpy()
    l32, l16 = np.array(traj["fp32"]), np.array(traj["bf16"])
    t = targets[0].transpose(0, 2, 1).reshape(-1, 7863)
    c32 = orc.corr(finals["fp32"].transpose(0, 2, 1).reshape(t.shape), t, axis=0).mean()
    c16 = orc.corr(finals["bf16"].transpose(0, 2, 1).reshape(t.shape), t, axis=0).mean()
    cc = np.corrcoef(finals["fp32"].ravel(), finals["bf16"].ravel())[0, 1]
    out["trajectory_30_steps"] = {"loss_fp32": l32.tolist(), "loss_bf16": l16.tolist(),
                                  "max_loss_gap_over_loss_drop": float(np.max(np.abs(l16 - l32)) / abs(l32[0] - l32[-1])),
                                  "final_loss_gap_over_loss_drop": float(abs(l16[-1] - l32[-1]) / abs(l32[0] - l32[-1])),
                                  "loss_drop_fp32": float(l32[0] - l32[-1]), "loss_drop_bf16": float(l16[0] - l16[-1]),
                                  "corr_vs_targets_fp32": float(c32), "corr_vs_targets_bf16": float(c16),
                                  "pearson_between_final_predictions": float(cc)}
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "r2_bf16_parity.json").write_text(json.dumps(out, indent=1))
    brief = {k: v for k, v in out.items() if k != "trajectory_30_steps"}
    print(json.dumps(brief, indent=1)[:6000])
    tr = out["trajectory_30_steps"]
    print({k: v for k, v in tr.items() if not k.startswith("loss_")})
    print("loss fp32", [round(v, 1) for v in tr["loss_fp32"][::5]], "bf16", [round(v, 1) for v in tr["loss_bf16"][::5]])


if __name__ == "__main__":
    main()
