#!/usr/bin/env python3
"""Per-parameter gradient *gain* of the bf16 path against the fp32 HIP path (round-3 verdict item 3):

    gain(p) = <g16, g32> / ||g32||^2        rel(p) = ||g16 - g32|| / ||g32||

at B=2, T=8 (the digest shape) and at the metric batch (B=32, T=32), plus a bisection over *stages*: the model is run with one
stage (stem, block 0..8, head) in bf16 and the rest in fp32 — and the other way round — to find whose bf16 storage produces the
gain error at the far end of the backward chain (core.stem.0.weight).  Lives under tests/ because it takes its seeded weights
from the oracle's generator.  `python tests/bf16_gain_report.py [small] [full] [bisect]` on a GPU box; DWN_DETERMINISTIC=1
selects the ordered build (repeatable to the bit, so differences between variants are not summation noise).
Writes gpurun_out/r4_bf16_gain.json.
"""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

from oracle import dwiseneuro_oracle as orc
from tests.gpu_helpers import analytically_zero_grad, synth_inputs

dev = torch.device("cuda", 0)
F32, BF = torch.float32, torch.bfloat16
STAGES = ["stem"] + [f"block{i}" for i in range(9)] + ["head"]


def build(seed=11):
    from sensorium_amd import DwiseNeuro
    sd = orc.make_state_dict(readout_outputs=(7863,), expansion_ratio=7, seed=seed)
    m = DwiseNeuro(readout_outputs=(7863,), expansion_ratio=7, drop_rate=0.0, drop_path_rate=0.0)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).train()


def forward_mixed(model, x, dtypes):
    """DwiseNeuro.forward with a storage dtype per stage (DepthwiseCore.forward restated with casts between stages)."""
    from sensorium_amd import ops
    core = model.core
    bn = core.stem[1].bn
    mods = list(core.blocks)
    pes, blks = mods[0::2], mods[1::2]
    _, _, t, h, w = x.shape
    sizes = []
    for blk in blks:
        sizes.append((h, w))
        s = blk.spatial_stride
        h, w = (h - 1) // s + 1, (w - 1) // s + 1
    tables = [blk.geometry(pe, t, hw[0], hw[1], x.device)[:3] for pe, blk, hw in zip(pes, blks, sizes)]
    a = ops.StemFn.apply(x, core.stem[0].weight, bn.weight, bn.bias, core, dtypes[0], tables[0])
    for i, (pe, blk) in enumerate(zip(pes, blks)):
        d = dtypes[1 + i]
        a = blk(a.to(d), pe, d, True, tables[i + 1] if i + 1 < len(blks) else None)
    d = dtypes[10]
    a = ops.PoolFn.apply(a.to(d))
    feats = model.cortex(a, d)
    return [r(feats) for r in model.readouts]


def grads(model, x, t, w, dtypes):
    from sensorium_amd import MicePoissonLoss
    model.zero_grad(set_to_none=True)
    preds = forward_mixed(model, x, dtypes)
    loss = MicePoissonLoss()(preds, ([t], w))
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().double().clone() for k, p in model.named_parameters()}


def table(g16, g32):
    out = {}
    for k in g32:
        if analytically_zero_grad(k):
            continue
        n2 = float((g32[k] * g32[k]).sum())
        out[k] = {"gain": float((g16[k] * g32[k]).sum()) / n2, "rel": float((g16[k] - g32[k]).norm()) / n2 ** 0.5}
    return out


def total_norm(g):
    return sum(float(v.norm()) ** 2 for v in g.values()) ** 0.5


KEYS = ["core.stem.0.weight", "core.stem.1.bn.weight", "core.blocks.1.conv_pw.0.weight", "core.blocks.1.spat_covn_dw.0.weight",
        "core.blocks.1.conv_pwl.0.weight", "core.blocks.3.conv_pw.0.weight", "core.blocks.9.conv_pw.0.weight",
        "core.blocks.17.conv_pw.0.weight", "cortex.layers.0.conv.weight", "readouts.0.layer.1.weight"]


def brief(tab):
    return {k.replace("core.", "").replace(".weight", ""): (round(tab[k]["gain"], 4), round(tab[k]["rel"], 4)) for k in KEYS}


def batch(b, t):
    rng = np.random.default_rng(20231122)
    x, targets, _ = synth_inputs(rng, b, t, 36, 64, (7863,))
    return torch.from_numpy(x).to(dev), torch.from_numpy(targets[0]).to(dev), torch.ones(b, 1, device=dev)


def main():
    which = sys.argv[1:] or ["small", "bisect", "full"]
    out = {"deterministic_build": os.environ.get("DWN_DETERMINISTIC", "0") == "1"}
    model = build()
    for name, (b, t) in (("small", (2, 8)), ("full", (32, 32))):
        if name not in which:
            continue
        x, tg, w = batch(b, t)
        l32, g32 = grads(model, x, tg, w, [F32] * 11)
        l16, g16 = grads(model, x, tg, w, [BF] * 11)
        l16b, g16b = grads(model, x, tg, w, [BF] * 11)
        tab = table(g16, g32)
        worst = sorted(tab.items(), key=lambda kv: -abs(kv[1]["gain"] - 1))[:12]
        out[name] = {"B": b, "T": t, "loss_fp32": l32, "loss_bf16": l16, "total_norm_ratio": total_norm(g16) / total_norm(g32),
                     "total_norm_ratio_second_run": total_norm(g16b) / total_norm(g32), "per_parameter": tab}
        print(f"== {name} B={b} T={t}: loss {l32:.4f} / {l16:.4f}; total-norm ratio {out[name]['total_norm_ratio']:.5f} "
              f"(second bf16 run {out[name]['total_norm_ratio_second_run']:.5f})", flush=True)
        print("   key params (gain, rel):", brief(tab), flush=True)
        print("   second run            :", brief(table(g16b, g32)), flush=True)
        print("   worst gains:", [(k, round(v["gain"], 4)) for k, v in worst], flush=True)
        if name == "small" and "bisect" in which:
            bis = {}
            for i, st in enumerate(STAGES):
                only = [F32] * 11
                only[i] = BF
                _, g = grads(model, x, tg, w, only)
                allbut = [BF] * 11
                allbut[i] = F32
                _, g2 = grads(model, x, tg, w, allbut)
                bis[st] = {"only_this_bf16": brief(table(g, g32)), "all_but_this_bf16": brief(table(g2, g32))}
                print(f"   only {st:7s} bf16:", bis[st]["only_this_bf16"], flush=True)
                print(f"   all but {st:7s}  :", bis[st]["all_but_this_bf16"], flush=True)
            # cumulative: stages 0..k in bf16
            for k in range(11):
                d = [BF if i <= k else F32 for i in range(11)]
                _, g = grads(model, x, tg, w, d)
                bis[f"upto_{STAGES[k]}"] = brief(table(g, g32))
                print(f"   stem..{STAGES[k]:7s} bf16:", bis[f"upto_{STAGES[k]}"], flush=True)
            out["bisect_small"] = bis
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    tag = "_det" if out["deterministic_build"] else ""
    (ROOT / "gpurun_out" / f"r4_bf16_gain{tag}.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
