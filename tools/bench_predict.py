#!/usr/bin/env python3
"""Sliding-window trial inference timing (reference: src/predictors.py:36-55,
scripts/predict.py:24-50).  One model, one trial of L frames at the reference's window (16 frames, step 2); reports
trials/s for the reference's launch pattern (one window per forward, device->host copy per window is already gone)
and for batched windows with / without hipGraph replay.  Not the headline metric: prints a small JSON summary.

    python tools/bench_predict.py [--length 300 --height 64 --width 64 --expansion 7 --dtype fp32]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402


def eval_executed_bytes(batch, frames, height, width, expansion, esize, n_neurons, rc_blocks):
    """HBM bytes ONE eval-mode forward of `batch` windows has to move in the pass structure the library runs:
    per block, `rc_blocks` (bf16, 64 / 128 input channels) read a0 and write y2 from the y1-rebuilding stencil; the others
    write + read y1 through conv_pw and the stencil; then y2 read -> z3 write (temporal pass, SE pooling inside), z3 read ->
    y4 write (conv_pwl), the C-wide residual pass (x, y4 read; out write); plus stem, pool, cortex and one readout."""
    from bench import CORE_FEATURES, STRIDES
    total = 0
    h, w = height, width
    total += batch * frames * h * w * (5 * 4 + 2 * CORE_FEATURES[0] * esize)            # stem: fp32 input, y0 write + read + out write ~
    for i, st in enumerate(STRIDES):
        cin, cout = CORE_FEATURES[i], CORE_FEATURES[min(i + 1, len(CORE_FEATURES) - 1)]
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        mi, mo, e_ = batch * frames * h * w, batch * frames * ho * wo, cin * expansion
        if i in rc_blocks:
            blk = mi * cin + mo * e_                                                     # a0 -> y2
        else:
            blk = mi * cin + 2 * mi * e_ + mo * e_                                       # a0 -> y1 -> y2
        blk += 2 * mo * e_                                                                # y2 -> z3
        blk += mo * e_ + mo * cout                                                        # z3 -> y4
        blk += mi * cin + 2 * mo * cout                                                   # residual: x (gathered), y4 -> out
        total += blk * esize
        h, w = ho, wo
    m = batch * frames
    cx = (CORE_FEATURES[-1], 1024, 2048, 4096)
    total += esize * (m * h * w * cx[0] + sum(3 * m * b for b in cx[1:]))                 # pool read, cortex y / out
    total += esize * sum(a * b // 2 for a, b in zip(cx[:-1], cx[1:]))                     # cortex weights
    npad = (n_neurons + 1) // 2 * 2
    total += esize * (m * 4096 + 2048 * npad) + 4 * m * npad                              # readout: x, weights, fp32 predictions
    return int(total)


def ensemble_bench(folds=7, length=300, height=64, width=64, expansion=7, dtype="bf16", windows=32, repeats=3, device=None,
                   use_graph=True, warmup=True):
    """BASELINE.json configs[4]: `folds` models (scripts/predict.py:44-50), one trial of `length` frames at 64x64, window 16
    step 2 (src/predictors.py:36-55), every fold inside one captured hipGraph per window batch.  Returns trials/s and the
    HBM fraction of the bytes the eval path executes (not of the training pass structure)."""
    import numpy as np
    import sensorium_amd._lib as L
    from bench import HBM_PEAK_GBS, NUM_NEURONS_MOUSE0, CORE_FEATURES, STRIDES, model_params
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.predictors import EnsemblePredictor
    dev = device or torch.device("cuda", 0)
    params = model_params(expansion)
    params["device"] = str(dev)
    params["amp"] = False
    models = []
    for k in range(folds):
        torch.manual_seed(100 + k)
        mk = MouseModel(params)
        if dtype == "bf16":
            mk.nn_module.compute_dtype = torch.bfloat16
        models.append(mk)
    g = torch.Generator().manual_seed(1)
    inputs = torch.zeros(5, length, height, width)
    inputs[0] = torch.randint(0, 256, (length, height, width), generator=g).float()
    inputs[1:] = (torch.rand(4, length, 1, 1, generator=g) * 50)
    ens = EnsemblePredictor(models, frame_stack_size=16, frame_stack_step=2, windows_per_batch=windows, use_graph=use_graph)
    if warmup:
        r = ens.predict_trial(inputs, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        r = ens.predict_trial(inputs, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / repeats
    es = 2 if dtype == "bf16" else 4
    dt_code = L.DWN_BF16 if dtype == "bf16" else L.DWN_F32
    rc_blocks, h, w = [], height, width
    for i, st in enumerate(STRIDES):
        if L.lib.dwn_dw_spatial_rc_supported(dt_code, CORE_FEATURES[i], CORE_FEATURES[i] * expansion, 3, st, h, w):
            rc_blocks.append(i)
        h, w = (h - 1) // st + 1, (w - 1) // st + 1
    nwin = length - 30
    nbatch = -(-nwin // windows)
    per_fwd = eval_executed_bytes(windows, 16, height, width, expansion, es, NUM_NEURONS_MOUSE0, rc_blocks)
    tail = nwin - (nbatch - 1) * windows
    per_trial = folds * ((nbatch - 1) * per_fwd + eval_executed_bytes(tail, 16, height, width, expansion, es, NUM_NEURONS_MOUSE0, rc_blocks))
    return {"folds": folds, "length": length, "hw": [height, width], "window": [16, 2], "windows_per_forward": windows,
            "dtype": dtype, "trials_per_s": round(1.0 / dt, 3), "ms_per_trial": round(dt * 1e3, 1),
            "windows_per_s_per_model": round(nwin / dt, 1),
            "executed_bytes_per_trial": per_trial, "achieved_GBs": round(per_trial / dt / 1e9, 1),
            "hbm_frac": round(per_trial / dt / 1e9 / HBM_PEAK_GBS, 4),
            "y1_rebuilt_in_blocks": rc_blocks, "finite": bool(np.isfinite(r).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--expansion", type=int, default=7)
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--folds", type=int, default=7, help="fold models of the ensemble leg (scripts/predict.py:44-50: 7)")
    ap.add_argument("--windows", type=int, default=32, help="windows per forward of the ensemble leg")
    ap.add_argument("--pmc-trial", action="store_true", help="exactly ONE ensemble trial, eager launches (no hipGraph, no warm-up): "
                    "the command to run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/predict_pmc.py sums the passes)")
    args = ap.parse_args()
    if args.pmc_trial:
        print(json.dumps(ensemble_bench(folds=args.folds, length=args.length, height=args.height, width=args.width,
                                        expansion=args.expansion, dtype="bf16" if args.dtype == "bf16" else "fp32",
                                        windows=args.windows, repeats=1, use_graph=False, warmup=False)))
        return
    from bench import HBM_PEAK_GBS, NUM_NEURONS_MOUSE0, family_work, model_params
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.predictors import EnsemblePredictor, Predictor

    dev = torch.device("cuda", 0)
    params = model_params(args.expansion)
    params["device"] = str(dev)
    params["amp"] = False
    torch.manual_seed(0)
    model = MouseModel(params)
    if args.dtype == "bf16":
        model.nn_module.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(1)
    inputs = torch.zeros(5, args.length, args.height, args.width)
    inputs[0] = torch.randint(0, 256, (args.length, args.height, args.width), generator=g).float()
    inputs[1:] = (torch.rand(4, args.length, 1, 1, generator=g) * 50)
    out = {"length": args.length, "hw": [args.height, args.width], "dtype": args.dtype, "expansion": args.expansion,
           "window": [16, 2], "neurons": NUM_NEURONS_MOUSE0, "trials_per_s": {}}
    ref = None
    W = args.windows
    for name, wpb, graph in (("one_window_per_forward", 1, False), (f"{W}_windows_per_forward", W, False),
                             (f"{W}_windows_per_forward_hipgraph", W, True)):
        pred = Predictor(model, frame_stack_size=16, frame_stack_step=2, windows_per_batch=wpb, use_graph=graph)
        r = pred.predict_trial(inputs, 0)                       # warm-up (graph capture included)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.repeats):
            r = pred.predict_trial(inputs, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.repeats
        out["trials_per_s"][name] = round(1.0 / dt, 3)
        if ref is None:
            ref = r
        else:
            import numpy as np
            out.setdefault("max_rel_diff_vs_one_window", {})[name] = float(np.abs(r - ref).max() / (np.abs(ref).max() + 1e-12))
    # ---- BASELINE.json configs[4]: the 7-fold ensemble, every fold inside one captured graph per window batch
    if args.folds > 1:
        import numpy as np
        models = [model]
        for k in range(1, args.folds):
            torch.manual_seed(100 + k)
            mk = MouseModel(params)
            if args.dtype == "bf16":
                mk.nn_module.compute_dtype = torch.bfloat16
            models.append(mk)
        legs = {}
        for name, cls_args in (("sequential_predictors_hipgraph", None), ("one_graph_all_folds", True)):
            if cls_args is None:
                preds = [Predictor(m, frame_stack_size=16, frame_stack_step=2, windows_per_batch=args.windows, use_graph=True) for m in models]
                run = lambda: np.mean([p.predict_trial(inputs, 0) for p in preds], axis=0)
            else:
                ens = EnsemblePredictor(models, frame_stack_size=16, frame_stack_step=2, windows_per_batch=args.windows, use_graph=True)
                run = lambda: ens.predict_trial(inputs, 0)
            r = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.repeats):
                r = run()
            torch.cuda.synchronize()
            legs[name] = ((time.perf_counter() - t0) / args.repeats, r)
        es = 2 if args.dtype == "bf16" else 4
        alg, _ = family_work(args.windows, 16, args.height, args.width, args.expansion, (NUM_NEURONS_MOUSE0,), [], es)
        fwd_bytes = sum(alg[k] for k in ("pw_fwd", "dws_fwd", "dwt_fwd", "se_pool", "pwl_fwd"))       # one window batch, one fold
        nbatch = -(-(args.length - 30) // args.windows)
        dt = legs["one_graph_all_folds"][0]
        out["ensemble"] = {
            "folds": args.folds, "windows_per_forward": args.windows,
            "trials_per_s": {k: round(1.0 / v[0], 3) for k, v in legs.items()},
            "max_rel_diff_between_legs": float(np.abs(legs["one_graph_all_folds"][1] - legs["sequential_predictors_hipgraph"][1]).max()
                                               / (np.abs(legs["sequential_predictors_hipgraph"][1]).max() + 1e-12)),
            "algorithmic_forward_bytes_per_trial": int(fwd_bytes * nbatch * args.folds),
            "achieved_GBs": round(fwd_bytes * nbatch * args.folds / dt / 1e9, 1),
            "hbm_frac": round(fwd_bytes * nbatch * args.folds / dt / 1e9 / HBM_PEAK_GBS, 4),
            "note": "algorithmic bytes = the E-wide forward passes of the materialised pass structure (conv_pw write, stencils, "
                    "SE pooling, conv_pwl read) per window batch x batches per trial x folds; the eval path moves fewer bytes than "
                    "this (bf16: no conv_pw pass on blocks 0-6, y1 rebuilt inside the stencil kernel; both dtypes: z3 and the SE "
                    "pooling sums come from the temporal pass, y3 is never stored), so hbm_frac is a rate of useful work, not of "
                    "traffic",
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
