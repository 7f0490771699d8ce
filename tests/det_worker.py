"""Worker of tests/test_gpu_determinism.py: the same training steps twice from the same state, every result compared bit for
bit.  Run as a fresh process so that DWN_DETERMINISTIC (read when sensorium_amd._lib is imported) selects the library.
Prints one line: DET_WORKER deterministic=<0|1> identical=<0|1> differing=<n tensors> max_rel=<...>"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch


def run(kind: str):
    from sensorium_amd import _lib as L
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.synthetic import make_batch
    dev = torch.device("cuda:0")
    if kind.startswith("tiny"):
        kw = dict(readout_outputs=(24, 40), in_channels=5, core_features=(8, 8, 16), spatial_strides=(2, 1, 2), spatial_kernel=3,
                  temporal_kernel=5, expansion_ratio=3, se_reduce_ratio=4, cortex_features=(32, 64), groups=2,
                  softplus_beta=0.07, drop_rate=0.2, drop_path_rate=0.1)
        shape = (3, 6, 12, 16)
    else:
        # the metric architecture (configs/true_batch_001.py: nine blocks, expansion 7, 36x64 frames) at a small batch: every
        # kernel family of the benchmarked step runs, including the row-walk stencils and the fused conv_pw backward
        kw = dict(readout_outputs=(512,), in_channels=5, core_features=(64, 64, 64, 64, 128, 128, 128, 256, 256),
                  spatial_strides=(2, 1, 1, 1, 2, 1, 1, 2, 1), spatial_kernel=3, temporal_kernel=5, expansion_ratio=7,
                  se_reduce_ratio=32, cortex_features=(1024, 2048, 4096), groups=2, softplus_beta=0.07, drop_rate=0.4,
                  drop_path_rate=0.1)
        shape = (2, 8, 36, 64)
    amp = not kind.endswith("_f32")
    params = {"nn_module": ("dwiseneuro", kw), "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 1e-3, "weight_decay": 0.05}),
              "device": str(dev), "amp": amp, "iter_size": 1}
    results = []
    for rep in range(2):
        torch.manual_seed(1234)
        torch.cuda.manual_seed_all(1234)
        model = MouseModel(params)
        model.set_ema(0.99)
        batch = make_batch(*shape, kw["readout_outputs"], seed=7, device=dev)
        snap = {}
        for step in range(2):
            out = model.train_step(batch)
            snap[f"loss{step}"] = torch.tensor(out["loss"], dtype=torch.float64)
            for m, p in enumerate(out["prediction"]):
                snap[f"pred{step}.{m}"] = p.detach().float().cpu().clone()
            for n, p in model.nn_module.named_parameters():
                if p.grad is not None:
                    snap[f"grad{step}:{n}"] = p.grad.detach().cpu().clone()
        for n, t in model.nn_module.state_dict().items():
            snap[f"state:{n}"] = t.detach().cpu().clone()
        for n, t in model.model_ema.ema.state_dict().items():
            snap[f"ema:{n}"] = t.detach().cpu().clone()
        torch.cuda.synchronize()
        results.append(snap)
    a, b = results
    assert a.keys() == b.keys()
    differing, worst, worst_name = 0, 0.0, ""
    for k in a:
        if not torch.equal(a[k], b[k]):
            differing += 1
            d = float((a[k].double() - b[k].double()).abs().max()) / (float(a[k].double().abs().max()) + 1e-30)
            if d > worst:
                worst, worst_name = d, k
    # run-to-run noise of the FIRST step's outputs (later steps compound it through Adam's sign-like first update)
    from tests.gpu_helpers import analytically_zero_grad       # BatchNorm biases in front of another BatchNorm: pure rounding noise
    noise = []
    for k in a:
        if k.startswith(("grad0:", "pred0", "loss0")) and not (k.startswith("grad0:") and analytically_zero_grad(k[6:])):
            d = float((a[k].double() - b[k].double()).norm()) / (float(a[k].double().norm()) + 1e-30)
            noise.append((d, k))
    noise.sort(reverse=True)
    print("DET_NOISE step0 (relative L2 per tensor, worst five): " + "; ".join(f"{k} {d:.2e}" for d, k in noise[:5]), flush=True)
    print(f"DET_STEP0 max_rel_l2={noise[0][0]:.3e}", flush=True)
    print(f"DET_WORKER deterministic={int(L.DETERMINISTIC)} lib={L.LIB_PATH.name} tensors={len(a)} identical={int(differing == 0)} "
          f"differing={differing} max_rel={worst:.3e} worst={worst_name}", flush=True)


def run_predict():
    """Sliding-window prediction of one trial by a two-model ensemble (src/predictors.py:36-55), twice: bit-identical?"""
    from sensorium_amd import _lib as L
    from sensorium_amd.argus_models import MouseModel
    from sensorium_amd.predictors import EnsemblePredictor
    kw = dict(readout_outputs=(96,), in_channels=5, core_features=(64, 64, 128), spatial_strides=(2, 1, 2), spatial_kernel=3,
              temporal_kernel=5, expansion_ratio=7, se_reduce_ratio=32, cortex_features=(256, 512), groups=2, softplus_beta=0.07,
              drop_rate=0.0, drop_path_rate=0.0)
    outs = []
    for bf16 in (True, False):
        models = []
        for k in range(2):
            torch.manual_seed(50 + k)
            m = MouseModel({"nn_module": ("dwiseneuro", kw), "loss": ("mice_poisson", {}), "optimizer": ("AdamW", {"lr": 1e-3}),
                            "device": "cuda:0", "amp": False, "iter_size": 1})
            if bf16:
                m.nn_module.compute_dtype = torch.bfloat16
            models.append(m)
        g = torch.Generator().manual_seed(1)
        inputs = torch.zeros(5, 48, 64, 64)
        inputs[0] = torch.randint(0, 256, (48, 64, 64), generator=g).float()
        inputs[1:] = torch.rand(4, 48, 1, 1, generator=g) * 50
        ens = EnsemblePredictor(models, frame_stack_size=16, frame_stack_step=2, windows_per_batch=6, use_graph=False)
        outs.append([torch.from_numpy(ens.predict_trial(inputs, 0)) for _ in range(2)])
    differing = sum(int(not torch.equal(a, b)) for a, b in outs)
    worst = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in outs)
    finite = all(bool(torch.isfinite(a).all()) for a, _ in outs)
    print(f"DET_STEP0 max_rel_l2={worst:.3e}", flush=True)
    print(f"DET_WORKER deterministic={int(L.DETERMINISTIC)} lib={L.LIB_PATH.name} tensors={len(outs)} identical={int(differing == 0 and finite)} "
          f"differing={differing} max_rel={worst:.3e} worst=prediction", flush=True)


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if kind == "predict":
        run_predict()
    else:
        run(kind)
