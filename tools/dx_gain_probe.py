"""One InvertedResidual3d block at block 0's shape: gain of the bf16 input gradient (and of the weight gradients) against the fp32
HIP path, y1 stored vs y1-free (dwn_block_args.y1_mode) — is there a systematic difference between the two bf16 paths?"""
import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests.test_gpu_block import make_block
from tests.gpu_helpers import dev

def run(cin, cout, stride, B, T, H, W, seed=3):
    blk, pe = make_block(cin, cout, stride, 7, 32, seed=seed)
    blk = blk.to(dev()).train(); pe = pe.to(dev())
    g = torch.Generator().manual_seed(1)
    # correlated, offset inputs like a BatchNorm'ed stem output: 5 latent channels mixed into cin, plus per-channel offsets
    lat = torch.randn(B, T, H, W, 5, generator=g)
    mix = torch.randn(5, cin, generator=g)
    x = (lat @ mix) * 0.6 + torch.randn(cin, generator=g) * 0.5
    gout = torch.randn(B, T, (H - 1) // stride + 1, (W - 1) // stride + 1, cout, generator=g)
    res = {}
    for name, dtype, mode in (("f32", torch.float32, 1), ("stored", torch.bfloat16, 1), ("free", torch.bfloat16, 0), ("stored2", torch.bfloat16, 1), ("free2", torch.bfloat16, 0)):
        blk._dwn_y1_mode = mode
        blk.zero_grad(set_to_none=True)
        xd = x.to(dev()).to(dtype).requires_grad_(True)
        out = blk(xd, pe, dtype)
        out.backward(gout.to(dev()).to(dtype))
        torch.cuda.synchronize()
        res[name] = (xd.grad.double().clone(), {k: p.grad.double().clone() for k, p in blk.named_parameters()}, out.detach().double().clone())
    ref = res["f32"]
    for name in ("stored", "free", "stored2", "free2"):
        dx, gp, out = res[name]
        gain = float((dx * ref[0]).sum() / (ref[0] ** 2).sum())
        rel = float((dx - ref[0]).norm() / ref[0].norm())
        go = float((out * ref[2]).sum() / (ref[2] ** 2).sum())
        wg = float((gp["conv_pw.0.weight"] * ref[1]["conv_pw.0.weight"]).sum() / (ref[1]["conv_pw.0.weight"] ** 2).sum())
        g1 = float((gp["conv_pw.1.bn.weight"] * ref[1]["conv_pw.1.bn.weight"]).sum() / (ref[1]["conv_pw.1.bn.weight"] ** 2).sum())
        print(f"  {name:8s} dx gain {gain:.5f} rel {rel:.4f}  out gain {go:.5f}  dW1 gain {wg:.5f}  dgamma1 gain {g1:.5f}")
    a, b = res["stored"][0], res["free"][0]
    print(f"  free vs stored: dx rel diff {float((a - b).norm() / a.norm()):.4f}; stored vs stored2 {float((a - res['stored2'][0]).norm() / a.norm()):.4f}; "
          f"gain of free on stored {float((a * b).sum() / (a * a).sum()):.5f}")

for cfg in ((64, 64, 2, 8, 8, 36, 64), (64, 64, 1, 8, 8, 18, 32), (64, 64, 2, 32, 16, 36, 64)):
    print(cfg)
    run(*cfg)
