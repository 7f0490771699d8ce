import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from tests.test_gpu_bf16_depth import _model, _batch, _fwd_bwd
from tests.gpu_helpers import analytically_zero_grad
model = _model()
x, t, w, _ = _batch()
for rep in range(3):
    l32, p32, g32 = _fwd_bwd(model, x, t, w, False)
    l16, p16, g16 = _fwd_bwd(model, x, t, w, True)
    tot32 = math.sqrt(sum(float(g.norm()) ** 2 for g in g32.values()))
    tot16 = math.sqrt(sum(float(g.norm()) ** 2 for g in g16.values()))
    dot = sum(float((g16[k] * g32[k]).sum()) for k in g32)
    gain = dot / tot32 ** 2
    rho = math.sqrt(max(tot16 ** 2 / tot32 ** 2 - gain ** 2, 0.0))
    # per-parameter gains of the big ones
    big = sorted(((float(g32[k].norm()), k) for k in g32), reverse=True)[:6]
    print(f"gain {gain:.5f} rho {rho:.4f} loss16-32 {l16-l32:.3f}", " ".join(f"{k.split('.')[-3] if k.count('.')>2 else k}:{float((g16[k]*g32[k]).sum())/float(g32[k].norm())**2:.4f}" for _, k in big), flush=True)
