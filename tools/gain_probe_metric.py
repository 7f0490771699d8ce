import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from tests.test_gpu_bf16_depth import _model, _fwd_bwd
from tests.gpu_helpers import synth_inputs, dev
model = _model()
import os
rng = np.random.default_rng(int(os.environ.get('PROBE_SEED', '20231122')))
x, targets, _ = synth_inputs(rng, 32, 32, 36, 64, (7863,))
x, t, w = torch.from_numpy(x).to(dev()), torch.from_numpy(targets[0]).to(dev()), torch.ones(32, 1, device=dev())
_, _, g32 = _fwd_bwd(model, x, t, w, False)
tot32 = math.sqrt(sum(float(g.norm()) ** 2 for g in g32.values()))
keys = ["core.stem.0.weight", "core.blocks.1.conv_pw.0.weight", "core.blocks.1.spat_covn_dw.0.weight", "core.blocks.1.conv_pwl.0.weight",
        "core.blocks.3.conv_pw.0.weight", "core.blocks.9.conv_pw.0.weight", "core.blocks.17.conv_pw.0.weight"]
for rep in range(2):
    _, _, g16 = _fwd_bwd(model, x, t, w, True)
    gain_all = sum(float((g16[k] * g32[k]).sum()) for k in g32) / tot32 ** 2
    print(f"all {gain_all:.4f}", " ".join(f"{k.split('.')[1]}.{k.split('.')[2]}.{k.split('.')[3]}:{float((g16[k]*g32[k]).sum())/float(g32[k].norm())**2:.4f}" for k in keys), flush=True)
